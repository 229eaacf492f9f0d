/*
 * colbwt_oracle.c -- CPU restatement of col-bwt's pml_query path, plain C.
 *
 * TEST INFRASTRUCTURE ONLY (see colbwt_oracle.h).  The product never links,
 * loads or calls this file.  Parity: pinned by the SURVEY.md Appendix D KAT
 * only; otherwise "parity unpinned" (reference unbuildable in this image).
 *
 * Citations are file:line under /root/reference.
 */
#define _GNU_SOURCE
#include "colbwt_oracle.h"

#include <ctype.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------ */
/* Row decoding.  col_thr is a packed bit-field struct dumped raw by   */
/* write_vec (LF_table.hpp:339, common.hpp:311-316):                   */
/*   byte 0      character  (LF_table.hpp:36, ALPHABET_BITS = 8)       */
/*   bytes 1-5   idx        (LF_table.hpp:37, BWT_BITS = 40)           */
/*   bytes 6-9   interval   (LF_table.hpp:38, RUN_BITS = 32)           */
/*   bytes 10-11 offset     (LF_table.hpp:39, LEN_BITS = 16)           */
/*   byte 12     col_id     (col_bwt.hpp:43,  ID_BITS = 8)             */
/*   bytes 13-17 threshold  (col_bwt.hpp:84,  BWT_BITS = 40)           */
/* ------------------------------------------------------------------ */

static inline uint64_t le_bytes(const uint8_t *p, unsigned nbytes)
{
    uint64_t v = 0;
    for (unsigned b = 0; b < nbytes; ++b) v |= (uint64_t)p[b] << (8u * b);
    return v;
}

static inline const uint8_t *rowp(const oracle_index *x, uint64_t i)
{
    return x->rows + i * (uint64_t)ORACLE_ROW_BYTES;
}

uint8_t  oracle_row_char(const oracle_index *x, uint64_t i)      { return rowp(x, i)[0]; }
uint64_t oracle_row_idx(const oracle_index *x, uint64_t i)       { return le_bytes(rowp(x, i) + 1, 5); }
uint64_t oracle_row_interval(const oracle_index *x, uint64_t i)  { return le_bytes(rowp(x, i) + 6, 4); }
uint64_t oracle_row_offset(const oracle_index *x, uint64_t i)    { return le_bytes(rowp(x, i) + 10, 2); }
uint8_t  oracle_row_col_id(const oracle_index *x, uint64_t i)    { return rowp(x, i)[12]; }
uint64_t oracle_row_threshold(const oracle_index *x, uint64_t i) { return le_bytes(rowp(x, i) + 13, 5); }

/* LF_table::get_length, LF_table.hpp:204-207:
 *   (i == r - 1) ? (n - get_idx(i)) : (get_idx(i + 1) - get_idx(i)) */
uint64_t oracle_get_length(const oracle_index *x, uint64_t i)
{
    return (i == x->r - 1) ? (x->n - oracle_row_idx(x, i))
                           : (oracle_row_idx(x, i + 1) - oracle_row_idx(x, i));
}

/* ------------------------------------------------------------------ */
/* Loading: col_bwt::load reads bwt_r (col_bwt.hpp:375-380), then      */
/* LF_table::load reads n, r, size and one raw blob of size*sizeof(row)*/
/* (LF_table.hpp:347-357; read_vec common.hpp:318-323).                */
/* ------------------------------------------------------------------ */

int oracle_load_memory(const uint8_t *bytes, uint64_t len, oracle_index *out)
{
    if (len < ORACLE_HEADER_BYTES) return -1;
    out->bwt_r = le_bytes(bytes + 0, 8);
    out->n     = le_bytes(bytes + 8, 8);
    out->r     = le_bytes(bytes + 16, 8);
    out->size  = le_bytes(bytes + 24, 8);
    if (out->size > (len - ORACLE_HEADER_BYTES) / ORACLE_ROW_BYTES) return -2;
    out->rows = bytes + ORACLE_HEADER_BYTES;
    out->owned = NULL;
    return 0;
}

int oracle_load_file(const char *path, oracle_index *out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return -1; }
    long long len = ftello(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc((size_t)len ? (size_t)len : 1);
    if (!buf) { fclose(f); return -3; }
    size_t got = fread(buf, 1, (size_t)len, f);
    fclose(f);
    if ((long long)got != len) { free(buf); return -1; }
    int rc = oracle_load_memory(buf, (uint64_t)len, out);
    if (rc != 0) { free(buf); return rc; }
    out->owned = buf;
    return 0;
}

void oracle_free(oracle_index *idx)
{
    if (idx && idx->owned) { free(idx->owned); idx->owned = NULL; idx->rows = NULL; }
}

/* ------------------------------------------------------------------ */
/* LF_table::LF, LF_table.hpp:251-262, and LF_idx :264-268.            */
/* ------------------------------------------------------------------ */
static inline void oracle_LF_idx(const oracle_index *x, uint64_t run, uint64_t offset,
                                 uint64_t *ni, uint64_t *no, uint64_t *npos)
{
    uint64_t next_interval = oracle_row_interval(x, run);          /* :253 */
    uint64_t next_offset = oracle_row_offset(x, run) + offset;     /* :254 */
    while (next_offset >= oracle_get_length(x, next_interval)) {   /* :256 */
        next_offset -= oracle_get_length(x, next_interval++);      /* :258 */
    }
    *ni = next_interval;
    *no = next_offset;
    *npos = oracle_row_idx(x, next_interval) + next_offset;        /* to_idx :214-217 */
}

/* LF_table::pred_char, LF_table.hpp:271-283: largest run <= `run` whose char
 * is c; offset = len-1.  Returns 0 when the scan passes run 0. */
static inline int oracle_pred_char(const oracle_index *x, uint64_t run, uint8_t c,
                                   uint64_t *q, uint64_t *qoff)
{
    while (oracle_row_char(x, run) != c) {
        if (run == 0) return 0;
        --run;
    }
    *q = run;
    *qoff = oracle_get_length(x, run) - 1;
    return 1;
}

/* LF_table::succ_char, LF_table.hpp:286-298: smallest run >= `run` whose char
 * is c; offset 0.  Returns 0 when the scan passes run r-1. */
static inline int oracle_succ_char(const oracle_index *x, uint64_t run, uint8_t c,
                                   uint64_t *s, uint64_t *soff)
{
    while (oracle_row_char(x, run) != c) {
        if (run == x->r - 1) return 0;
        ++run;
    }
    *s = run;
    *soff = 0;
    return 1;
}

/* col_pml::threshold_step, col_bwt.hpp:531-574 (the !MULTI_THREAD arm; the
 * MULTI_THREAD arm computes the same values). */
static inline void oracle_threshold_step(const oracle_index *x, uint64_t *interval,
                                         uint64_t *offset, uint64_t pos, uint8_t c)
{
    uint64_t new_interval = *interval;                      /* :533 */
    uint64_t new_offset = *offset;                          /* :534 */
    uint64_t thr = x->n;                                    /* :535 */
    uint64_t s, soff;
    if (oracle_succ_char(x, *interval, c, &s, &soff)) {     /* :548,:552 */
        thr = oracle_row_threshold(x, s);                   /* :554 */
        new_interval = s;                                   /* :555 */
        new_offset = soff;                                  /* :556 */
    }
    if (pos < thr) {                                        /* :560 */
        uint64_t q, qoff;
        if (oracle_pred_char(x, *interval, c, &q, &qoff)) { /* :562,:565 */
            new_interval = q;                               /* :567 */
            new_offset = qoff;                              /* :568 */
        }
    }
    *interval = new_interval;                               /* :572 */
    *offset = new_offset;                                   /* :573 */
}

/* col_pml::_query_pml core, col_bwt.hpp:498-529. */
void oracle_query_pml(const oracle_index *x, const uint8_t *pattern, uint64_t m,
                      uint64_t *pml, uint64_t *cid)
{
    uint64_t pos = x->n - 1;                                 /* :503 */
    uint64_t interval = x->r - 1;                            /* :504 */
    uint64_t offset = oracle_get_length(x, interval) - 1;    /* :505 */
    uint64_t length = 0;                                     /* :507 */
    uint64_t col_id = 0;                                     /* :508 */

    for (uint64_t i = 0; i < m; ++i) {                       /* :510 */
        uint8_t c = pattern[m - i - 1];                      /* :512 */
        col_id = oracle_row_col_id(x, interval);             /* :513 */
        if (oracle_row_char(x, interval) == c) {             /* :516 */
            ++length;                                        /* :517 */
        } else {
            length = 0;                                      /* :521 */
            oracle_threshold_step(x, &interval, &offset, pos, c); /* :522 */
        }
        pml[m - i - 1] = length;                             /* :525 via :463-466 */
        cid[m - i - 1] = col_id;
        oracle_LF_idx(x, interval, offset, &interval, &offset, &pos); /* :527 */
    }
}

/* ------------------------------------------------------------------ */
/* Batch helpers (test/baseline plumbing, no reference counterpart:    */
/* the reference loops over reads sequentially, pml_query.cpp:74-86).  */
/* ------------------------------------------------------------------ */
typedef struct batch_job {
    const oracle_index *x;
    const uint8_t *bases;
    const uint64_t *read_off;
    uint64_t lo, hi;
    void *pml;
    uint8_t *cid;
    int wide;
} batch_job;

static void *batch_worker(void *arg)
{
    batch_job *j = (batch_job *)arg;
    uint64_t cap = 0;
    uint64_t *tp = NULL, *tc = NULL;
    for (uint64_t k = j->lo; k < j->hi; ++k) {
        uint64_t b = j->read_off[k], m = j->read_off[k + 1] - b;
        if (m > cap) {
            cap = m * 2 + 16;
            tp = (uint64_t *)realloc(tp, cap * sizeof(uint64_t));
            tc = (uint64_t *)realloc(tc, cap * sizeof(uint64_t));
        }
        oracle_query_pml(j->x, j->bases + b, m, tp, tc);
        if (j->wide) {
            uint32_t *o = (uint32_t *)j->pml + b;
            for (uint64_t t = 0; t < m; ++t) o[t] = (uint32_t)tp[t];
        } else {
            uint16_t *o = (uint16_t *)j->pml + b;
            for (uint64_t t = 0; t < m; ++t) o[t] = (uint16_t)tp[t];
        }
        for (uint64_t t = 0; t < m; ++t) j->cid[b + t] = (uint8_t)tc[t];
    }
    free(tp);
    free(tc);
    return NULL;
}

static void batch_run(const oracle_index *x, const uint8_t *bases, const uint64_t *read_off,
                      uint64_t n_reads, void *pml, uint8_t *cid, int threads, int wide)
{
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n_reads) threads = n_reads ? (int)n_reads : 1;
    batch_job *jobs = (batch_job *)calloc((size_t)threads, sizeof(batch_job));
    pthread_t *tids = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    uint64_t total = n_reads ? read_off[n_reads] - read_off[0] : 0;
    uint64_t k = 0;
    for (int t = 0; t < threads; ++t) {
        /* contiguous shards balanced by base count */
        uint64_t target = read_off[0] + (total * (uint64_t)(t + 1)) / (uint64_t)threads;
        uint64_t lo = k;
        while (k < n_reads && (t == threads - 1 || read_off[k + 1] <= target)) ++k;
        jobs[t] = (batch_job){x, bases, read_off, lo, k, pml, cid, wide};
    }
    if (threads == 1) {
        batch_worker(&jobs[0]);
    } else {
        for (int t = 0; t < threads; ++t) pthread_create(&tids[t], NULL, batch_worker, &jobs[t]);
        for (int t = 0; t < threads; ++t) pthread_join(tids[t], NULL);
    }
    free(jobs);
    free(tids);
}

void oracle_query_batch_u16(const oracle_index *x, const uint8_t *bases, const uint64_t *read_off,
                            uint64_t n_reads, uint16_t *pml, uint8_t *cid, int threads)
{
    batch_run(x, bases, read_off, n_reads, pml, cid, threads, 0);
}

void oracle_query_batch_u32(const oracle_index *x, const uint8_t *bases, const uint64_t *read_off,
                            uint64_t n_reads, uint32_t *pml, uint8_t *cid, int threads)
{
    batch_run(x, bases, read_off, n_reads, pml, cid, threads, 1);
}

/* ------------------------------------------------------------------ */
/* Text output: pml_to_vec, pml_query.cpp:78-85:                       */
/*   fs << '>' << id << " \n"; copy(vals, ostream_iterator<size_t>(fs, " ")); fs << "\n"; */
/* ------------------------------------------------------------------ */
int oracle_write_text(void *file, const char *name, const uint64_t *vals, uint64_t m)
{
    FILE *f = (FILE *)file;
    if (fprintf(f, ">%s \n", name) < 0) return -1;
    for (uint64_t k = 0; k < m; ++k)
        if (fprintf(f, "%llu ", (unsigned long long)vals[k]) < 0) return -1;
    if (fputc('\n', f) == EOF) return -1;
    return 0;
}

/* ------------------------------------------------------------------ */
/* FASTA/FASTQ reading: PatternProcessor (io.hpp:6-35) = gzopen +      */
/* kseq_read.  klib is not in the container; this restates the         */
/* published kseq.h (attractivechaos/klib, kseq_read / ks_getuntil2)   */
/* semantics: name = header up to first isspace(); comment dropped;    */
/* sequence lines concatenated verbatim (no case folding), '\r' before */
/* '\n' stripped when the accumulated length > 1; a record ends at a   */
/* line starting with '>', '@' or '+'; FASTQ qualities are consumed    */
/* until qual.l >= seq.l.  PARITY UNPINNED for these edge cases.       */
/* ------------------------------------------------------------------ */
typedef struct kstream {
    gzFile fp;
    unsigned char buf[16384];
    int begin, end, is_eof;
} kstream;

static int ks_getc(kstream *ks)
{
    if (ks->is_eof && ks->begin >= ks->end) return -1;
    if (ks->begin >= ks->end) {
        ks->begin = 0;
        ks->end = gzread(ks->fp, ks->buf, sizeof(ks->buf));
        if (ks->end == 0) { ks->is_eof = 1; return -1; }
        if (ks->end < 0) { ks->is_eof = 1; return -3; }
    }
    return (int)ks->buf[ks->begin++];
}

typedef struct kstr { char *s; size_t l, m; } kstr;

static void kstr_push(kstr *k, int c)
{
    if (k->l + 2 > k->m) { k->m = k->m ? k->m * 2 : 256; k->s = (char *)realloc(k->s, k->m); }
    k->s[k->l++] = (char)c;
    k->s[k->l] = 0;
}

#define SEP_SPACE 0
#define SEP_LINE 2

/* ks_getuntil2: read up to (not including) the delimiter; returns str->l or
 * -1 on EOF with nothing read. */
static long ks_getuntil2(kstream *ks, int delimiter, kstr *str, int *dret, int append)
{
    int gotany = 0, c;
    if (dret) *dret = 0;
    if (!append) str->l = 0;
    for (;;) {
        c = ks_getc(ks);
        if (c < 0) break;
        gotany = 1;
        if (delimiter == SEP_LINE ? (c == '\n') : isspace(c)) {
            if (dret) *dret = c;
            break;
        }
        kstr_push(str, c);
    }
    if (!gotany && c < 0) return c == -3 ? -3 : -1;
    if (str->s == NULL) { str->m = 1; str->s = (char *)calloc(1, 1); }
    else if (delimiter == SEP_LINE && str->l > 1 && str->s[str->l - 1] == '\r') --str->l;
    str->s[str->l] = '\0';
    return (long)str->l;
}

typedef struct kseq {
    kstr name, comment, seq, qual;
    int last_char;
    kstream ks;
} kseq;

static long kseq_read(kseq *seq)
{
    int c;
    long r;
    kstream *ks = &seq->ks;
    if (seq->last_char == 0) {
        while ((c = ks_getc(ks)) >= 0 && c != '>' && c != '@') {}
        if (c < 0) return c;
        seq->last_char = c;
    }
    seq->comment.l = seq->seq.l = seq->qual.l = 0;
    if ((r = ks_getuntil2(ks, SEP_SPACE, &seq->name, &c, 0)) < 0) return r;
    if (c != '\n') ks_getuntil2(ks, SEP_LINE, &seq->comment, NULL, 0);
    if (seq->seq.s == NULL) { seq->seq.m = 256; seq->seq.s = (char *)malloc(seq->seq.m); seq->seq.s[0] = 0; }
    while ((c = ks_getc(ks)) >= 0 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;
        kstr_push(&seq->seq, c);
        ks_getuntil2(ks, SEP_LINE, &seq->seq, NULL, 1);
    }
    if (c == '>' || c == '@') seq->last_char = c;
    seq->seq.s[seq->seq.l] = 0;
    if (c != '+') return (long)seq->seq.l;
    while ((c = ks_getc(ks)) >= 0 && c != '\n') {}
    if (c == -1) return -2;
    while ((r = ks_getuntil2(ks, SEP_LINE, &seq->qual, NULL, 1)) >= 0 && seq->qual.l < seq->seq.l) {}
    if (r == -3) return -3;
    seq->last_char = 0;
    if (seq->seq.l != seq->qual.l) return -2;
    return (long)seq->seq.l;
}

/* pml_query main in vec mode: pml_query.cpp:110-131 + pml_to_vec :65-90. */
int oracle_pml_query_files(const oracle_index *x, const char *pattern_path,
                           const char *pml_path, const char *cid_path)
{
    kseq *seq = (kseq *)calloc(1, sizeof(kseq));
    seq->ks.fp = gzopen(pattern_path, "r");                 /* io.hpp:9 */
    if (!seq->ks.fp) { free(seq); return -1; }
    FILE *fp = fopen(pml_path, "w");                        /* pml_query.cpp:67 */
    FILE *fc = fopen(cid_path, "w");                        /* pml_query.cpp:70 */
    if (!fp || !fc) { if (fp) fclose(fp); if (fc) fclose(fc); gzclose(seq->ks.fp); free(seq); return -2; }
    uint64_t cap = 0;
    uint64_t *pml = NULL, *cid = NULL;
    long l;
    while ((l = kseq_read(seq)) >= 0) {                     /* io.hpp:13-15 */
        uint64_t m = (uint64_t)seq->seq.l;
        if (m > cap) {
            cap = 2 * m + 16;
            pml = (uint64_t *)realloc(pml, cap * sizeof(uint64_t));
            cid = (uint64_t *)realloc(cid, cap * sizeof(uint64_t));
        }
        oracle_query_pml(x, (const uint8_t *)seq->seq.s, m, pml, cid);  /* :76 */
        oracle_write_text(fp, seq->name.s ? seq->name.s : "", pml, m);  /* :79-81 */
        oracle_write_text(fc, seq->name.s ? seq->name.s : "", cid, m);  /* :83-85 */
    }
    fclose(fp);
    fclose(fc);
    gzclose(seq->ks.fp);
    free(pml); free(cid);
    free(seq->name.s); free(seq->comment.s); free(seq->seq.s); free(seq->qual.s);
    free(seq);
    return 0;
}

/* ------------------------------------------------------------------ */
/* Builder restatement.                        */
/* ------------------------------------------------------------------ */
typedef struct brow { uint8_t c; uint64_t idx, interval, offset, id, thr; } brow;

static void put_le(uint8_t *p, uint64_t v, unsigned nbytes)
{
    for (unsigned b = 0; b < nbytes; ++b) p[b] = (uint8_t)(v >> (8u * b));
}

uint64_t oracle_build_col_pml(const uint8_t *heads, uint64_t n_heads, const uint64_t *lens,
                              const uint8_t *col_ids, uint64_t n_ids, const uint64_t *split_pos,
                              uint64_t n_splits, const uint64_t *thr_pos, uint64_t n_thr,
                              uint8_t *out, uint64_t out_cap)
{
    /* col_bwt(heads, lengths, col_ids, splits), col_bwt.hpp:124-230 */
    uint64_t cap = n_heads + n_splits + 1, nrows = 0;
    brow *rows = (brow *)calloc(cap, sizeof(brow));
    uint64_t s_set_bits = n_splits;                               /* :141-142 */
    uint64_t n = 0, bwt_r = 0, s = 0, id_pos = 0;                 /* :158-163 */
    uint64_t s_curr = n_splits ? split_pos[0] : 0;                /* :165 s_select(s + 1) */
    uint64_t curr_id = 0;                                         /* :166 */
#define SELECT_NEXT() ((s < s_set_bits) ? split_pos[s] : 0)       /* s_select(s + 1), else 0 (:180,:198) */
#define READ_ID() do { if (id_pos < n_ids) curr_id = col_ids[id_pos++]; } while (0) /* 1 byte into a zeroed size_t */
    for (uint64_t h = 0; h < n_heads; ++h) {                      /* :167 while ((c = heads.get()) != EOF) */
        signed char c = (signed char)heads[h];
        if (c == (signed char)EOF) break;                         /* char 0xFF compares equal to EOF */
        uint64_t length = lens[h];                                /* :169-170 */
        if (c <= 1) c = 1;                                        /* :171 TERMINATOR; bytes >= 0x80 are negative chars */
        uint64_t run_end = n + length;                            /* :176 */
        if (s_curr == n) {                                        /* :177-181 */
            READ_ID();
            ++s;
            s_curr = SELECT_NEXT();
        }
        while (s < s_set_bits && s_curr < run_end) {              /* :183 */
            rows[nrows].c = (uint8_t)c; rows[nrows].idx = n; rows[nrows].id = curr_id; ++nrows;  /* :184-185 */
            uint64_t delta = s_curr - n;                          /* :186 */
            n += delta;                                           /* :187 */
            length -= delta;                                      /* :188 */
            ++s;                                                  /* :197 */
            s_curr = SELECT_NEXT();                               /* :198 */
            READ_ID();                                            /* :199 */
        }
        if (length > 0) {                                         /* :202 */
            rows[nrows].c = (uint8_t)c; rows[nrows].idx = n; rows[nrows].id = curr_id; ++nrows;  /* :203-204 */
            n += length;                                          /* :205 */
        }
        ++bwt_r;                                                  /* :214 */
    }
    uint64_t r = nrows;                                           /* :216 */
#undef SELECT_NEXT
#undef READ_ID

    /* LF_table::compute_table, LF_table.hpp:365-387: L_block_indices[c] lists the rows
     * holding c in row order; the outer loop runs over c ascending. */
#define BLEN(i) (((i) == r - 1 ? n : rows[(i) + 1].idx) - rows[i].idx)  /* get_length :204-207 */
    uint64_t curr_L_num = 0, L_seen = 0, F_seen = 0;              /* :366-368 */
    for (unsigned cc = 0; cc < 256; ++cc) {                       /* :369 */
        for (uint64_t pos = 0; pos < r; ++pos) {                  /* :371 (rows with char cc, ascending) */
            if (rows[pos].c != cc) continue;
            rows[pos].interval = curr_L_num & 0xFFFFFFFFull;      /* :375, RUN_BITS bit-field */
            rows[pos].offset = (F_seen - L_seen) & 0xFFFFull;     /* :376, LEN_BITS bit-field */
            F_seen += BLEN(pos);                                  /* :378 */
            while (curr_L_num < r && F_seen >= L_seen + BLEN(curr_L_num)) {  /* :380 */
                L_seen += BLEN(curr_L_num);                       /* :382 */
                ++curr_L_num;                                     /* :383 */
            }
        }
    }
#undef BLEN

    /* col_pml::read_thresholds, col_bwt.hpp:440-457 */
    {
        uint64_t i = 0;
        for (uint64_t k = 0; k < n_thr && i < r; ++k) {           /* :446 while (thresholds.read(...)) */
            uint8_t last = rows[i].c;                             /* :448 */
            do {
                rows[i++].thr = thr_pos[k] & 0xFFFFFFFFFFull;     /* :450, BWT_BITS bit-field */
            } while (i < r && rows[i].c == last);                 /* :451 */
        }
    }

    /* col_bwt::serialize + LF_table::serialize (col_bwt.hpp:360-370, LF_table.hpp:325-342) */
    uint64_t need = ORACLE_HEADER_BYTES + r * (uint64_t)ORACLE_ROW_BYTES;
    if (need > out_cap) { free(rows); return 0; }
    put_le(out + 0, bwt_r, 8);
    put_le(out + 8, n, 8);
    put_le(out + 16, r, 8);
    put_le(out + 24, r, 8);
    for (uint64_t i = 0; i < r; ++i) {
        uint8_t *p = out + ORACLE_HEADER_BYTES + i * ORACLE_ROW_BYTES;
        p[0] = rows[i].c;
        put_le(p + 1, rows[i].idx & 0xFFFFFFFFFFull, 5);
        put_le(p + 6, rows[i].interval, 4);
        put_le(p + 10, rows[i].offset, 2);
        p[12] = (uint8_t)rows[i].id;                              /* col_row ctor :47-52: ids < 256 unchanged */
        put_le(p + 13, rows[i].thr, 5);
    }
    free(rows);
    return need;
}
