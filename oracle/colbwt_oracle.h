/*
 * colbwt_oracle.h -- CPU restatement of col-bwt's pml_query path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.
 *
 * Parity status: the reference (drnatebrown/col-bwt @ 2024-12-20) cannot be
 * compiled in this image (it needs sdsl-lite, klib/kseq.h and malloc_count,
 * all network FetchContent, thirdparty/CMakeLists.txt:5-87) and ships no
 * fixtures of its own.  The oracle is pinned by the one known-answer vector
 * recorded in SURVEY.md Appendix D (302-byte index, 3 reads; produced by the
 * reference during the survey).  Everything beyond that vector is a
 * line-by-line restatement with citations: "parity unpinned" beyond the KAT.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef COLBWT_ORACLE_H
#define COLBWT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Format constants: include/common/common.hpp:46-54 (RW_BYTES 5, ID_BITS 8,
 * BWT_BYTES 5, RUN_BYTES 4, LEN_BYTES 2; DNA_ALPHABET off => 8-bit chars). */
#define ORACLE_ROW_BYTES 18u   /* sizeof(col_thr), packed: col_bwt.hpp:81-115 */
#define ORACLE_HEADER_BYTES 32u

typedef struct oracle_index {
    uint64_t bwt_r;        /* col_bwt.hpp:383  */
    uint64_t n;            /* LF_table.hpp:360 */
    uint64_t r;            /* LF_table.hpp:361 */
    uint64_t size;         /* vector length, LF_table.hpp:349-355 */
    const uint8_t *rows;   /* size * 18 bytes, raw memory image of col_thr */
    void *owned;           /* malloc'ed block when the oracle owns the bytes */
} oracle_index;

/* col_bwt::load -> LF_table::load -> read_vec
 * (col_bwt.hpp:375-380; LF_table.hpp:347-357; common.hpp:318-323). */
int oracle_load_file(const char *path, oracle_index *out);
/* Same, over a caller-owned memory image of the file (no copy). */
int oracle_load_memory(const uint8_t *bytes, uint64_t len, oracle_index *out);
void oracle_free(oracle_index *idx);

/* Field accessors over the packed row (Appendix A of SURVEY.md; bit-field
 * layout of LF_row/col_row/col_thr, LF_table.hpp:33-40, col_bwt.hpp:43,84). */
uint8_t  oracle_row_char(const oracle_index *x, uint64_t i);
uint64_t oracle_row_idx(const oracle_index *x, uint64_t i);
uint64_t oracle_row_interval(const oracle_index *x, uint64_t i);
uint64_t oracle_row_offset(const oracle_index *x, uint64_t i);
uint8_t  oracle_row_col_id(const oracle_index *x, uint64_t i);
uint64_t oracle_row_threshold(const oracle_index *x, uint64_t i);
/* LF_table::get_length, LF_table.hpp:204-207 */
uint64_t oracle_get_length(const oracle_index *x, uint64_t i);

/* col_pml::query_pml(const char*, size_t) (col_bwt.hpp:409-412, 460-472,
 * 498-529): pml[k], cid[k] <-> pattern[k].  64-bit outputs like the
 * reference's vector<ulint>. */
void oracle_query_pml(const oracle_index *x, const uint8_t *pattern, uint64_t m,
                      uint64_t *pml, uint64_t *cid);

/* Batch convenience for the parity tests / CPU baseline: reads concatenated in
 * `bases`, read k = bases[read_off[k] .. read_off[k+1]).  Narrow outputs
 * (u16 saturating is NOT applied: values are truncated exactly like a cast;
 * callers use pml32 when a read is longer than 65535).  `threads` > 1 shards
 * reads over pthreads (the reference itself is sequential over reads,
 * pml_query.cpp:74). */
void oracle_query_batch_u16(const oracle_index *x, const uint8_t *bases,
                            const uint64_t *read_off, uint64_t n_reads,
                            uint16_t *pml, uint8_t *cid, int threads);
void oracle_query_batch_u32(const oracle_index *x, const uint8_t *bases,
                            const uint64_t *read_off, uint64_t n_reads,
                            uint32_t *pml, uint8_t *cid, int threads);

/* pml_to_vec text format (pml_query.cpp:65-90): for each read
 * '>' name ' ' '\n' then every value followed by one space, then '\n'.
 * Appends to an open FILE* (passed as void* to keep the header C-ABI plain). */
int oracle_write_text(void *file, const char *name, const uint64_t *vals, uint64_t m);

/* Whole-program restatement of pml_query's vec mode (pml_query.cpp:92-143 with
 * pml_to_vec :65-90 and PatternProcessor io.hpp:6-35): reads FASTA/FASTQ
 * (gz-transparent) from pattern_path, writes <pattern>.pml / <pattern>.cid.
 * The FASTA/FASTQ tokenisation restates klib's kseq_read (klib is an
 * un-vendored dependency, thirdparty/CMakeLists.txt:22-32, unpinned fork
 * drnatebrown/klib) -- parity unpinned for reader edge cases. */
int oracle_pml_query_files(const oracle_index *x, const char *pattern_path,
                           const char *pml_path, const char *cid_path);


/* ------------------------------------------------------------------------
 * SURVEY.md 8(f) "next" #1: the in-repo index builder, build_col_bwt
 * (src/build_col_bwt.cpp:38-52): col_pml(heads, lengths, col_ids, thresholds,
 * splits) = col_bwt ctor (col_bwt.hpp:124-230) + LF_table::compute_table
 * (LF_table.hpp:365-387) + col_pml::read_thresholds (col_bwt.hpp:440-457) +
 * col_bwt::serialize (col_bwt.hpp:360-370).  Inputs are the decoded streams:
 *   heads[n_heads]          .bwt.heads bytes (col_bwt.hpp:167)
 *   lens[n_heads]           .bwt.len 5-byte values, already widened (:170)
 *   col_ids[n_ids]          .col_ids bytes (:178,:199)
 *   split_pos[n_splits]     positions of the set bits of .col_runs, ascending
 *                           (what s_select(k) returns, :165,:180,:198)
 *   thr_pos[n_thr]          .thr_pos 5-byte values, widened (:446)
 * Writes the .col_pml image into out (capacity out_cap); returns its length,
 * or 0 when out_cap is too small.  Parity: pinned by the Appendix D KAT (its
 * 302-byte index was written by this reference constructor). */
uint64_t oracle_build_col_pml(const uint8_t *heads, uint64_t n_heads, const uint64_t *lens,
                              const uint8_t *col_ids, uint64_t n_ids, const uint64_t *split_pos,
                              uint64_t n_splits, const uint64_t *thr_pos, uint64_t n_thr,
                              uint8_t *out, uint64_t out_cap);
#ifdef __cplusplus
}
#endif
#endif
