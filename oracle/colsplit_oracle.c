/*
 * colsplit_oracle.c -- CPU restatement of the reference's sub-run splitter:
 * build_FL (src/build_FL.cpp:27-74 -> FL_table, include/ds/FL_table.hpp) followed by col_split
 * (src/col_split.cpp:62-140 -> col_split<>, include/col_split.hpp).
 *
 * TEST INFRASTRUCTURE ONLY (see colbwt_oracle.h): the checker of the product's
 * colbwt_col_split, never a product path.
 *
 * Parity status: the reference cannot be compiled here (sdsl-lite bit_vector / sd_vector /
 * rank / select are part of this path, thirdparty/CMakeLists.txt:5-20) and ships no fixtures.
 * Pinned only by the counts the survey recorded from the compiled reference (SURVEY.md
 * Appendix C.6: the Appendix D text, one multi-MUM GATTAC, len 6, SA rank 15, N = 2:
 * tunnels => 15 sub-runs / 5 col runs / 10 col chars; all => 16 / 7 / 12) --
 * tests/test_col_split.py.  Beyond those counts: a line-by-line restatement, "parity unpinned".
 * The `.FL_table` file is not read or written (it embeds an sdsl sd_vector whose byte format
 * cannot be checked here): the table is rebuilt from .bwt.heads / .bwt.len as build_FL does.
 *
 * sdsl pieces restated by their documented meaning: bit_vector(n, 0); rank_1(i) = number of
 * ones in [0, i); select_1(k) = position of the k-th one, k >= 1; sd_vector of the L run heads
 * (FL_table.hpp:378-391): select(k) = start of the k-th BWT run.  select past the last one is
 * undefined in sdsl; the reference only reaches it with run_cursor = r + 1 (col_split.hpp:
 * 296-306), where any value >= n gives the same result -- n is used.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "colsplit_oracle.h"

/* ---- FL_table --------------------------------------------------------------------------- */

/* FL_table(std::ifstream &heads, std::ifstream &lengths), FL_table.hpp:82-130 + compute_table
 * :343-376 + compute_L_heads :378-391. */
int oracle_fl_build(const uint8_t *heads, uint64_t heads_len, const uint64_t *lens, oracle_fl *t) {
    memset(t, 0, sizeof(*t));
    /* :99-113  while ((c = heads.get()) != EOF): `char c`, so byte 0xFF ends the stream; c <= TERMINATOR
     * (signed compare: bytes >= 0x80 too) becomes TERMINATOR = 1 */
    uint64_t r = 0;
    while (r < heads_len && heads[r] != 0xFF) ++r;
    if (r == 0) return -1;
    uint8_t *L_chars = (uint8_t *)malloc(r);
    uint64_t *L_start = (uint64_t *)malloc((r + 1) * sizeof(uint64_t));
    uint64_t n = 0;
    uint64_t cnt[256];
    memset(cnt, 0, sizeof(cnt));
    for (uint64_t i = 0; i < r; ++i) {
        int c = (int)(signed char)heads[i];
        if (c <= 1) c = 1;                       /* :102 */
        L_chars[i] = (uint8_t)c;
        L_start[i] = n;                          /* :111 n += length */
        n += lens[i];
        ++cnt[c];                                /* char_runs[c].push_back(length) :109 */
    }
    L_start[r] = n;
    t->n = n;
    t->r = r;
    t->ch = (uint8_t *)malloc(r);
    t->idx = (uint64_t *)malloc((r + 1) * sizeof(uint64_t));
    t->interval = (uint64_t *)malloc(r * sizeof(uint64_t));
    t->offset = (uint16_t *)malloc(r * sizeof(uint16_t));
    t->L_head = L_start;                         /* compute_L_heads: one bit per L run start */
    /* :345-357 F order: characters ascending, runs of one character in L order; idx = running sum */
    uint64_t first_of[257];
    first_of[0] = 0;
    for (int c = 0; c < 256; ++c) first_of[c + 1] = first_of[c] + cnt[c];
    uint64_t *slot = (uint64_t *)malloc(256 * sizeof(uint64_t));
    uint64_t *len_F = (uint64_t *)malloc(r * sizeof(uint64_t));
    uint64_t *L_of = (uint64_t *)malloc(r * sizeof(uint64_t));   /* L run of F row k (L_block_indices, in order) */
    memcpy(slot, first_of, 256 * sizeof(uint64_t));
    for (uint64_t i = 0; i < r; ++i) {
        const uint64_t k = slot[L_chars[i]]++;
        t->ch[k] = L_chars[i];
        len_F[k] = lens[i];
        L_of[k] = i;
    }
    uint64_t curr = 0;
    for (uint64_t k = 0; k < r; ++k) {
        t->idx[k] = curr;
        curr += len_F[k];
    }
    t->idx[r] = n;
    /* :359-375 per character: scan L and F in step; row k gets the F run holding its L start */
    for (int c = 0; c < 256; ++c) {
        uint64_t F_curr = 0, F_seen = 0;
        for (uint64_t k = first_of[c]; k < first_of[c + 1]; ++k) {
            const uint64_t L_seen = L_start[L_of[k]];          /* :366-368 */
            while (F_seen + (t->idx[F_curr + 1] - t->idx[F_curr]) <= L_seen) {   /* :369-371 get_length(F_curr) */
                F_seen += t->idx[F_curr + 1] - t->idx[F_curr];
                ++F_curr;
            }
            t->interval[k] = F_curr;                            /* :373 */
            t->offset[k] = (uint16_t)(L_seen - F_seen);         /* :374 `ulint offset : LEN_BITS` keeps 16 bits */
        }
    }
    free(slot);
    free(len_F);
    free(L_of);
    free(L_chars);
    return 0;
}

void oracle_fl_free(oracle_fl *t) {
    free(t->ch);
    free(t->idx);
    free(t->interval);
    free(t->offset);
    free(t->L_head);
    memset(t, 0, sizeof(*t));
}

/* get_length, FL_table.hpp:245-248 (idx[r] = n removes the last-row case) */
static uint64_t fl_len(const oracle_fl *t, uint64_t i) { return t->idx[i + 1] - t->idx[i]; }

/* FL, FL_table.hpp:227-238 */
static void fl_step(const oracle_fl *t, uint64_t run, uint64_t offset, uint64_t *out_run, uint64_t *out_off) {
    uint64_t next_interval = t->interval[run];
    uint64_t next_offset = (uint64_t)t->offset[run] + offset;
    while (next_offset >= fl_len(t, next_interval)) next_offset -= fl_len(t, next_interval++);
    *out_run = next_interval;
    *out_off = next_offset;
}

/* ---- col_split -------------------------------------------------------------------------- */

typedef struct {
    uint64_t interval, offset;
    uint16_t height;                               /* len_t = uint16_t, col_split.hpp:59,180-184 */
} range_t;

typedef struct {
    range_t *v;
    uint64_t n, cap;
} range_vec;

static void rv_push(range_vec *a, range_t x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? 2 * a->cap : 16;
        a->v = (range_t *)realloc(a->v, a->cap * sizeof(range_t));
    }
    a->v[a->n++] = x;
}

/* FL_range, col_split.hpp:226-247 */
static void FL_range(const oracle_fl *t, range_t rg, range_vec *out) {
    while (rg.height > 0) {
        uint64_t FL_interval, FL_offset;
        fl_step(t, rg.interval, rg.offset, &FL_interval, &FL_offset);
        if (rg.offset + rg.height > fl_len(t, rg.interval)) {
            const uint16_t covered = (uint16_t)(fl_len(t, rg.interval) - rg.offset);   /* :234 len_t */
            rv_push(out, (range_t){FL_interval, FL_offset, covered});
            rg.height = (uint16_t)(rg.height - covered);
            rg.offset = 0;
        } else {
            rv_push(out, (range_t){FL_interval, FL_offset, rg.height});
            rg.height = 0;
        }
        ++rg.interval;                              /* :244 */
    }
}

static uint8_t bin_id(uint64_t id) {               /* col_split.hpp:222-224, id_max = bit_max(ID_BITS) = 256 */
    return (uint8_t)((id >= 256) ? (id % 255) + 1 : id);
}

typedef void (*mark_fn)(void *ctx, uint64_t col_span_start, uint64_t c_id, uint16_t height);

/* FL_loop, col_split.hpp:69-109 */
static void FL_loop(const oracle_fl *t, const uint64_t *col_len, const uint64_t *col_pos, uint64_t n_cols, uint16_t N,
                    int mode_all, int split_rate, mark_fn func, void *ctx) {
    uint64_t run_start = 0, c_id = 1, c = 0;        /* :70-73: 0 denotes no id */
    range_vec FL_ranges = {0, 0, 0}, next_ranges = {0, 0, 0};
    for (uint64_t i = 0; i < t->r; ++i) {
        const uint64_t run_len = fl_len(t, i);
        while (c < n_cols && col_pos[c] >= run_start && col_pos[c] < run_start + run_len) {   /* :77 */
            range_t rg = {i, col_pos[c] - run_start, N};
            FL_ranges.n = 0;
            FL_range(t, rg, &FL_ranges);                                                    /* :79 */
            int skip_non_tunnel = !mode_all && FL_ranges.n > 1;                             /* :81 */
            for (uint64_t j = 0; j < col_len[c] && !skip_non_tunnel; ++j) {
                next_ranges.n = 0;
                for (uint64_t k = 0; k < FL_ranges.n; ++k) {
                    rg = FL_ranges.v[k];
                    if (j % (uint64_t)split_rate == 0)                                      /* :87-93 */
                        func(ctx, t->idx[rg.interval] + rg.offset, c_id, rg.height);
                    FL_range(t, rg, &next_ranges);                                          /* :95-96 */
                }
                range_vec tmp = FL_ranges;
                FL_ranges = next_ranges;
                next_ranges = tmp;
                skip_non_tunnel = !mode_all && FL_ranges.n > 1;                             /* :99 */
            }
            ++c;
            ++c_id;
        }
        run_start += run_len;
    }
    free(FL_ranges.v);
    free(next_ranges.v);
}

typedef struct {
    uint64_t *bits;                                  /* mark_start_bv */
    uint64_t set_count;
    uint64_t *rank_blk;                              /* ones before every 64-bit word (rank support) */
    uint8_t *id;                                     /* marked_ids[].first  */
    uint16_t *height;                                /* marked_ids[].second */
    int mode_all;
} split_ctx;

static void collect_boundaries(void *vctx, uint64_t pos, uint64_t c_id, uint16_t height) {   /* :64-66, bitvec::set :201-204 */
    split_ctx *s = (split_ctx *)vctx;
    (void)c_id;
    (void)height;
    if (!((s->bits[pos >> 6] >> (pos & 63)) & 1)) ++s->set_count;
    s->bits[pos >> 6] |= 1ull << (pos & 63);
}

static uint64_t rank1(const split_ctx *s, uint64_t pos) {
    return s->rank_blk[pos >> 6] + (uint64_t)__builtin_popcountll(s->bits[pos >> 6] & ((1ull << (pos & 63)) - 1));
}

static void collect_ids(void *vctx, uint64_t pos, uint64_t c_id, uint16_t height) {          /* :116-129 */
    split_ctx *s = (split_ctx *)vctx;
    const uint64_t k = rank1(s, pos);
    if (s->mode_all && ((s->bits[pos >> 6] >> (pos & 63)) & 1)) {
        const uint8_t existing_id = s->id[k];
        const uint16_t existing_height = s->height[k];
        const uint64_t max_height = existing_height > height ? existing_height : height;
        const uint64_t max_id = (existing_height >= height) ? existing_id : c_id;
        s->id[k] = bin_id(max_id);
        s->height[k] = (uint16_t)max_height;
    } else {
        s->id[k] = bin_id(c_id);
        s->height[k] = height;
    }
}

/* min-heap of intervals ordered by (end, start): std::priority_queue<interval, vector, greater>
 * with interval::operator>, col_split.hpp:274-289 */
typedef struct {
    uint64_t start, end;
    uint8_t id;
} ival;
typedef struct {
    ival *v;
    uint64_t n, cap;
} heap_t;
static int ival_less(const ival *a, const ival *b) { return a->end < b->end || (a->end == b->end && a->start < b->start); }
static void heap_push(heap_t *h, ival x) {
    if (h->n == h->cap) {
        h->cap = h->cap ? 2 * h->cap : 64;
        h->v = (ival *)realloc(h->v, h->cap * sizeof(ival));
    }
    uint64_t i = h->n++;
    while (i > 0 && ival_less(&x, &h->v[(i - 1) / 2])) {
        h->v[i] = h->v[(i - 1) / 2];
        i = (i - 1) / 2;
    }
    h->v[i] = x;
}
static ival heap_pop(heap_t *h) {
    const ival top = h->v[0], last = h->v[--h->n];
    uint64_t i = 0;
    for (;;) {
        uint64_t c = 2 * i + 1;
        if (c >= h->n) break;
        if (c + 1 < h->n && ival_less(&h->v[c + 1], &h->v[c])) ++c;
        if (!ival_less(&h->v[c], &last)) break;
        h->v[i] = h->v[c];
        i = c;
    }
    if (h->n) h->v[i] = last;
    return top;
}

typedef struct {
    const oracle_fl *t;
    uint64_t *col_runs;                              /* output bit_vector(n, 0) */
    uint8_t *ids;                                    /* col_run_ids */
    uint64_t n_ids, ids_cap;
    uint64_t run_cursor, curr_bwt_pos;
    uint8_t last_id;
    heap_t open;
} sweep_t;

static uint64_t bwt_run_select(const oracle_fl *t, uint64_t k) { return k <= t->r ? t->L_head[k - 1] : t->n; }

static void add_col_run_id(sweep_t *w, uint8_t id) {                                        /* :253-255 */
    if (w->n_ids < w->ids_cap) w->ids[w->n_ids] = id;
    ++w->n_ids;
}
static void set_col_run(sweep_t *w, uint64_t pos) { w->col_runs[pos >> 6] |= 1ull << (pos & 63); }

static void update_bwt_pos(sweep_t *w, uint64_t idx, uint8_t id) {                          /* :296-308 */
    while (w->run_cursor <= w->t->r && w->curr_bwt_pos < idx) {
        set_col_run(w, w->curr_bwt_pos);
        add_col_run_id(w, w->last_id);
        ++w->run_cursor;
        w->curr_bwt_pos = bwt_run_select(w->t, w->run_cursor);
    }
    if (w->curr_bwt_pos == idx) {
        ++w->run_cursor;
        w->curr_bwt_pos = bwt_run_select(w->t, w->run_cursor);
    }
    w->last_id = id;
}

static void update_col_ranges(sweep_t *w, uint64_t idx) {                                   /* :310-325 */
    while (w->open.n && w->open.v[0].end <= idx) {
        const ival e = heap_pop(&w->open);
        if (w->open.n == 1 && w->open.v[0].end > e.end) {
            update_bwt_pos(w, e.end, w->open.v[0].id);
            set_col_run(w, e.end);
            add_col_run_id(w, w->open.v[0].id);
        } else if (w->open.n == 0 && e.end < idx) {
            update_bwt_pos(w, e.end, 0);
            set_col_run(w, e.end);
            add_col_run_id(w, 0);
        }
    }
}

/* col_split::split (:54-136) + find_col_runs (:258-372) + the id folding of save (:138-157).
 * col_runs: ceil(n / 64) zeroed words; ids: up to ids_cap bytes; *n_ids receives the number of ids
 * the reference would write; stats = {col id runs, total runs (set bits), col chars}
 * (PRINT_STATS block, :339-371). */
void oracle_col_split(const oracle_fl *t, const uint64_t *col_len, const uint64_t *col_pos, uint64_t n_cols, uint32_t num_docs,
                      int mode_all, int split_rate, uint64_t *col_runs, uint8_t *ids, uint64_t ids_cap, uint64_t *n_ids,
                      uint64_t stats[3]) {
    const uint64_t n = t->n, words = (n + 63) / 64;
    const uint16_t N = (uint16_t)num_docs;            /* split(..., len_t N, ...) :56 */
    split_ctx s;
    memset(&s, 0, sizeof(s));
    s.mode_all = mode_all;
    s.bits = (uint64_t *)calloc(words + 1, sizeof(uint64_t));
    FL_loop(t, col_len, col_pos, n_cols, N, mode_all, split_rate, collect_boundaries, &s);      /* :111-113 */
    s.rank_blk = (uint64_t *)malloc((words + 1) * sizeof(uint64_t));
    uint64_t acc = 0;
    for (uint64_t wd = 0; wd <= words; ++wd) {
        s.rank_blk[wd] = acc;
        if (wd < words) acc += (uint64_t)__builtin_popcountll(s.bits[wd]);
    }
    s.id = (uint8_t *)calloc(s.set_count + 1, 1);                                               /* :116 {0, 0} */
    s.height = (uint16_t *)calloc(s.set_count + 1, sizeof(uint16_t));
    FL_loop(t, col_len, col_pos, n_cols, N, mode_all, split_rate, collect_ids, &s);             /* :131-134 */

    *n_ids = 0;
    memset(stats, 0, 3 * sizeof(uint64_t));
    if (s.set_count != 0) {                                                                     /* :259-261 */
        sweep_t w;
        memset(&w, 0, sizeof(w));
        w.t = t;
        w.col_runs = col_runs;
        w.ids = ids;
        w.ids_cap = ids_cap;
        w.run_cursor = 1;                                                                       /* :291-293 */
        w.curr_bwt_pos = bwt_run_select(t, 1);
        w.last_id = 0;
        uint64_t rank = 0;
        for (uint64_t wd = 0; wd < words; ++wd) {                                               /* :328-340 start_select(i), i = 1 .. set_bits */
            uint64_t m = s.bits[wd];
            while (m) {
                const uint64_t curr_start = (wd << 6) + (uint64_t)__builtin_ctzll(m);
                m &= m - 1;
                const uint8_t curr_col_id = s.id[rank];
                const uint16_t curr_col_height = s.height[rank];
                ++rank;
                update_col_ranges(&w, curr_start);
                heap_push(&w.open, (ival){curr_start, curr_start + curr_col_height, curr_col_id});
                if (w.open.n == 1 && curr_col_id > 0) {
                    update_bwt_pos(&w, curr_start, curr_col_id);
                    set_col_run(&w, curr_start);
                    add_col_run_id(&w, curr_col_id);
                }
            }
        }
        update_col_ranges(&w, n);                                                               /* :341-342 */
        update_bwt_pos(&w, n, 0);
        free(w.open.v);
        *n_ids = w.n_ids;
        /* PRINT_STATS, :345-371 */
        uint64_t set_bits = 0, last_idx = 0, k = 0, col_chars = 0, col_id_runs = 0;
        uint8_t last = 0;
        int first = 1;
        for (uint64_t wd = 0; wd < words; ++wd) {
            uint64_t m = col_runs[wd];
            while (m) {
                const uint64_t curr_idx = (wd << 6) + (uint64_t)__builtin_ctzll(m);
                m &= m - 1;
                const uint8_t curr_id = k < ids_cap ? ids[k] : 0;
                if (first) {                         /* i = 1: last_idx = select(1), last_id = ids[0]; the loop body sees them equal */
                    last_idx = curr_idx;
                    last = curr_id;
                    first = 0;
                }
                if (last >= 1) {
                    ++col_id_runs;
                    col_chars += curr_idx - last_idx;
                }
                last_idx = curr_idx;
                last = curr_id;
                ++k;
                ++set_bits;
            }
        }
        if (last >= 1) {
            col_chars += n - last_idx;
            ++col_id_runs;
        }
        stats[0] = col_id_runs;
        stats[1] = set_bits;
        stats[2] = col_chars;
    }
    free(s.bits);
    free(s.rank_blk);
    free(s.id);
    free(s.height);
}
