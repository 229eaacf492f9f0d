/*
 * colsplit_oracle.h -- CPU restatement of build_FL + col_split (see colsplit_oracle.c).
 * TEST INFRASTRUCTURE ONLY; parity pinned by the SURVEY.md Appendix C.6 counts alone.
 */
#ifndef COLSPLIT_ORACLE_H
#define COLSPLIT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_fl {      /* FL_table, include/ds/FL_table.hpp:38-81 (FL_row), :338-341 */
    uint64_t n, r;
    uint8_t *ch;                /* r  : character of F run k */
    uint64_t *idx;              /* r+1: first F position of run k (idx[r] = n) */
    uint64_t *interval;         /* r  : F run holding the L position of run k's first character */
    uint16_t *offset;           /* r  : its offset there (16 bits kept, as the bit-field does) */
    uint64_t *L_head;           /* r+1: start of every L (BWT) run: the ones of L_heads */
} oracle_fl;

int oracle_fl_build(const uint8_t *heads, uint64_t heads_len, const uint64_t *lens, oracle_fl *out);
void oracle_fl_free(oracle_fl *t);
void oracle_col_split(const oracle_fl *t, const uint64_t *col_len, const uint64_t *col_pos, uint64_t n_cols, uint32_t num_docs,
                      int mode_all, int split_rate, uint64_t *col_runs, uint8_t *ids, uint64_t ids_cap, uint64_t *n_ids,
                      uint64_t stats[3]);

#ifdef __cplusplus
}
#endif
#endif
