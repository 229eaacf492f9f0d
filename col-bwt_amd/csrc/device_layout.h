// device_layout.h -- HBM layout of the col-bwt run table on MI355X.
//
// The on-disk row (col_thr, 18 packed bytes: LF_table.hpp:33-40,
// col_bwt.hpp:43,84) straddles 64-byte lines and needs
// the NEXT row's idx for every run length (LF_table.hpp:204-207).  The query
// is a pure function of BWT positions, so the table
// is re-laid out once at load time:
//
//   rows[r+1]  16-byte aligned rows, one 128-bit load per LF landing:
//       .x  interval                      (32)   LF_row::interval
//       .y  offset | len16 << 16          (16+16) LF_row::offset ; run length,
//                                          0xFFFF = "long": idx[j+1]-idx[j]
//       .z  cut_a | len_b << 8 (8 bits each, 0xFF = none): the LF image of this row starts
//           at offset `offset` of row `interval`; positions at offsets >= cut_a =
//           len(interval) - offset fall into row interval + 1 (at offset - cut_a), and
//           those at >= cut_a + len_b (len_b = len(interval + 1)) into row interval + 2:
//           the first steps of the fast-forward loop (LF_table.hpp:256-259) are taken
//           before the landing row is loaded, not with extra dependent loads after it;
//           | dist << 16: 4 bits per hint slot, the distance in rows to the run a
//           mismatch on that character re-orients to (the predecessor when the
//           hint says pred-side, else the successor; 15 = scan for it): a
//           mismatch then costs one row load instead of a scan plus a row load
//       .w  (spare 8) | char << 8 | col_id << 16 | hints << 24
//     hints: 2 bits for each of the 4 most frequent OTHER characters c (dense
//     character indices are ordered by frequency), precomputed at
//     load: how `pos < threshold(succ_c(row))` (col_bwt.hpp:560) comes out for
//     EVERY offset inside the row -- kHintPred (always true, or no successor),
//     kHintSucc (always false) or kHintCompare (the threshold falls inside the
//     row: compare at query time).  Saves the threshold load and one of the two
//     scans on almost every mismatch.
//     row r is a sentinel with idx = n, so len(j) = idx[j+1]-idx[j] holds for
//     the last row too (LF_table.hpp:206 special-cases it).
//   idx[r+1]   u64 BWT position of each row's first character (LF_row::idx;
//              idx[r] = n).  COLD: only a compare-hint mismatch (pos), a long
//              run's length and the load-time kernels read it.
//   thr[r]     u64 thresholds (col_thr::threshold), compare-hint mismatches only.
//   next_tbl / prev_tbl  [nblk][sigma] u32: first run >= b*B / last run < b*B
//     holding each present character; bound the succ_char / pred_char scans
//     (LF_table.hpp:271-298 are unbounded linear scans) to one block of B rows.
//   cmap[256]  byte -> dense character index, 0xFF = byte absent from the BWT
//     (then neither scan can succeed: col_bwt.hpp:533-534,572-573).
#pragma once
#include <stdint.h>

#include "disk_format.h"

namespace colbwt {

constexpr uint32_t kLenLong = 0xFFFFu;      // len16 escape
constexpr uint32_t kNone = 0xFFFFFFFFu;     // "no such run" in jump tables
constexpr uint32_t kBlockShift = 8;         // B = 256 rows per jump block
constexpr uint32_t kAbsent = 0xFFu;         // cmap: byte not in the BWT
constexpr uint32_t kAlgBytesPerBase = 27;
constexpr uint32_t kHintPred = 0, kHintSucc = 1, kHintCompare = 2;
constexpr uint32_t kHintAllCompare = 0xAAu; // every slot = kHintCompare
constexpr uint32_t kCutNone = 0xFFu;        // .z cut escape
constexpr uint32_t kDistFar = 15u;          // .z distance nibble: target not within 14 rows / unknown
constexpr uint32_t kHintSlots = 4;          // 2 bits each in the row's spare byte
constexpr uint32_t kHintMaxSigma = 5;       // characters that can own a slot: the 5 most frequent

struct DevTable {
    const uint4 *rows;        // r + 1 (padded to whole lines)
    const uint64_t *idx;      // r + 1
    const uint64_t *thr;      // r
    const uint32_t *next_tbl; // nblk * sigma
    const uint32_t *prev_tbl; // nblk * sigma
    const uint8_t *cmap;      // 256
    uint64_t n;
    uint32_t r;
    uint32_t sigma;
    uint32_t nblk;
    uint32_t use_hints;       // 1 once hint_kernel has run (always, after load)
};

}  // namespace colbwt
