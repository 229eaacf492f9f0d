// s2_layout.h -- the "two-step" HBM layout of the run table.
//
// Why: every LF landing costs one 128-byte HBM line fill for a 16-byte row
// (DESIGN.md 4.1), and the fills are the roofline.  The query is a pure
// function of BWT positions, so rows may be split
// further without changing any output.  Here each row of the on-disk table is
// split at the pre-images of row boundaries under LF, so that ALL positions of
// a refined row land, after one LF step, in the SAME original row.  The
// character and col id the NEXT base will be compared against / will report
// (col_bwt.hpp:513-516 one iteration later) are then constants of the refined
// row and are stored in it, together with the landing of LF o LF.  When the next
// read base matches that character the lane emits two bases and jumps two LF
// steps with ONE line fill; otherwise it falls back to the one-step jump.
// HBM capacity (288 GB) is traded for fewer dependent line fills: at most 2r
// (+ cuts of rows longer than 65534) refined rows of 24 bytes, 5 per line.
//
// Row (24 bytes, 8-byte aligned, 5 rows per 128-byte line, last 8 bytes unused):
//   d0  I1     refined row holding LF(first position of the row)
//   d1  I2     refined row holding LF(LF(first position))
//   d2  O1 | O2 << 16          offsets of those images inside I1 / I2
//   d3  len16 | char << 16 | col_id << 24        (len <= 65534 by construction)
//   d4  len8 of row j+1 | len8 of row j+2 << 8 | mismatch-target distances << 16
//       (as .z of the one-step layout, device_layout.h)
//   d5  (spare 8) | char2 << 8 | col_id2 << 16 | hints << 24
//       char2 / col_id2 = character / col id of the original row every position
//       of this row maps into (LF_row::character, col_row::col_id of that row).
// idx2[r2+1]: first BWT position of each refined row (cold; idx2[r2] = n).
// thr2[r2]: the BWT run's threshold per refined row (compare-hints only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"

namespace colbwt {

constexpr uint32_t kS2RowsPerLine = 5;
constexpr uint32_t kS2RowBytes = 24;
constexpr uint32_t kS2BlockLines = 64;                                // jump block = 64 lines
constexpr uint32_t kS2BlockRows = kS2BlockLines * kS2RowsPerLine;     // 320 rows
constexpr uint32_t kS2MaxLen = 65534;                                 // longer rows are cut (legal by B.3)

struct S2Table {
    const uint8_t *lines;     // ceil((r2 + 1) / 5) + 1 lines of 128 bytes; row r2 is a sentinel
    const uint64_t *idx;      // r2 + 1
    const uint64_t *thr;      // r2
    const uint32_t *next_tbl; // nblk * sigma : first row >= b * 320 holding c
    const uint32_t *prev_tbl; // nblk * sigma : last row < b * 320 holding c
    const uint8_t *cmap;      // 256
    uint64_t n;
    uint32_t r2;              // refined rows
    uint32_t sigma;
    uint32_t nblk;
    uint32_t pad_;
};

struct S2Row {  // register image of one row
    uint32_t d[6];
};

__device__ __forceinline__ uint64_t s2_row_off(uint32_t j) {
    const uint32_t line = j / kS2RowsPerLine;
    return (uint64_t)line * 128u + (uint64_t)(j - line * kS2RowsPerLine) * kS2RowBytes;
}
__device__ __forceinline__ S2Row s2_load(const S2Table &T, uint32_t j) {
    const uint2 *p = reinterpret_cast<const uint2 *>(T.lines + s2_row_off(j));
    const uint2 a = p[0], b = p[1], c = p[2];
    S2Row w;
    w.d[0] = a.x; w.d[1] = a.y; w.d[2] = b.x; w.d[3] = b.y; w.d[4] = c.x; w.d[5] = c.y;
    return w;
}
__device__ __forceinline__ uint32_t s2_i1(const S2Row &w) { return w.d[0]; }
__device__ __forceinline__ uint32_t s2_i2(const S2Row &w) { return w.d[1]; }
__device__ __forceinline__ uint32_t s2_o1(const S2Row &w) { return w.d[2] & 0xFFFFu; }
__device__ __forceinline__ uint32_t s2_o2(const S2Row &w) { return w.d[2] >> 16; }
__device__ __forceinline__ uint32_t s2_len(const S2Row &w) { return w.d[3] & 0xFFFFu; }
__device__ __forceinline__ uint32_t s2_char(const S2Row &w) { return (w.d[3] >> 16) & 0xFFu; }
__device__ __forceinline__ uint32_t s2_cid(const S2Row &w) { return w.d[3] >> 24; }
__device__ __forceinline__ uint32_t s2_len8_next1(const S2Row &w) { return w.d[4] & 0xFFu; }
__device__ __forceinline__ uint32_t s2_len8_next2(const S2Row &w) { return (w.d[4] >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t s2_dist(const S2Row &w, uint32_t slot) { return (w.d[4] >> (16 + 4 * slot)) & 0xFu; }
__device__ __forceinline__ uint32_t s2_char2(const S2Row &w) { return (w.d[5] >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t s2_cid2(const S2Row &w) { return (w.d[5] >> 16) & 0xFFu; }
__device__ __forceinline__ uint32_t s2_hints(const S2Row &w) { return w.d[5] >> 24; }

// d3 dwords (len | char | cid) of the 5 rows of one line: 5 independent loads off one base.
__device__ __forceinline__ void s2_line_chars(const S2Table &T, uint32_t line, uint32_t (&ch)[kS2RowsPerLine]) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(T.lines + (uint64_t)line * 128u) + 3;
#pragma unroll
    for (uint32_t q = 0; q < kS2RowsPerLine; ++q) ch[q] = (p[q * 6] >> 16) & 0xFFu;
}

// succ_char (LF_table.hpp:286-298) over refined rows: smallest row > i holding c.
__device__ __forceinline__ uint32_t s2_succ_char(const S2Table &T, uint32_t i, uint32_t c, uint32_t cidx, S2Row &ws) {
    const uint32_t blk = i / kS2BlockRows;
    const uint64_t lim64 = ((uint64_t)blk + 1) * kS2BlockRows - 1;
    const uint32_t last = lim64 < (uint64_t)(T.r2 - 1) ? (uint32_t)lim64 : T.r2 - 1;
    for (uint64_t s0 = (uint64_t)i + 1; s0 <= last;) {
        const uint32_t line = (uint32_t)s0 / kS2RowsPerLine;
        const uint32_t lb = line * kS2RowsPerLine;
        const uint32_t lo_q = (uint32_t)s0 - lb;
        const uint32_t hi_q = (lb + kS2RowsPerLine - 1 < last ? lb + kS2RowsPerLine - 1 : last) - lb;
        uint32_t ch[kS2RowsPerLine];
        s2_line_chars(T, line, ch);
        uint32_t hit = kS2RowsPerLine;
#pragma unroll
        for (uint32_t q = kS2RowsPerLine; q-- > 0;) hit = (ch[q] == c && q >= lo_q && q <= hi_q) ? q : hit;
        if (hit < kS2RowsPerLine) {
            ws = s2_load(T, lb + hit);
            return lb + hit;
        }
        s0 = (uint64_t)lb + kS2RowsPerLine;
    }
    if (blk + 1 < T.nblk) {
        const uint32_t s = T.next_tbl[(uint64_t)(blk + 1) * T.sigma + cidx];
        if (s != kNone) ws = s2_load(T, s);
        return s;
    }
    return kNone;
}

// pred_char (LF_table.hpp:271-283) over refined rows: largest row < i holding c.
__device__ __forceinline__ uint32_t s2_pred_char(const S2Table &T, uint32_t i, uint32_t c, uint32_t cidx, S2Row &wq) {
    const uint32_t blk = i / kS2BlockRows;
    const uint32_t first = blk * kS2BlockRows;   // a multiple of 5: blocks start on a line
    for (uint32_t q0 = i; q0 > first;) {
        const uint32_t top = q0 - 1;
        const uint32_t line = top / kS2RowsPerLine;
        const uint32_t lb = line * kS2RowsPerLine;
        const uint32_t hi_q = top - lb;
        uint32_t ch[kS2RowsPerLine];
        s2_line_chars(T, line, ch);
        uint32_t hit = kS2RowsPerLine;
#pragma unroll
        for (uint32_t q = 0; q < kS2RowsPerLine; ++q) hit = (ch[q] == c && q <= hi_q) ? q : hit;
        if (hit < kS2RowsPerLine) {
            wq = s2_load(T, lb + hit);
            return lb + hit;
        }
        q0 = lb;
    }
    if (blk > 0) {
        const uint32_t q = T.prev_tbl[(uint64_t)blk * T.sigma + cidx];
        if (q != kNone) wq = s2_load(T, q);
        return q;
    }
    return kNone;
}

}  // namespace colbwt
