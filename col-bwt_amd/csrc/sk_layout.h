// sk_layout.h -- "K-step" HBM layouts of the run table (K = 2, 3).
//
// Why: every LF landing costs one 128-byte HBM line fill and one dependent
// memory round trip for a 16-byte row.  The query is a pure
// function of BWT positions, so rows may be split
// further without changing any output.  Level K is obtained from level K-1 by
// splitting every row at the pre-images, under LF, of the level-(K-1) row
// boundaries.  By induction all positions of a level-K row then walk, for K-1
// LF steps, through the SAME original rows: the characters and col ids the
// next K-1 bases will be compared against / will report (col_bwt.hpp:513-516,
// one iteration later each) are constants of the row and are stored in it,
// together with the landings of LF, LF^2, .. LF^K.  While the next read bases
// keep matching those characters the lane emits them without touching memory
// and then jumps up to K LF steps with ONE row load.  HBM capacity (288 GB) is
// traded for fewer dependent line fills: at most K*r rows.
//
// Row dwords (8-byte aligned; K = 2: 24 bytes, 5 rows per 128-byte line;
// K = 3: 32 bytes, 4 per line):
//   I[1..K]   level-K row holding LF^s(first position of the row)
//   O[1..K]   16-bit offsets of those images inside I[s]
//   len16 | char << 16 | col_id << 24                  (len <= 65534 by construction)
//   cuts | 3 x 8-bit mismatch-target distances (hint slots 0..2) << 8
//             cuts = cut_a | len_b << 4 (4 bits each, 15 = none): the image of this row under
//             the deepest jump LF^K starts at offset O[K] of row I[K]; positions at offsets
//             >= cut_a = len(I[K]) - O[K] fall into row I[K] + 1 at offset - cut_a, and those
//             at >= len_b = len(I[K] + 1) there into row I[K] + 2.  Two thirds of the jumps are
//             the deepest, 4 in 10 of those leave their landing row, nearly all by one or two
//             short rows: these steps of the fast-forward (LF_table.hpp:256-259) cost no load.
//   distance of slot 3 | char2 << 8 | col_id2 << 16 | hints << 24 ;  K = 3: char3, col_id3
//             (distances as in the one-step layout, device_layout.h, but 8 bits
//             wide, 255 = scan: refined rows put the target run up to K times
//             more rows away, and a scan costs the wave several round trips)
//             char_s / col_id_s = character / col id of the original row every
//             position of this row is in after s-1 LF steps
// idx[r+1]: first BWT position of each row (cold; idx[r] = n).  thr[r]: the BWT
// run's threshold (compare-hints only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"

namespace colbwt {

constexpr uint32_t kSKMaxLen = 65534;   // longer rows are cut (legal by B.3)
constexpr uint32_t kSKDistFar = 255u;   // 8-bit distance escape
constexpr uint32_t kSKCutNone = 15u;    // 4-bit cut escape

template <int K>
struct SKGeom;
template <>
struct SKGeom<2> {
    static constexpr uint32_t kDwords = 6, kRowsPerLine = 5;
    static constexpr uint32_t kO = 2;      // dword of O1 | O2 << 16
    static constexpr uint32_t kLen = 3;    // len | char | cid ; kLen+1 = lens|dists ; kLen+2 = tags
};
template <>
struct SKGeom<3> {
    static constexpr uint32_t kDwords = 8, kRowsPerLine = 4;
    static constexpr uint32_t kO = 3;      // O1 | O2 << 16 ; dword 4 = O3 | char3 << 16 | cid3 << 24
    static constexpr uint32_t kLen = 5;
};
template <int K>
constexpr uint32_t sk_row_bytes() { return SKGeom<K>::kDwords * 4; }
template <int K>
constexpr uint32_t sk_block_rows() { return 64 * SKGeom<K>::kRowsPerLine; }   // jump block = 64 lines

struct SKTable {
    const uint8_t *lines;     // ceil((r + 1) / rows_per_line) + 1 lines of 128 bytes; row r is a sentinel
    const uint64_t *idx;      // r + 1
    const uint64_t *thr;      // r
    const uint32_t *next_tbl; // nblk * sigma : first row >= b * block_rows holding c
    const uint32_t *prev_tbl; // nblk * sigma : last row < b * block_rows holding c
    const uint8_t *cmap;      // 256
    uint64_t n;
    uint32_t r;               // rows of this level
    uint32_t sigma;
    uint32_t nblk;
    uint32_t steps;           // K
};

template <int K>
struct SKRow {  // register image of one row
    uint32_t d[SKGeom<K>::kDwords];
};

template <int K>
__device__ __forceinline__ uint64_t sk_row_off(uint32_t j) {
    constexpr uint32_t rpl = SKGeom<K>::kRowsPerLine;
    if constexpr (rpl == 4) {
        return (uint64_t)j * 32u;
    } else {
        const uint32_t line = j / rpl;
        return (uint64_t)line * 128u + (uint64_t)(j - line * rpl) * sk_row_bytes<K>();
    }
}
template <int K>
__device__ __forceinline__ SKRow<K> sk_load(const SKTable &T, uint32_t j) {
    SKRow<K> w;
    if constexpr (K == 2) {
        const uint2 *p = reinterpret_cast<const uint2 *>(T.lines + sk_row_off<2>(j));
        const uint2 a = p[0], b = p[1], c = p[2];
        w.d[0] = a.x; w.d[1] = a.y; w.d[2] = b.x; w.d[3] = b.y; w.d[4] = c.x; w.d[5] = c.y;
    } else {
        const uint4 *p = reinterpret_cast<const uint4 *>(T.lines + sk_row_off<3>(j));
        const uint4 a = p[0], b = p[1];
        w.d[0] = a.x; w.d[1] = a.y; w.d[2] = a.z; w.d[3] = a.w; w.d[4] = b.x; w.d[5] = b.y; w.d[6] = b.z; w.d[7] = b.w;
    }
    return w;
}

// landing of LF^s, s in [1, K] (s is run-time data: how many look-ahead characters matched)
template <int K>
__device__ __forceinline__ uint32_t sk_I(const SKRow<K> &w, uint32_t s) {
    if constexpr (K == 2) return s == 1 ? w.d[0] : w.d[1];
    else return s == 1 ? w.d[0] : (s == 2 ? w.d[1] : w.d[2]);
}
template <int K>
__device__ __forceinline__ uint32_t sk_O(const SKRow<K> &w, uint32_t s) {
    constexpr uint32_t o = SKGeom<K>::kO;
    if constexpr (K == 2) return s == 1 ? (w.d[o] & 0xFFFFu) : (w.d[o] >> 16);
    else return s == 1 ? (w.d[o] & 0xFFFFu) : (s == 2 ? (w.d[o] >> 16) : (w.d[o + 1] & 0xFFFFu));
}
template <int K>
__device__ __forceinline__ uint32_t sk_len(const SKRow<K> &w) { return w.d[SKGeom<K>::kLen] & 0xFFFFu; }
template <int K>
__device__ __forceinline__ uint32_t sk_char(const SKRow<K> &w) { return (w.d[SKGeom<K>::kLen] >> 16) & 0xFFu; }
template <int K>
__device__ __forceinline__ uint32_t sk_cid(const SKRow<K> &w) { return w.d[SKGeom<K>::kLen] >> 24; }
template <int K>
__device__ __forceinline__ uint32_t sk_cut_a(const SKRow<K> &w) { return w.d[SKGeom<K>::kLen + 1] & 0xFu; }
template <int K>
__device__ __forceinline__ uint32_t sk_len_b(const SKRow<K> &w) { return (w.d[SKGeom<K>::kLen + 1] >> 4) & 0xFu; }
template <int K>
__device__ __forceinline__ uint32_t sk_dist(const SKRow<K> &w, uint32_t slot) {   // slot in [0, 3]
    const uint32_t three = w.d[SKGeom<K>::kLen + 1] >> 8;                        // slots 0..2
    return slot < 3 ? (three >> (8 * slot)) & 0xFFu : w.d[SKGeom<K>::kLen + 2] & 0xFFu;
}
template <int K>
__device__ __forceinline__ uint32_t sk_hints(const SKRow<K> &w) { return w.d[SKGeom<K>::kLen + 2] >> 24; }
// character / col id met after a-1 LF steps, a in [2, K] (compile-time a)
template <int K, int A>
__device__ __forceinline__ uint32_t sk_char_at(const SKRow<K> &w) {
    static_assert(A >= 2 && A <= K, "look-ahead depth");
    if constexpr (A == 2) return (w.d[SKGeom<K>::kLen + 2] >> 8) & 0xFFu;
    else return (w.d[4] >> 16) & 0xFFu;
}
template <int K, int A>
__device__ __forceinline__ uint32_t sk_cid_at(const SKRow<K> &w) {
    if constexpr (A == 2) return (w.d[SKGeom<K>::kLen + 2] >> 16) & 0xFFu;
    else return w.d[4] >> 24;
}

// (len | char | cid) dwords of the rows of one line: independent loads off one base.
template <int K>
__device__ __forceinline__ void sk_line_chars(const SKTable &T, uint32_t line, uint32_t (&ch)[SKGeom<K>::kRowsPerLine]) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(T.lines + (uint64_t)line * 128u) + SKGeom<K>::kLen;
#pragma unroll
    for (uint32_t q = 0; q < SKGeom<K>::kRowsPerLine; ++q) ch[q] = (p[q * SKGeom<K>::kDwords] >> 16) & 0xFFu;
}

// succ_char (LF_table.hpp:286-298) over level-K rows: smallest row > i holding c.
template <int K>
__device__ __forceinline__ uint32_t sk_succ_char(const SKTable &T, uint32_t i, uint32_t c, uint32_t cidx, SKRow<K> &ws) {
    constexpr uint32_t rpl = SKGeom<K>::kRowsPerLine;
    const uint32_t blk = i / sk_block_rows<K>();
    const uint64_t lim64 = ((uint64_t)blk + 1) * sk_block_rows<K>() - 1;
    const uint32_t last = lim64 < (uint64_t)(T.r - 1) ? (uint32_t)lim64 : T.r - 1;
    for (uint64_t s0 = (uint64_t)i + 1; s0 <= last;) {
        const uint32_t line = (uint32_t)s0 / rpl;
        const uint32_t lb = line * rpl;
        const uint32_t lo_q = (uint32_t)s0 - lb;
        const uint32_t hi_q = (lb + rpl - 1 < last ? lb + rpl - 1 : last) - lb;
        uint32_t ch[rpl];
        sk_line_chars<K>(T, line, ch);
        uint32_t hit = rpl;
#pragma unroll
        for (uint32_t q = rpl; q-- > 0;) hit = (ch[q] == c && q >= lo_q && q <= hi_q) ? q : hit;
        if (hit < rpl) {
            ws = sk_load<K>(T, lb + hit);
            return lb + hit;
        }
        s0 = (uint64_t)lb + rpl;
    }
    if (blk + 1 < T.nblk) {
        const uint32_t s = T.next_tbl[(uint64_t)(blk + 1) * T.sigma + cidx];
        if (s != kNone) ws = sk_load<K>(T, s);
        return s;
    }
    return kNone;
}

// pred_char (LF_table.hpp:271-283) over level-K rows: largest row < i holding c.
template <int K>
__device__ __forceinline__ uint32_t sk_pred_char(const SKTable &T, uint32_t i, uint32_t c, uint32_t cidx, SKRow<K> &wq) {
    constexpr uint32_t rpl = SKGeom<K>::kRowsPerLine;
    const uint32_t blk = i / sk_block_rows<K>();
    const uint32_t first = blk * sk_block_rows<K>();   // blocks start on a line
    for (uint32_t q0 = i; q0 > first;) {
        const uint32_t top = q0 - 1;
        const uint32_t line = top / rpl;
        const uint32_t lb = line * rpl;
        const uint32_t hi_q = top - lb;
        uint32_t ch[rpl];
        sk_line_chars<K>(T, line, ch);
        uint32_t hit = rpl;
#pragma unroll
        for (uint32_t q = 0; q < rpl; ++q) hit = (ch[q] == c && q <= hi_q) ? q : hit;
        if (hit < rpl) {
            wq = sk_load<K>(T, lb + hit);
            return lb + hit;
        }
        q0 = lb;
    }
    if (blk > 0) {
        const uint32_t q = T.prev_tbl[(uint64_t)blk * T.sigma + cidx];
        if (q != kNone) wq = sk_load<K>(T, q);
        return q;
    }
    return kNone;
}

}  // namespace colbwt
