// lf_device.h -- device-side accessors of the HBM row layout (device_layout.h)
// and the two bounded scans of the reference's move structure:
//   LF_table::get_length  LF_table.hpp:204-207
//   LF_table::succ_char   LF_table.hpp:286-298
//   LF_table::pred_char   LF_table.hpp:271-283
// Shared by the query kernel and the load-time hint kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"

namespace colbwt {

__device__ __forceinline__ uint32_t row_interval(const uint4 &w) { return w.x; }
__device__ __forceinline__ uint32_t row_offset(const uint4 &w) { return w.y & 0xFFFFu; }
__device__ __forceinline__ uint32_t row_len16(const uint4 &w) { return w.y >> 16; }
__device__ __forceinline__ uint32_t row_cut_a(const uint4 &w) { return w.z & 0xFFu; }
__device__ __forceinline__ uint32_t row_len_b(const uint4 &w) { return (w.z >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t row_dist(const uint4 &w, uint32_t slot) { return (w.z >> (16 + 4 * slot)) & 0xFu; }
__device__ __forceinline__ uint64_t row_idx(const DevTable &T, uint32_t j) { return T.idx[j]; }
__device__ __forceinline__ uint32_t row_char(const uint4 &w) { return (w.w >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t row_cid(const uint4 &w) { return (w.w >> 16) & 0xFFu; }
__device__ __forceinline__ uint32_t row_hints(const uint4 &w) { return w.w >> 24; }

// LF_table::get_length; the sentinel row r (idx = n) removes the last-row special case.
__device__ __forceinline__ uint64_t row_len(const DevTable &T, uint32_t j, const uint4 &w) {
    const uint32_t l16 = row_len16(w);
    if (__builtin_expect(l16 != kLenLong, 1)) return l16;
    return T.idx[(uint64_t)j + 1] - T.idx[j];
}

// Rows are 16 bytes, 8 per 128-byte HBM line.  The scans below walk line by
// line: the characters (.w dwords) of all 8 rows of a line are fetched by 8
// independent loads off ONE base address (immediate offsets) and compared in
// registers, then the matching row is loaded whole.  Inside the line that
// already holds row i these are cache hits, so a mismatch usually costs no
// extra HBM line.  (Loads are unconditional: a predicated load is serialised
// behind its own branch + s_waitcnt.  The rows array is padded to whole lines.)
__device__ __forceinline__ void line_chars(const DevTable &T, uint32_t line_first_row, uint32_t (&ch)[8]) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(T.rows) + (uint64_t)line_first_row * 4 + 3;
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) ch[q] = (p[q * 4] >> 8) & 0xFFu;
}

// succ_char from run i whose char != c: smallest run > i holding c.  Scan
// inside the 256-row jump block, then one jump-table lookup.  kNone when the
// reference's scan (LF_table.hpp:286-298) would pass run r-1.
__device__ __forceinline__ uint32_t succ_char(const DevTable &T, uint32_t i, uint32_t c, uint32_t cidx, uint4 &ws) {
    const uint32_t blk = i >> kBlockShift;
    const uint64_t lim64 = (((uint64_t)blk + 1) << kBlockShift) - 1;
    const uint32_t last = lim64 < (uint64_t)(T.r - 1) ? (uint32_t)lim64 : T.r - 1;
    for (uint64_t s0 = (uint64_t)i + 1; s0 <= last;) {
        const uint32_t lb = (uint32_t)s0 & ~7u;
        const uint32_t lo_q = (uint32_t)s0 & 7u;
        const uint32_t hi_q = (lb + 7u < last ? lb + 7u : last) - lb;
        uint32_t ch[8];
        line_chars(T, lb, ch);
        uint32_t hit = 8;
#pragma unroll
        for (uint32_t q = 8; q-- > 0;) hit = (ch[q] == c && q >= lo_q && q <= hi_q) ? q : hit;  // lowest match
        if (hit < 8) {
            ws = T.rows[lb + hit];
            return lb + hit;
        }
        s0 = (uint64_t)lb + 8;
    }
    if (blk + 1 < T.nblk) {
        const uint32_t s = T.next_tbl[(uint64_t)(blk + 1) * T.sigma + cidx];
        if (s != kNone) ws = T.rows[s];
        return s;
    }
    return kNone;
}

// pred_char (LF_table.hpp:271-283): largest run < i holding c.
__device__ __forceinline__ uint32_t pred_char(const DevTable &T, uint32_t i, uint32_t c, uint32_t cidx, uint4 &wq) {
    const uint32_t blk = i >> kBlockShift;
    const uint32_t first = blk << kBlockShift;   // a multiple of 256, hence of 8
    for (uint32_t q0 = i; q0 > first;) {          // candidates are rows first .. q0-1
        const uint32_t top = q0 - 1;
        const uint32_t lb = top & ~7u;
        const uint32_t hi_q = top & 7u;
        uint32_t ch[8];
        line_chars(T, lb, ch);
        uint32_t hit = 8;
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) hit = (ch[q] == c && q <= hi_q) ? q : hit;  // highest match
        if (hit < 8) {
            wq = T.rows[lb + hit];
            return lb + hit;
        }
        q0 = lb;
    }
    if (blk > 0) {
        const uint32_t q = T.prev_tbl[(uint64_t)blk * T.sigma + cidx];
        if (q != kNone) wq = T.rows[q];
        return q;
    }
    return kNone;
}


// Slot of character index cidx among the characters other than the row's own
// (aidx): 2 hint bits per slot, 4 slots => usable when sigma <= 5.
__device__ __forceinline__ uint32_t hint_slot(uint32_t cidx, uint32_t aidx) { return cidx < aidx ? cidx : cidx - 1; }

}  // namespace colbwt
