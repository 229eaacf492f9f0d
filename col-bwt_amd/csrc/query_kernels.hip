// query_kernels.hip -- the PML / col-ID query on gfx950 (MI355X), wave64.
//
// One lane per read; each lane walks its read right to left and carries the
// reference's whole per-base state (interval, offset, PML length) in
// registers.  Reference semantics, step for step:
//   col_pml::_query_pml     col_bwt.hpp:498-529   (extend / reset, col id)
//   col_pml::threshold_step col_bwt.hpp:531-574   (mismatch re-orientation)
//   LF_table::succ_char / pred_char  LF_table.hpp:271-298
//   LF_table::LF / LF_idx   LF_table.hpp:251-268  (LF step + fast-forward)
//
// Memory behaviour (HBM-bound integer pointer chasing, no MFMA):
//   * the only random access of a step is the 16-byte row the LF step lands
//     on (device_layout.h); that one load also carries char, col id, the next
//     interval/offset, idx (for pos) and the run length, so the following
//     step needs no further table access unless it fast-forwards / mismatches;
//   * read bytes are pulled 16 at a time through a 128-bit shift register;
//   * PML (u16) and col id (u8) are collected for 8 bases in registers and
//     leave as one 16-byte and one 8-byte aligned store.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"
#include "query_kernels.h"

namespace colbwt {

__device__ __forceinline__ uint32_t row_interval(const uint4 &w) { return w.x; }
__device__ __forceinline__ uint32_t row_offset(const uint4 &w) { return w.y & 0xFFFFu; }
__device__ __forceinline__ uint32_t row_len16(const uint4 &w) { return w.y >> 16; }
__device__ __forceinline__ uint64_t row_idx(const uint4 &w) {
    return (uint64_t)w.z | ((uint64_t)(w.w & 0xFFu) << 32);
}
__device__ __forceinline__ uint32_t row_char(const uint4 &w) { return (w.w >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t row_cid(const uint4 &w) { return (w.w >> 16) & 0xFFu; }

// LF_table::get_length (LF_table.hpp:204-207); the sentinel row r (idx = n)
// removes the last-row special case.
__device__ __forceinline__ uint64_t row_len(const DevTable &T, uint32_t j, const uint4 &w) {
    uint32_t l16 = row_len16(w);
    if (__builtin_expect(l16 != kLenLong, 1)) return l16;
    uint4 nx = T.rows[(uint64_t)j + 1];
    return row_idx(nx) - row_idx(w);
}

// LF_table::succ_char (LF_table.hpp:286-298) from run i whose char != c:
// smallest run > i holding c.  Linear scan inside the 256-row block, then one
// jump-table lookup.  Returns kNone when the scan would pass run r-1.
__device__ __forceinline__ uint32_t succ_char(const DevTable &T, uint32_t i, uint32_t c,
                                              uint32_t cidx, uint4 &ws) {
    const uint32_t blk = i >> kBlockShift;
    uint64_t lim64 = (((uint64_t)blk + 1) << kBlockShift) - 1;
    const uint32_t last = lim64 < (uint64_t)(T.r - 1) ? (uint32_t)lim64 : T.r - 1;
    for (uint32_t s = i; s < last;) {
        ++s;
        ws = T.rows[s];
        if (row_char(ws) == c) return s;
    }
    if (blk + 1 < T.nblk) {
        uint32_t s = T.next_tbl[(uint64_t)(blk + 1) * T.sigma + cidx];
        if (s != kNone) ws = T.rows[s];
        return s;
    }
    return kNone;
}

// LF_table::pred_char (LF_table.hpp:271-283): largest run < i holding c.
__device__ __forceinline__ uint32_t pred_char(const DevTable &T, uint32_t i, uint32_t c,
                                              uint32_t cidx, uint4 &wq) {
    const uint32_t blk = i >> kBlockShift;
    const uint32_t first = blk << kBlockShift;
    for (uint32_t q = i; q > first;) {
        --q;
        wq = T.rows[q];
        if (row_char(wq) == c) return q;
    }
    if (blk > 0) {
        uint32_t q = T.prev_tbl[(uint64_t)blk * T.sigma + cidx];
        if (q != kNone) wq = T.rows[q];
        return q;
    }
    return kNone;
}

// col_pml::threshold_step (col_bwt.hpp:531-574), the !MULTI_THREAD arm: pred
// is only searched when pos < thr (same result as the threaded arm).
__device__ __forceinline__ void threshold_step(const DevTable &T, const uint8_t *s_cmap,
                                               uint32_t &i, uint64_t &o, uint4 &w, uint32_t c) {
    const uint32_t cidx = s_cmap[c];
    if (cidx == kAbsent) return;  // c occurs nowhere: (interval, offset) unchanged (:533-534)
    const uint64_t pos = row_idx(w) + o;  // LF_table::to_idx (LF_table.hpp:214-217)
    uint64_t thr = T.n;                   // :535
    uint32_t ni = i;
    uint64_t no = o;
    uint4 nw = w;
    uint4 t;
    const uint32_t s = succ_char(T, i, c, cidx, t);  // :548
    if (s != kNone) {                                 // :552-557
        thr = T.thr[s];
        ni = s;
        no = 0;
        nw = t;
    }
    if (pos < thr) {                                  // :560
        const uint32_t q = pred_char(T, i, c, cidx, t);  // :562
        if (q != kNone) {                             // :565-569
            ni = q;
            no = row_len(T, q, t) - 1;                // LF_table.hpp:282
            nw = t;
        }
    }
    i = ni;                                           // :572-573
    o = no;
    w = nw;
}

// 128-bit shift register holding up to 16 read bytes; the next byte to
// consume (highest address) sits in the top byte.
struct ReadWindow {
    uint64_t lo, hi;
    __device__ __forceinline__ void load(const uint8_t *bases, uint64_t g) {
        const uint4 v = *reinterpret_cast<const uint4 *>(bases + (g & ~(uint64_t)15));
        lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
        hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
    __device__ __forceinline__ void drop_top(uint32_t nbytes) {  // nbytes in [0,15]
        uint32_t sh = nbytes * 8;
        if (sh >= 64) { hi = lo; lo = 0; sh -= 64; }
        if (sh) { hi = (hi << sh) | (lo >> (64 - sh)); lo <<= sh; }
    }
    __device__ __forceinline__ uint32_t pop() {
        uint32_t c = (uint32_t)(hi >> 56);
        hi = (hi << 8) | (lo >> 56);
        lo <<= 8;
        return c;
    }
};

// Output collector: 8 bases per flush, flush boundaries at global element
// indices that are multiples of 8 so full flushes are aligned vector stores.
template <typename PmlT>
struct OutAcc;

template <>
struct OutAcc<uint16_t> {
    uint64_t plo = 0, phi = 0, c8 = 0;
    uint32_t cnt = 0;
    __device__ __forceinline__ void push(uint32_t L, uint32_t cid) {
        phi = (phi << 16) | (plo >> 48);
        plo = (plo << 16) | (uint64_t)(L & 0xFFFFu);
        c8 = (c8 << 8) | (uint64_t)cid;
        ++cnt;
    }
    __device__ __forceinline__ void flush(uint16_t *pml, uint8_t *cid, uint64_t g) {
        if (cnt == 8) {
            uint4 v = make_uint4((uint32_t)plo, (uint32_t)(plo >> 32), (uint32_t)phi, (uint32_t)(phi >> 32));
            *reinterpret_cast<uint4 *>(pml + g) = v;
            *reinterpret_cast<uint2 *>(cid + g) = make_uint2((uint32_t)c8, (uint32_t)(c8 >> 32));
        } else {
            for (uint32_t e = 0; e < cnt; ++e) {
                pml[g + e] = (uint16_t)plo;
                cid[g + e] = (uint8_t)c8;
                plo = (plo >> 16) | (phi << 48);
                phi >>= 16;
                c8 >>= 8;
            }
        }
        cnt = 0;
    }
};

template <>
struct OutAcc<uint32_t> {  // reads longer than 65535 bases: wide PML, same cadence
    uint64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, c8 = 0;
    uint32_t cnt = 0;
    __device__ __forceinline__ void push(uint32_t L, uint32_t cid) {
        p3 = (p3 << 32) | (p2 >> 32);
        p2 = (p2 << 32) | (p1 >> 32);
        p1 = (p1 << 32) | (p0 >> 32);
        p0 = (p0 << 32) | (uint64_t)L;
        c8 = (c8 << 8) | (uint64_t)cid;
        ++cnt;
    }
    __device__ __forceinline__ void flush(uint32_t *pml, uint8_t *cid, uint64_t g) {
        if (cnt == 8) {
            uint4 *dst = reinterpret_cast<uint4 *>(pml + g);
            dst[0] = make_uint4((uint32_t)p0, (uint32_t)(p0 >> 32), (uint32_t)p1, (uint32_t)(p1 >> 32));
            dst[1] = make_uint4((uint32_t)p2, (uint32_t)(p2 >> 32), (uint32_t)p3, (uint32_t)(p3 >> 32));
            *reinterpret_cast<uint2 *>(cid + g) = make_uint2((uint32_t)c8, (uint32_t)(c8 >> 32));
        } else {
            for (uint32_t e = 0; e < cnt; ++e) {
                pml[g + e] = (uint32_t)p0;
                cid[g + e] = (uint8_t)c8;
                p0 = (p0 >> 32) | (p1 << 32);
                p1 = (p1 >> 32) | (p2 << 32);
                p2 = (p2 >> 32) | (p3 << 32);
                p3 >>= 32;
                c8 >>= 8;
            }
        }
        cnt = 0;
    }
};

template <typename PmlT>
__global__ __launch_bounds__(kQueryBlock) void pml_query_kernel(DevTable T, const uint8_t *__restrict__ bases,
                                                                const uint64_t *__restrict__ read_off,
                                                                uint64_t n_reads, PmlT *__restrict__ pml,
                                                                uint8_t *__restrict__ cid) {
    __shared__ uint8_t s_cmap[256];
    for (uint32_t t = threadIdx.x; t < 256; t += kQueryBlock) s_cmap[t] = T.cmap[t];
    __syncthreads();

    const uint64_t rd = (uint64_t)blockIdx.x * kQueryBlock + threadIdx.x;
    if (rd >= n_reads) return;
    const uint64_t off = read_off[rd];
    const uint64_t m = read_off[rd + 1] - off;
    if (m == 0) return;

    // col_bwt.hpp:503-508: pos = n-1, interval = r-1, offset = len(r-1)-1, length = 0
    uint32_t i = T.r - 1;
    uint4 w = T.rows[i];
    uint64_t o = row_len(T, i, w) - 1;
    uint32_t L = 0;

    ReadWindow win;
    OutAcc<PmlT> acc;
    {
        const uint64_t g = off + m - 1;
        win.load(bases, g);
        win.drop_top(15u - (uint32_t)(g & 15));
    }

    for (uint64_t k = m; k-- > 0;) {                 // col_bwt.hpp:510, i = m-1-k
        const uint64_t g = off + k;
        const uint32_t c = win.pop();                // :512 pattern[m-i-1], raw byte
        const uint32_t col_id = row_cid(w);          // :513 before any re-orientation
        if (row_char(w) == c) {                      // :516
            ++L;                                     // :517
        } else {
            L = 0;                                   // :521
            threshold_step(T, s_cmap, i, o, w, c);   // :522
        }
        acc.push(L, col_id);                         // :525
        if ((g & 7) == 0 || k == 0) acc.flush(pml, cid, g);
        if (k == 0) break;  // the reference's last LF (col_bwt.hpp:527) has no observable effect
        if ((g & 15) == 0) win.load(bases, g - 1);

        // LF_table::LF (LF_table.hpp:251-262)
        uint32_t j = row_interval(w);                // :253
        uint64_t t = (uint64_t)row_offset(w) + o;    // :254
        w = T.rows[j];
        uint64_t len = row_len(T, j, w);
        while (t >= len && j < T.r - 1) {            // :256 (bounded: a validated table never passes r-1)
            t -= len;                                // :258
            ++j;
            w = T.rows[j];
            len = row_len(T, j, w);
        }
        i = j;
        o = t;
    }
}

void launch_pml_query(const DevTable &T, const uint8_t *d_bases, const uint64_t *d_read_off,
                      uint64_t n_reads, void *d_pml, int pml_bytes, uint8_t *d_cid, hipStream_t stream) {
    if (n_reads == 0) return;
    const uint64_t blocks = (n_reads + kQueryBlock - 1) / kQueryBlock;
    dim3 grid((uint32_t)blocks), block(kQueryBlock);
    if (pml_bytes == 2)
        hipLaunchKernelGGL(pml_query_kernel<uint16_t>, grid, block, 0, stream, T, d_bases, d_read_off, n_reads,
                           (uint16_t *)d_pml, d_cid);
    else
        hipLaunchKernelGGL(pml_query_kernel<uint32_t>, grid, block, 0, stream, T, d_bases, d_read_off, n_reads,
                           (uint32_t *)d_pml, d_cid);
}

}  // namespace colbwt
