// refine.h -- the refinement pass shared by the K-step layouts (sk_build.hip) and the line-row
// layout (fat_build.hip): level L+1 = the rows of level L split at the pre-images, under LF, of
// the level-L row boundaries (legal: the query is a function of BWT positions).
//   count   per source row: number of new rows = pieces of its LF image between source-row
//           boundaries (+ cuts at thresholds in the first pass, + cuts at 65534 positions)
//   scan    exclusive prefix sum -> first new row of every source row
// (emit / link are layout specific).  Everything lives in an unnamed namespace: each
// translation unit that includes this header gets its own kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/colbwt.h"
#include "dev_mem.h"
#include "device_layout.h"
#include "lf_device.h"
#include "query_kernels.h"
#include "sk_layout.h"

namespace colbwt {

namespace {

// ---- source views: what a refinement pass needs from the level below ------
struct SrcL1 {  // the one-step table
    DevTable T;
    HintChars chars;
    static constexpr int kSteps = 1;
    // Positions strictly inside row i where a mismatch on one of the hinted characters
    // changes sides: the threshold of that character's next run (col_bwt.hpp:552-560).
    // The first refinement cuts there too, so that no refined row contains a threshold
    // and every hint is decided -- in a real index the threshold of a run lies between
    // the previous run of its character and its head, i.e. inside one of the rows in
    // between, and the query would otherwise fall back to scans + position compares.
    __device__ __forceinline__ uint32_t cuts(uint32_t i, uint64_t (&cut)[kHintSlots]) const {
        const uint4 w = T.rows[i];
        const uint32_t aidx = T.cmap[row_char(w)];
        const uint32_t top = T.sigma < kHintMaxSigma ? T.sigma : kHintMaxSigma;
        uint32_t nc = 0;
        for (uint32_t cidx = 0; cidx < top; ++cidx) {
            const uint32_t slot = hint_slot(cidx, aidx);
            if (cidx == aidx || slot >= kHintSlots) continue;
            if (((row_hints(w) >> (2 * slot)) & 3u) != kHintCompare) continue;
            uint4 t;
            const uint32_t s = succ_char(T, i, chars.c[cidx], cidx, t);
            if (s == kNone) continue;                  // thr = n: never inside a row
            const uint64_t thr = T.thr[s];
            uint32_t q = nc++;                         // insertion sort, ascending
            while (q > 0 && cut[q - 1] > thr) { cut[q] = cut[q - 1]; --q; }
            cut[q] = thr;
        }
        return nc;
    }
    // Does an ORIGIN row start at position b of row i?  Origin rows are the rows of the file cut at
    // the thresholds inside them (the cuts above): where threshold_step goes is one position per
    // origin row and character (fat_layout.h, mismatch lines).
    __device__ __forceinline__ bool origin_start(uint32_t i, uint64_t b) const {
        if (b == T.idx[i]) return true;
        uint64_t cut[kHintSlots];
        const uint32_t nc = cuts(i, cut);
        for (uint32_t q = 0; q < nc; ++q)
            if (cut[q] == b) return true;
        return false;
    }
    __device__ __forceinline__ uint32_t rows() const { return T.r; }
    __device__ __forceinline__ uint64_t n() const { return T.n; }
    __device__ __forceinline__ uint64_t idx(uint32_t j) const { return T.idx[j]; }
    __device__ __forceinline__ uint64_t len(uint32_t j) const { return T.idx[(uint64_t)j + 1] - T.idx[j]; }
    __device__ __forceinline__ uint64_t thr(uint32_t j) const { return T.thr[j]; }
    // (row, offset) of LF^s(first position of row j); may still need the fast-forward
    __device__ __forceinline__ void lf(uint32_t j, int, uint32_t &dj, uint64_t &dt) const {
        const uint4 w = T.rows[j];
        dj = row_interval(w);
        dt = row_offset(w);
    }
    // character / col id met after a-1 LF steps from any position of row j (a = 1 only)
    __device__ __forceinline__ uint32_t ch_at(uint32_t j, int) const { return row_char(T.rows[j]); }
    __device__ __forceinline__ uint32_t cid_at(uint32_t j, int) const { return row_cid(T.rows[j]); }
};

// Fast-forward (LF_table.hpp:256-259) of (j, t) over the source rows.
template <class Src>
__device__ __forceinline__ void src_ff(const Src &S, uint32_t &j, uint64_t &t) {
    uint64_t lenj = S.len(j);
    while (t >= lenj && j < S.rows() - 1) {
        t -= lenj;
        ++j;
        lenj = S.len(j);
    }
}

// Walks the LF image of source row i piece by piece: f(piece_start, piece_len, j, t)
// with (j, t) = source row / offset the piece's first position maps to.
template <class Src, typename F>
__device__ __forceinline__ void for_each_piece(const Src &S, uint32_t i, F f) {
    uint64_t rem = S.len(i);
    uint64_t b = S.idx(i);
    uint64_t cut[kHintSlots];
    const uint32_t nc = S.cuts(i, cut);
    uint32_t ci = 0;
    uint32_t j;
    uint64_t t;
    S.lf(i, 1, j, t);
    src_ff(S, j, t);
    uint64_t lenj = S.len(j);
    while (rem > 0) {
        const uint64_t avail = (j < S.rows() - 1 && t < lenj) ? lenj - t : rem;  // the last row absorbs everything
        uint64_t take = avail < rem ? avail : rem;
        while (ci < nc && cut[ci] <= b) ++ci;                                   // thresholds inside the row
        if (ci < nc && cut[ci] - b < take) take = cut[ci] - b;
        rem -= take;
        while (take > 0) {               // cut pieces longer than kSKMaxLen
            const uint64_t piece = take < kSKMaxLen ? take : kSKMaxLen;
            f(b, (uint32_t)piece, j, t);
            b += piece;
            t += piece;
            take -= piece;
        }
        if (rem > 0 && t >= lenj && j < S.rows() - 1) {
            ++j;
            t = 0;
            lenj = S.len(j);
        }
    }
}

template <class Src>
__global__ __launch_bounds__(256) void sk_count_kernel(Src S, uint32_t *__restrict__ count) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= S.rows()) return;
    uint32_t pieces = 0;
    for_each_piece(S, (uint32_t)i, [&](uint64_t, uint32_t, uint32_t, uint64_t) { ++pieces; });
    count[i] = pieces;
}

// Block-level exclusive scan of 1024 items per block; block totals go to `totals`.
__global__ __launch_bounds__(256) void scan_block_kernel(uint32_t *__restrict__ data, uint64_t n,
                                                         uint32_t *__restrict__ totals) {
    __shared__ uint32_t s_sum[256];
    const uint64_t base = (uint64_t)blockIdx.x * 1024 + (uint64_t)threadIdx.x * 4;
    uint32_t v[4], run = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[q] = base + q < n ? data[base + q] : 0;
        const uint32_t x = v[q];
        v[q] = run;
        run += x;
    }
    s_sum[threadIdx.x] = run;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {   // Hillis-Steele over the 256 per-thread sums
        const uint32_t add = threadIdx.x >= d ? s_sum[threadIdx.x - d] : 0;
        __syncthreads();
        s_sum[threadIdx.x] += add;
        __syncthreads();
    }
    const uint32_t before = threadIdx.x ? s_sum[threadIdx.x - 1] : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (base + q < n) data[base + q] = v[q] + before;
    if (threadIdx.x == 255) totals[blockIdx.x] = s_sum[255];
}

__global__ __launch_bounds__(256) void scan_add_kernel(uint32_t *__restrict__ data, uint64_t n,
                                                       const uint32_t *__restrict__ block_off) {
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint64_t k = i + (uint64_t)q * 256;
        if (k < n) data[k] += block_off[blockIdx.x];
    }
}

// New row holding BWT position `pos`, which lies in source row j.
__device__ __forceinline__ uint32_t sk_find(const uint64_t *idx_new, const uint32_t *first, uint32_t j, uint64_t pos) {
    uint32_t lo = first[j], hi = first[j + 1];   // rows lo .. hi-1 tile source row j
    while (hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (idx_new[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

// A failed HIP call ends the pass with the matching C-ABI code; every temporary is a DevPtr, so
// nothing stays allocated (the AUTO fallback retries a smaller layout in the same process).
#define SK_TRY(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                   \
            (void)hipGetLastError();                                                   \
            return e_ == hipErrorOutOfMemory ? COLBWT_ERR_NOMEM : COLBWT_ERR_HIP;      \
        }                                                                              \
    } while (0)

// Exclusive prefix sum of d_data[0 .. n) in place (u32 items; the sum may exceed 32 bits, then the
// items are meaningless and `total` says so).
inline int exclusive_scan_u32(uint32_t *d_data, uint64_t n, uint64_t &total, std::string &err) {
    DevPtr tot_buf;
    const uint32_t sblocks = (uint32_t)((n + 1023) / 1024);
    SK_TRY(tot_buf.alloc((sblocks ? sblocks : 1) * sizeof(uint32_t)));
    uint32_t *d_tot = tot_buf.as<uint32_t>();
    hipLaunchKernelGGL(scan_block_kernel, dim3(sblocks), dim3(256), 0, 0, d_data, n, d_tot);
    SK_TRY(hipStreamSynchronize(0));
    std::vector<uint32_t> tot(sblocks);
    SK_TRY(hipMemcpy(tot.data(), d_tot, sblocks * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint64_t run = 0;
    for (uint32_t b = 0; b < sblocks; ++b) {
        const uint64_t x = tot[b];
        tot[b] = (uint32_t)run;
        run += x;
    }
    total = run;
    if (run > 0xFFFFFFFEull) return COLBWT_OK;
    SK_TRY(hipMemcpy(d_tot, tot.data(), sblocks * sizeof(uint32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(scan_add_kernel, dim3(sblocks), dim3(256), 0, 0, d_data, n, d_tot);
    SK_TRY(hipStreamSynchronize(0));
    return COLBWT_OK;
}

// count + scan of one refinement pass: `first` receives (rows + 1) u32, first[i] = first new row
// of source row i and first[rows] = `total`, which may exceed 32 bits (then first[] is
// meaningless and the caller gives up).
template <class Src>
int count_and_scan(const Src &S, uint64_t rows, DevPtr &first, uint64_t &total, std::string &err) {
    SK_TRY(first.alloc((rows + 1) * sizeof(uint32_t)));
    uint32_t *d_first = first.as<uint32_t>();
    const uint32_t rblocks = (uint32_t)((rows + 255) / 256);
    hipLaunchKernelGGL(sk_count_kernel<Src>, dim3(rblocks), dim3(256), 0, 0, S, d_first);
    SK_TRY(hipGetLastError());
    SK_TRY(hipMemset(d_first + rows, 0, sizeof(uint32_t)));   // the extra slot receives the total
    return exclusive_scan_u32(d_first, rows + 1, total, err);
}

}  // namespace

}  // namespace colbwt
