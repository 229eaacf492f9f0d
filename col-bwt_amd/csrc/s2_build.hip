// s2_build.hip -- builds the two-step layout (s2_layout.h) on the device from
// the one-step layout (device_layout.h) that the loader has just produced.
// Runs once per index, off the query path.
//
//   count   per original row: number of refined rows = pieces of its LF image
//           between destination-row boundaries (+ cuts at 65534 positions)
//   scan    exclusive prefix sum -> first refined row of every original row
//   emit    refined rows: idx, len, char, col id, and the image of their first
//           position in ORIGINAL coordinates (row j1, offset t1), parked in d0..d2
//   link    d0..d2 -> (I1,O1) and (I2,O2) in refined coordinates, char2/cid2
//   then the same jump tables and threshold hints as the one-step layout
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "device_layout.h"
#include "jump_tables.h"
#include "lf_device.h"
#include "query_kernels.h"
#include "s2_layout.h"

namespace colbwt {

namespace {

// Walks the LF image of original row i piece by piece.  f(piece_start_in_source, piece_len, j, t)
// with (j, t) = original row / offset the piece's first position maps to.
template <typename F>
__device__ __forceinline__ void for_each_piece(const DevTable &T, uint32_t i, F f) {
    const uint4 w = T.rows[i];
    uint64_t rem = row_len(T, i, w);
    uint64_t b = row_idx(T, i);
    uint32_t j = row_interval(w);
    uint64_t t = row_offset(w);
    uint4 wj = T.rows[j];
    uint64_t lenj = row_len(T, j, wj);
    while (t >= lenj && j < T.r - 1) {   // the stored (interval, offset) may need the fast-forward itself
        t -= lenj;
        ++j;
        wj = T.rows[j];
        lenj = row_len(T, j, wj);
    }
    while (rem > 0) {
        uint64_t avail = (j < T.r - 1 && t < lenj) ? lenj - t : rem;  // the last row absorbs everything
        uint64_t take = avail < rem ? avail : rem;
        rem -= take;
        while (take > 0) {               // cut pieces longer than kS2MaxLen
            const uint64_t piece = take < kS2MaxLen ? take : kS2MaxLen;
            f(b, (uint32_t)piece, j, t);
            b += piece;
            t += piece;
            take -= piece;
        }
        if (rem > 0) {
            ++j;
            t = 0;
            wj = T.rows[j];
            lenj = row_len(T, j, wj);
        }
    }
}

__global__ __launch_bounds__(256) void s2_count_kernel(DevTable T, uint32_t *__restrict__ count) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= T.r) return;
    uint32_t pieces = 0;
    for_each_piece(T, (uint32_t)i, [&](uint64_t, uint32_t, uint32_t, uint64_t) { ++pieces; });
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

__device__ __forceinline__ uint32_t *s2_row_ptr(uint8_t *lines, uint32_t j) {
    return reinterpret_cast<uint32_t *>(lines + s2_row_off(j));
}

__global__ __launch_bounds__(256) void s2_emit_kernel(DevTable T, const uint32_t *__restrict__ first2,
                                                      uint8_t *__restrict__ lines, uint64_t *__restrict__ idx2,
                                                      uint64_t *__restrict__ thr2, uint32_t r2) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= T.r) return;
    const uint4 w = T.rows[i];
    const uint32_t ch = row_char(w), cidv = row_cid(w);
    const uint64_t thr = T.thr[i];
    uint32_t out = first2[i];
    for_each_piece(T, (uint32_t)i, [&](uint64_t b, uint32_t len, uint32_t j, uint64_t t) {
        uint32_t *p = s2_row_ptr(lines, out);
        p[0] = j;                       // parked: image of the first position, original coordinates
        p[1] = (uint32_t)t;
        p[2] = (uint32_t)(t >> 32);
        p[3] = len | (ch << 16) | (cidv << 24);
        p[4] = 0xFFFFFFFFu;             // next-row lengths unknown, mismatch targets far (filled later)
        p[5] = kHintAllCompare << 24;
        idx2[out] = b;
        thr2[out] = thr;
        ++out;
    });
    if (i + 1 == T.r) {                 // sentinel refined row: idx = n
        uint32_t *p = s2_row_ptr(lines, r2);
        p[0] = p[1] = p[2] = p[3] = p[5] = 0;
        p[4] = 0xFFFFFFFFu;
        idx2[r2] = T.n;
    }
}

// Refined row holding BWT position `pos`, which lies in original row j.
__device__ __forceinline__ uint32_t s2_find(const uint64_t *idx2, const uint32_t *first2, uint32_t j, uint64_t pos) {
    uint32_t lo = first2[j], hi = first2[j + 1];   // rows lo .. hi-1 tile original row j
    while (hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (idx2[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void s2_link_kernel(DevTable T, const uint32_t *__restrict__ first2,
                                                      uint8_t *lines, const uint64_t *__restrict__ idx2, uint32_t r2) {
    const uint64_t i2 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i2 >= r2) return;
    uint32_t *p = s2_row_ptr(lines, (uint32_t)i2);
    const uint32_t j1 = p[0];
    const uint64_t t1 = (uint64_t)p[1] | ((uint64_t)p[2] << 32);
    const uint4 w1 = T.rows[j1];
    // one step: LF(first position) = idx[j1] + t1
    const uint64_t pos1 = row_idx(T, j1) + t1;
    const uint32_t I1 = s2_find(idx2, first2, j1, pos1);
    const uint32_t O1 = (uint32_t)(pos1 - idx2[I1]);
    // two steps: LF from (j1, t1) in original coordinates (LF_table.hpp:251-262)
    uint32_t j = row_interval(w1);
    uint64_t t = (uint64_t)row_offset(w1) + t1;
    uint4 wj = T.rows[j];
    uint64_t lenj = row_len(T, j, wj);
    while (t >= lenj && j < T.r - 1) {
        t -= lenj;
        ++j;
        wj = T.rows[j];
        lenj = row_len(T, j, wj);
    }
    const uint64_t pos2 = row_idx(T, j) + t;
    const uint32_t I2 = s2_find(idx2, first2, j, pos2);
    const uint32_t O2 = (uint32_t)(pos2 - idx2[I2]);
    p[0] = I1;
    p[1] = I2;
    p[2] = (O1 & 0xFFFFu) | (O2 << 16);
    // lengths of the next two refined rows (multi-hop fast-forward)
    uint32_t l1 = kLen8Long, l2 = kLen8Long;
    if (i2 + 1 < r2) {
        const uint64_t a = idx2[i2 + 2] - idx2[i2 + 1];
        if (a < kLen8Long) l1 = (uint32_t)a;
    }
    if (i2 + 2 < r2) {
        const uint64_t a = idx2[i2 + 3] - idx2[i2 + 2];
        if (a < kLen8Long) l2 = (uint32_t)a;
    }
    p[4] = l1 | (l2 << 8) | 0xFFFF0000u;
    p[5] = (p[5] & 0xFF0000FFu) | (row_char(w1) << 8) | (row_cid(w1) << 16);
}

// One wave per jump block of 320 refined rows (see block_first_last_kernel).
__global__ __launch_bounds__(256) void s2_block_first_last_kernel(S2Table T, uint32_t *__restrict__ first,
                                                                  uint32_t *__restrict__ last) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= T.nblk) return;
    const uint64_t base = (uint64_t)b * kS2BlockRows;
    uint32_t cx[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const uint64_t row = base + (uint64_t)s * 64 + lane;
        cx[s] = kNone;
        if (row < T.r2) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(T.lines + s2_row_off((uint32_t)row));
            cx[s] = T.cmap[(p[3] >> 16) & 0xFFu];
        }
    }
    for (uint32_t c = 0; c < T.sigma; ++c) {
        uint32_t f = kNone, l = kNone;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const unsigned long long m = __ballot(cx[s] == c);
            if (m) {
                const uint32_t lo = (uint32_t)base + s * 64 + (uint32_t)__builtin_ctzll(m);
                const uint32_t hi = (uint32_t)base + s * 64 + 63u - (uint32_t)__builtin_clzll(m);
                if (f == kNone) f = lo;
                l = hi;
            }
        }
        if (lane == 0) {
            first[(uint64_t)b * T.sigma + c] = f;
            last[(uint64_t)b * T.sigma + c] = l;
        }
    }
}

// Threshold hints over refined rows (see hint_kernel in index_kernels.hip).
__global__ __launch_bounds__(256) void s2_hint_kernel(S2Table T, uint8_t *lines_rw, HintChars chars) {
    const uint64_t i64 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i64 >= T.r2) return;
    const uint32_t i = (uint32_t)i64;
    const S2Row w = s2_load(T, i);
    const uint32_t aidx = T.cmap[s2_char(w)];
    const uint64_t lo = T.idx[i];
    const uint64_t hi = lo + s2_len(w) - 1;
    uint32_t hints = kHintAllCompare, dists = 0xFFFFu;
    const uint32_t top = T.sigma < kHintMaxSigma ? T.sigma : kHintMaxSigma;
    for (uint32_t cidx = 0; cidx < top; ++cidx) {
        if (cidx == aidx || hint_slot(cidx, aidx) >= kHintSlots) continue;
        S2Row t;
        const uint32_t s = s2_succ_char(T, i, chars.c[cidx], cidx, t);
        const uint64_t thr = (s != kNone) ? T.thr[s] : T.n;
        const uint32_t code = hi < thr ? kHintPred : (lo >= thr ? kHintSucc : kHintCompare);
        const uint32_t slot = hint_slot(cidx, aidx);
        hints = (hints & ~(3u << (2 * slot))) | (code << (2 * slot));
        uint32_t dist = kDistFar;
        if (code == kHintSucc && s - i < kDistFar) dist = s - i;
        if (code == kHintPred) {
            const uint32_t q = s2_pred_char(T, i, chars.c[cidx], cidx, t);
            if (q != kNone && i - q < kDistFar) dist = i - q;
        }
        dists = (dists & ~(0xFu << (4 * slot))) | (dist << (4 * slot));
    }
    uint32_t *p = s2_row_ptr(lines_rw, i);
    p[4] = (w.d[4] & 0x0000FFFFu) | (dists << 16);
    p[5] = (w.d[5] & 0x00FFFFFFu) | (hints << 24);
}

}  // namespace

#define S2_TRY(expr)                                                          \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) {                                               \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);          \
            return false;                                                     \
        }                                                                     \
    } while (0)

// Builds the two-step layout from the one-step tables of `T`.  Returns false with
// `err` set when it cannot (more than 2^32-2 refined rows, out of memory).
bool build_s2(const DevTable &T, const HintChars &chars, S2Table &out, void **d_lines, void **d_idx, void **d_thr,
              void **d_next, void **d_prev, uint64_t &bytes, std::string &err) {
    const uint64_t r = T.r;
    uint32_t *d_first2 = nullptr, *d_tot = nullptr;
    S2_TRY(hipMalloc((void **)&d_first2, (r + 1) * sizeof(uint32_t)));
    const uint32_t rblocks = (uint32_t)((r + 255) / 256);
    hipLaunchKernelGGL(s2_count_kernel, dim3(rblocks), dim3(256), 0, 0, T, d_first2);
    S2_TRY(hipGetLastError());
    // exclusive scan of r + 1 counts (the extra slot receives the total)
    S2_TRY(hipMemset(d_first2 + r, 0, sizeof(uint32_t)));
    const uint64_t nscan = r + 1;
    const uint32_t sblocks = (uint32_t)((nscan + 1023) / 1024);
    S2_TRY(hipMalloc((void **)&d_tot, sblocks * sizeof(uint32_t)));
    hipLaunchKernelGGL(scan_block_kernel, dim3(sblocks), dim3(256), 0, 0, d_first2, nscan, d_tot);
    S2_TRY(hipStreamSynchronize(0));
    std::vector<uint32_t> tot(sblocks);
    S2_TRY(hipMemcpy(tot.data(), d_tot, sblocks * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint64_t run = 0;
    for (uint32_t b = 0; b < sblocks; ++b) {
        const uint64_t x = tot[b];
        tot[b] = (uint32_t)run;
        run += x;
    }
    if (run > 0xFFFFFFFEull) {
        (void)hipFree(d_first2);
        (void)hipFree(d_tot);
        err = "two-step layout needs " + std::to_string(run) + " refined rows (> 2^32-2)";
        return false;
    }
    S2_TRY(hipMemcpy(d_tot, tot.data(), sblocks * sizeof(uint32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(scan_add_kernel, dim3(sblocks), dim3(256), 0, 0, d_first2, nscan, d_tot);
    S2_TRY(hipStreamSynchronize(0));
    (void)hipFree(d_tot);
    const uint32_t r2 = (uint32_t)run;

    const uint64_t nlines = ((uint64_t)r2 + 1 + kS2RowsPerLine - 1) / kS2RowsPerLine + 1;
    S2_TRY(hipMalloc(d_lines, nlines * 128));
    S2_TRY(hipMemset(*d_lines, 0, nlines * 128));
    S2_TRY(hipMalloc(d_idx, ((uint64_t)r2 + 4) * sizeof(uint64_t)));
    S2_TRY(hipMemset(*d_idx, 0xFF, ((uint64_t)r2 + 4) * sizeof(uint64_t)));
    S2_TRY(hipMalloc(d_thr, (uint64_t)r2 * sizeof(uint64_t)));
    hipLaunchKernelGGL(s2_emit_kernel, dim3(rblocks), dim3(256), 0, 0, T, d_first2, (uint8_t *)*d_lines,
                       (uint64_t *)*d_idx, (uint64_t *)*d_thr, r2);
    S2_TRY(hipStreamSynchronize(0));
    const uint32_t r2blocks = (uint32_t)(((uint64_t)r2 + 255) / 256);
    hipLaunchKernelGGL(s2_link_kernel, dim3(r2blocks), dim3(256), 0, 0, T, d_first2, (uint8_t *)*d_lines,
                       (const uint64_t *)*d_idx, r2);
    S2_TRY(hipStreamSynchronize(0));
    (void)hipFree(d_first2);

    out.lines = (const uint8_t *)*d_lines;
    out.idx = (const uint64_t *)*d_idx;
    out.thr = (const uint64_t *)*d_thr;
    out.cmap = T.cmap;
    out.n = T.n;
    out.r2 = r2;
    out.sigma = T.sigma;
    out.nblk = (uint32_t)(((uint64_t)r2 + kS2BlockRows - 1) / kS2BlockRows);
    const uint64_t entries = (uint64_t)out.nblk * out.sigma;
    S2_TRY(hipMalloc(d_next, (entries ? entries : 1) * sizeof(uint32_t)));
    S2_TRY(hipMalloc(d_prev, (entries ? entries : 1) * sizeof(uint32_t)));
    out.next_tbl = (const uint32_t *)*d_next;
    out.prev_tbl = (const uint32_t *)*d_prev;
    hipLaunchKernelGGL(s2_block_first_last_kernel, dim3((out.nblk + 3) / 4), dim3(256), 0, 0, out, (uint32_t *)*d_next,
                       (uint32_t *)*d_prev);
    S2_TRY(hipStreamSynchronize(0));
    {
        std::vector<uint32_t> first(entries), last(entries), next, prev;
        S2_TRY(hipMemcpy(first.data(), *d_next, entries * sizeof(uint32_t), hipMemcpyDeviceToHost));
        S2_TRY(hipMemcpy(last.data(), *d_prev, entries * sizeof(uint32_t), hipMemcpyDeviceToHost));
        finish_jump_tables(first, last, out.nblk, out.sigma, next, prev);
        S2_TRY(hipMemcpy(*d_next, next.data(), entries * sizeof(uint32_t), hipMemcpyHostToDevice));
        S2_TRY(hipMemcpy(*d_prev, prev.data(), entries * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(s2_hint_kernel, dim3(r2blocks), dim3(256), 0, 0, out, (uint8_t *)*d_lines, chars);
    S2_TRY(hipGetLastError());
    S2_TRY(hipStreamSynchronize(0));
    bytes = nlines * 128 + (2 * (uint64_t)r2 + 4) * sizeof(uint64_t) + 2 * (entries ? entries : 1) * sizeof(uint32_t);
    return true;
}

}  // namespace colbwt
