// index_kernels.hip -- load-time kernels: on-disk rows -> HBM layout, jump
// tables, and the synthetic read sampler.  None of this is on the query path;
// it runs once per index (col_pml::load, col_bwt.hpp:375-380).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"
#include "lf_device.h"
#include "query_kernels.h"
#include "read_sampler.h"

namespace colbwt {

namespace {

constexpr uint32_t kRelayoutBlock = 256;
constexpr uint32_t kStageExtra = 3;                                    // following rows whose idx is needed
constexpr uint32_t kStageBytes = (kRelayoutBlock + kStageExtra) * kRowBytesDisk + 8;

__device__ __forceinline__ uint64_t lds_le(const uint8_t *p, uint32_t nbytes) {
    uint64_t v = 0;
    for (uint32_t b = 0; b < nbytes; ++b) v |= (uint64_t)p[b] << (8 * b);
    return v;
}

// One thread per row.  The block's packed rows are staged in LDS with
// coalesced dword loads (18-byte rows are only byte aligned), then decoded:
//   byte 0 char | 1-5 idx | 6-9 interval | 10-11 offset | 12 col_id | 13-17 threshold
// (memory image of col_thr: LF_table.hpp:33-40, col_bwt.hpp:43,84).
__global__ __launch_bounds__(kRelayoutBlock) void relayout_kernel(const uint8_t *__restrict__ raw, uint64_t row0,
                                                                  uint64_t count, uint64_t r, uint64_t n,
                                                                  uint4 *__restrict__ rows, uint64_t *__restrict__ idx_out,
                                                                  uint64_t *__restrict__ thr, RelayoutReport *report) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStageBytes];
    __shared__ uint32_t s_present[8], s_cids[8];
    __shared__ uint32_t s_count[256];
    __shared__ uint32_t s_flags, s_bad;
    const uint64_t blk_row = (uint64_t)blockIdx.x * kRelayoutBlock;  // relative to row0
    if (threadIdx.x < 8) s_present[threadIdx.x] = s_cids[threadIdx.x] = 0;
    for (uint32_t t = threadIdx.x; t < 256; t += kRelayoutBlock) s_count[t] = 0;
    if (threadIdx.x == 0) { s_flags = 0; s_bad = kNone; }

    const uint64_t rows_here = (count - blk_row) < kRelayoutBlock ? (count - blk_row) : kRelayoutBlock;
    // bytes of this block's rows, plus the following (up to 3) rows whose idx gives run lengths
    const uint64_t after = r - (row0 + blk_row + rows_here);
    const uint32_t extra = after < kStageExtra ? (uint32_t)after : kStageExtra;
    const uint32_t need = ((uint32_t)rows_here + extra) * kRowBytesDisk;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(raw + blk_row * kRowBytesDisk);  // 4608*k: dword aligned
    uint32_t *dst = reinterpret_cast<uint32_t *>(stage);
    for (uint32_t d = threadIdx.x; d * 4 < need; d += kRelayoutBlock) dst[d] = src[d];
    __syncthreads();

    if (threadIdx.x < rows_here) {
        const uint64_t i = row0 + blk_row + threadIdx.x;
        const uint8_t *p = stage + threadIdx.x * kRowBytesDisk;
        const uint32_t ch = p[0];
        const uint64_t idx = lds_le(p + 1, 5);
        const uint32_t interval = (uint32_t)lds_le(p + 6, 4);
        const uint32_t offset = (uint32_t)lds_le(p + 10, 2);
        const uint32_t cid = p[12];
        const uint64_t threshold = lds_le(p + 13, 5);
        // idx of row i+1 (n past the end): the length of this row
        const uint64_t next_idx = (i + 1 < r) ? lds_le(p + kRowBytesDisk + 1, 5) : n;

        uint32_t flags = 0;
        if (next_idx <= idx) flags |= (i + 1 < r) ? 1u : 4u;  // not strictly increasing / last idx >= n
        if ((uint64_t)interval >= r) flags |= 2u;
        if (i == 0 && idx != 0) flags |= 8u;
        if (flags) {
            atomicOr(&s_flags, flags);
            atomicMin(&s_bad, (uint32_t)i);
        }
        const uint64_t len = next_idx - idx;
        const uint32_t len16 = len < kLenLong ? (uint32_t)len : kLenLong;
        rows[i] = make_uint4(interval, offset | (len16 << 16), 0xFFFFFFFFu,   // cuts / distances: hint_kernel
                             (ch << 8) | (cid << 16) | (kHintAllCompare << 24));
        idx_out[i] = idx;
        thr[i] = threshold;
        if (i + 1 == r) {  // sentinel row: idx = n
            rows[r] = make_uint4(0, 0, 0xFFFFFFFFu, 0);
            idx_out[r] = n;
        }
        atomicOr(&s_present[ch >> 5], 1u << (ch & 31));
        atomicOr(&s_cids[cid >> 5], 1u << (cid & 31));
        atomicAdd(&s_count[ch], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 8 && s_present[threadIdx.x]) atomicOr(&report->present[threadIdx.x], s_present[threadIdx.x]);
    if (threadIdx.x < 8 && s_cids[threadIdx.x]) atomicOr(&report->cids[threadIdx.x], s_cids[threadIdx.x]);
    for (uint32_t t = threadIdx.x; t < 256; t += kRelayoutBlock)
        if (s_count[t]) atomicAdd(&report->count[t], s_count[t]);
    if (threadIdx.x == 0 && s_flags) {
        atomicOr(&report->flags, s_flags);
        atomicMin(&report->first_bad, s_bad);
    }
}

// One wave per 256-row jump block: first / last row of the block holding each
// present character (kNone when the block has none).
__global__ __launch_bounds__(256) void block_first_last_kernel(const uint4 *__restrict__ rows, uint32_t r,
                                                               uint32_t nblk, uint32_t sigma,
                                                               const uint8_t *__restrict__ cmap,
                                                               uint32_t *__restrict__ first,
                                                               uint32_t *__restrict__ last) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nblk) return;
    const uint64_t base = (uint64_t)b << kBlockShift;
    uint32_t cx[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint64_t row = base + (uint64_t)s * 64 + lane;
        cx[s] = kNone;
        if (row < r) cx[s] = cmap[(rows[row].w >> 8) & 0xFFu];
    }
    for (uint32_t c = 0; c < sigma; ++c) {
        uint32_t f = kNone, l = kNone;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned long long m = __ballot(cx[s] == c);
            if (m) {
                const uint32_t lo = (uint32_t)base + s * 64 + (uint32_t)__builtin_ctzll(m);
                const uint32_t hi = (uint32_t)base + s * 64 + 63u - (uint32_t)__builtin_clzll(m);
                if (f == kNone) f = lo;
                l = hi;
            }
        }
        if (lane == 0) {
            first[(uint64_t)b * sigma + c] = f;
            last[(uint64_t)b * sigma + c] = l;
        }
    }
}

// Threshold hints (device_layout.h): for row i and every other present
// character c, how `pos < threshold(succ_c(i))` (col_bwt.hpp:552-560) comes out
// over the row's whole position range [idx, idx+len-1].
__global__ __launch_bounds__(256) void hint_kernel(DevTable T, uint4 *rows_rw, HintChars chars) {
    const uint64_t i64 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i64 >= T.r) return;
    const uint32_t i = (uint32_t)i64;
    const uint4 w = T.rows[i];
    const uint32_t aidx = T.cmap[row_char(w)];
    const uint64_t lo = row_idx(T, i);
    const uint64_t hi = lo + row_len(T, i, w) - 1;
    uint32_t hints = kHintAllCompare, dists = 0xFFFFu;
    // dense indices are ordered by character frequency: the 4 hint slots of a row go to the
    // 4 most frequent OTHER characters (cidx <= 4); rarer ones compare at query time
    const uint32_t top = T.sigma < kHintMaxSigma ? T.sigma : kHintMaxSigma;
    for (uint32_t cidx = 0; cidx < top; ++cidx) {
        if (cidx == aidx || hint_slot(cidx, aidx) >= kHintSlots) continue;
        uint4 t;
        const uint32_t s = succ_char(T, i, chars.c[cidx], cidx, t);
        const uint64_t thr = (s != kNone) ? T.thr[s] : T.n;   // :535 thr = n when there is no successor
        const uint32_t code = hi < thr ? kHintPred : (lo >= thr ? kHintSucc : kHintCompare);
        const uint32_t slot = hint_slot(cidx, aidx);
        hints = (hints & ~(3u << (2 * slot))) | (code << (2 * slot));
        // where a mismatch on this character re-orients to, when it is decided and close
        uint32_t dist = kDistFar;
        if (code == kHintSucc && s - i < kDistFar) dist = s - i;          // s exists: thr <= lo < n
        if (code == kHintPred) {
            const uint32_t q = pred_char(T, i, chars.c[cidx], cidx, t);
            if (q != kNone && i - q < kDistFar) dist = i - q;             // no predecessor: scan at query time
        }
        dists = (dists & ~(0xFu << (4 * slot))) | (dist << (4 * slot));
    }
    // Cuts of the LF jump (device_layout.h): the image of this row starts at offset O of row I;
    // offsets >= cut_a = len(I) - O fall into row I + 1, those >= cut_a + len_b into row I + 2.
    uint32_t cut_a = kCutNone, len_b = kCutNone;
    {
        const uint32_t I = row_interval(w);
        if ((uint64_t)I + 1 < T.r) {
            const uint64_t c = T.idx[(uint64_t)I + 1] - T.idx[I] - row_offset(w);
            if (c < kCutNone) {
                cut_a = (uint32_t)c;
                if ((uint64_t)I + 2 < T.r) {
                    const uint64_t l = T.idx[(uint64_t)I + 2] - T.idx[(uint64_t)I + 1];
                    if (l < kCutNone) len_b = (uint32_t)l;
                }
            }
        }
    }
    rows_rw[i].z = cut_a | (len_b << 8) | (dists << 16);
    rows_rw[i].w = (w.w & 0x00FFFFFFu) | (hints << 24);
}

struct OneStepView {
    DevTable T;
    __device__ __forceinline__ uint64_t n() const { return T.n; }
    __device__ __forceinline__ uint32_t rows() const { return T.r; }
    __device__ __forceinline__ uint64_t idx(uint32_t j) const { return T.idx[j]; }
    __device__ __forceinline__ uint4 load(uint32_t j) const { return T.rows[j]; }
    __device__ __forceinline__ uint32_t ch(const uint4 &w) const { return row_char(w); }
    __device__ __forceinline__ uint32_t lf_row(const uint4 &w) const { return row_interval(w); }
    __device__ __forceinline__ uint32_t lf_off(const uint4 &w) const { return row_offset(w); }
    __device__ __forceinline__ uint64_t len(uint32_t j, const uint4 &) const { return T.idx[(uint64_t)j + 1] - T.idx[j]; }
};

// Synthetic reads by backward walk (read_sampler.h).
__global__ __launch_bounds__(256) void synth_reads_kernel(OneStepView V, uint64_t n_reads, uint32_t m,
                                                          uint32_t sub_permille, uint64_t seed,
                                                          uint8_t *__restrict__ bases,
                                                          uint64_t *__restrict__ read_off) {
    const uint64_t rd = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (rd > n_reads) return;
    read_off[rd] = rd * m;
    if (rd == n_reads) return;
    sample_read(V, rd, m, sub_permille, seed, bases + rd * m);
}

}  // namespace

void launch_relayout(const uint8_t *d_raw, uint64_t row0, uint64_t count, uint64_t r, uint64_t n, uint4 *d_rows,
                     uint64_t *d_idx, uint64_t *d_thr, RelayoutReport *d_report, hipStream_t stream) {
    if (count == 0) return;
    const uint32_t blocks = (uint32_t)((count + kRelayoutBlock - 1) / kRelayoutBlock);
    hipLaunchKernelGGL(relayout_kernel, dim3(blocks), dim3(kRelayoutBlock), 0, stream, d_raw, row0, count, r, n,
                       d_rows, d_idx, d_thr, d_report);
}

void launch_block_first_last(const uint4 *d_rows, uint32_t r, uint32_t nblk, uint32_t sigma, const uint8_t *d_cmap,
                             uint32_t *d_first, uint32_t *d_last, hipStream_t stream) {
    if (nblk == 0) return;
    hipLaunchKernelGGL(block_first_last_kernel, dim3((nblk + 3) / 4), dim3(256), 0, stream, d_rows, r, nblk, sigma,
                       d_cmap, d_first, d_last);
}

void launch_hints(const DevTable &T, uint4 *d_rows_rw, const HintChars &chars, hipStream_t stream) {
    const uint32_t blocks = (uint32_t)(((uint64_t)T.r + 255) / 256);
    hipLaunchKernelGGL(hint_kernel, dim3(blocks), dim3(256), 0, stream, T, d_rows_rw, chars);
}

void launch_synth_reads(const DevTable &T, uint64_t n_reads, uint32_t read_len, uint32_t sub_permille, uint64_t seed,
                        uint8_t *d_bases, uint64_t *d_read_off, hipStream_t stream) {
    const uint64_t blocks = (n_reads + 1 + 255) / 256;
    hipLaunchKernelGGL(synth_reads_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, OneStepView{T}, n_reads, read_len,
                       sub_permille, seed, d_bases, d_read_off);
}

}  // namespace colbwt
