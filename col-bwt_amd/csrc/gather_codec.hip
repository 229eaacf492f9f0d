// gather_codec.hip -- codec for the one exchange step of the multi-GPU path, the
// gather of the per-base results to rank 0 over xGMI.
//
// The PML values of a read are determined by WHERE they are zero: going left
// from a base, the length grows by one per matching base and drops to 0 at a
// mismatch (col_bwt.hpp:516-521).  So a rank ships one bit per base (value == 0)
// instead of 16 and rank 0 rebuilds the values: with j the first position >= k
// that is a reset or the last base of k's read,
//     pml[k] = j - k + (pml[j] == 0 ? 0 : 1).
// 3 bytes per base become 1.125 (col ids travel as they are).  Bit k of the flat
// masks is base k of the rank's concatenated reads; 32 bases per word.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "query_kernels.h"

namespace colbwt {

namespace {

// word w <- (pml[32w + b] == 0) for b in [0, 32); bases past n_bases read as "not zero"
__global__ __launch_bounds__(256) void pml_pack_kernel(const uint16_t *__restrict__ pml, uint64_t n_bases,
                                                       uint64_t n_words, uint32_t *__restrict__ mask) {
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    const uint64_t k0 = w * 32;
    uint32_t bits = 0;
    if (k0 + 32 <= n_bases) {
        uint4 v[4];                                                     // 64-byte aligned block: pml is 32-byte aligned
        const uint4 *src = reinterpret_cast<const uint4 *>(pml + k0);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = src[q];
        uint16_t val[32];
        memcpy(val, v, sizeof(val));
#pragma unroll
        for (uint32_t b = 0; b < 32; ++b) bits |= (val[b] == 0 ? 1u : 0u) << b;
    } else {
        for (uint32_t b = 0; k0 + b < n_bases; ++b) bits |= (pml[k0 + b] == 0 ? 1u : 0u) << b;
    }
    mask[w] = bits;
}

// bit (read_off[r+1] - 1) <- 1 for every non-empty read r (mask zeroed by the caller)
__global__ __launch_bounds__(256) void read_end_mask_kernel(const uint64_t *__restrict__ read_off, uint64_t n_reads,
                                                            uint32_t *__restrict__ mask) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t a = read_off[r], b = read_off[r + 1];
    if (b > a) atomicOr(&mask[(b - 1) >> 5], 1u << ((b - 1) & 31));
}

// 32 values per thread, written as one aligned 64-byte block.
__global__ __launch_bounds__(256) void pml_unpack_kernel(const uint32_t *__restrict__ flag,
                                                         const uint32_t *__restrict__ last, uint64_t first_word,
                                                         uint64_t n_words, uint64_t total_words,
                                                         uint16_t *__restrict__ pml) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_words) return;
    const uint64_t w = first_word + t;
    const uint32_t f = flag[w];
    const uint32_t stop = f | last[w];
    // the stop that bases above the word's highest stop run into: in a following word
    // (a read's last base is a stop, so the search ends inside the read)
    uint64_t j = 0;
    uint32_t add = 0;
    bool have = false;
    for (uint64_t x = w + 1; x < total_words; ++x) {
        const uint32_t fx = flag[x], sx = fx | last[x];
        if (sx) {
            const uint32_t b = (uint32_t)__builtin_ctz(sx);
            j = x * 32 + b;
            add = (fx >> b) & 1u ? 0u : 1u;
            have = true;
            break;
        }
    }
    uint32_t out[16];
    for (int b = 31; b >= 0; --b) {
        const uint64_t k = w * 32 + (uint32_t)b;
        if ((stop >> b) & 1u) {
            j = k;
            add = (f >> b) & 1u ? 0u : 1u;
            have = true;
        }
        const uint32_t v = have ? (uint32_t)(j - k) + add : 0u;   // padding past the last read: 0
        if (b & 1) out[b >> 1] = v << 16;
        else out[b >> 1] |= v & 0xFFFFu;
    }
    uint4 *dst = reinterpret_cast<uint4 *>(pml + w * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = make_uint4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
}

}  // namespace

void launch_pml_pack(const uint16_t *d_pml, uint64_t n_bases, uint32_t *d_mask, hipStream_t stream) {
    const uint64_t n_words = (n_bases + 31) / 32;
    if (n_words == 0) return;
    hipLaunchKernelGGL(pml_pack_kernel, dim3((uint32_t)((n_words + 255) / 256)), dim3(256), 0, stream, d_pml, n_bases,
                       n_words, d_mask);
}

void launch_read_end_mask(const uint64_t *d_read_off, uint64_t n_reads, uint32_t *d_mask, hipStream_t stream) {
    if (n_reads == 0) return;
    hipLaunchKernelGGL(read_end_mask_kernel, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, stream, d_read_off,
                       n_reads, d_mask);
}

void launch_pml_unpack(const uint32_t *d_flag, const uint32_t *d_last, uint64_t first_word, uint64_t n_words,
                       uint64_t total_words, uint16_t *d_pml, hipStream_t stream) {
    if (n_words == 0) return;
    hipLaunchKernelGGL(pml_unpack_kernel, dim3((uint32_t)((n_words + 255) / 256)), dim3(256), 0, stream, d_flag, d_last,
                       first_word, n_words, total_words, d_pml);
}

}  // namespace colbwt
