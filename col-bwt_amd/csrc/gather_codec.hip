// gather_codec.hip -- codec for the one exchange step of the multi-GPU path, the
// gather of the per-base results to rank 0 over xGMI.
//
// The PML values of a read are determined by WHERE they are zero: going left
// from a base, the length grows by one per matching base and drops to 0 at a
// mismatch (col_bwt.hpp:516-521).  So a rank ships one bit per base (value == 0)
// instead of 16 and rank 0 rebuilds the values: with j the first position >= k
// that is a reset or the last base of k's read,
//     pml[k] = j - k + (pml[j] == 0 ? 0 : 1).
// 3 bytes per base become 1.125.  Bit k of the flat masks is base k of the rank's
// concatenated reads; 32 bases per word.
//
// The col ids are whatever the table's rows carry (col_bwt.hpp:513): a result can only hold an
// id that some row of the index holds, and every rank holds the same index.  So the ids travel as
// codes of that dictionary -- ceil(log2(#distinct ids)) bits per base, the same on every rank,
// hence gathers of equal, known sizes -- as bit planes: `bits` words per 32 bases, word p holding
// bit p of their 32 codes.  The C2 index has 7 distinct ids: 3 bits, 0.375 bytes per base; with the
// PML bit 0.5 bytes per base instead of 3.  An index with more than 16 distinct ids sends its col
// ids as they are (a run-length form would pay there, but needs a variable-size exchange).
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

// 32 col ids -> `bits` words: plane p, bit k = bit p of the code of base 32w + k
__global__ __launch_bounds__(256) void cid_pack_kernel(const uint8_t *__restrict__ cid, uint64_t n_bases, uint64_t n_words,
                                                       CidLut code_of, uint32_t bits, uint32_t *__restrict__ planes) {
    __shared__ uint8_t s_code[256];
    s_code[threadIdx.x] = code_of.v[threadIdx.x];
    __syncthreads();
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    const uint64_t k0 = w * 32;
    uint8_t v[32];
    if (k0 + 32 <= n_bases) {
        const uint4 *src = reinterpret_cast<const uint4 *>(cid + k0);      // 32-byte aligned block: cid is 16-byte aligned
        const uint4 a = src[0], b = src[1];
        memcpy(v, &a, 16);
        memcpy(v + 16, &b, 16);
    } else {
        for (uint32_t k = 0; k < 32; ++k) v[k] = k0 + k < n_bases ? cid[k0 + k] : 0;
    }
    uint32_t out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (uint32_t k = 0; k < 32; ++k) {
        const uint32_t c = s_code[v[k]];
#pragma unroll
        for (uint32_t p = 0; p < 8; ++p) out[p] |= ((c >> p) & 1u) << k;
    }
    for (uint32_t p = 0; p < bits; ++p) planes[w * bits + p] = out[p];
}

__global__ __launch_bounds__(256) void cid_unpack_kernel(const uint32_t *__restrict__ planes, uint64_t first_word, uint64_t n_words,
                                                         CidLut id_of, uint32_t bits, uint8_t *__restrict__ cid) {
    __shared__ uint8_t s_id[256];
    s_id[threadIdx.x] = id_of.v[threadIdx.x];
    __syncthreads();
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_words) return;
    const uint64_t w = first_word + t;
    uint32_t pl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t p = 0; p < bits; ++p) pl[p] = planes[w * bits + p];
    uint32_t out[8];
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) {
        uint32_t word = 0;
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
            const uint32_t k = 4 * q + e;
            uint32_t c = 0;
#pragma unroll
            for (uint32_t p = 0; p < 8; ++p) c |= ((pl[p] >> k) & 1u) << p;
            word |= (uint32_t)s_id[c] << (8 * e);
        }
        out[q] = word;
    }
    uint4 *dst = reinterpret_cast<uint4 *>(cid + w * 32);
    dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
    dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
}

}  // namespace

void launch_cid_pack(const uint8_t *d_cid, uint64_t n_bases, const CidLut &code_of, uint32_t bits, uint32_t *d_planes, hipStream_t stream) {
    const uint64_t n_words = (n_bases + 31) / 32;
    if (n_words == 0) return;
    hipLaunchKernelGGL(cid_pack_kernel, dim3((uint32_t)((n_words + 255) / 256)), dim3(256), 0, stream, d_cid, n_bases, n_words, code_of,
                       bits, d_planes);
}

void launch_cid_unpack(const uint32_t *d_planes, uint64_t first_word, uint64_t n_words, const CidLut &id_of, uint32_t bits, uint8_t *d_cid,
                       hipStream_t stream) {
    if (n_words == 0) return;
    hipLaunchKernelGGL(cid_unpack_kernel, dim3((uint32_t)((n_words + 255) / 256)), dim3(256), 0, stream, d_planes, first_word, n_words,
                       id_of, bits, d_cid);
}

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
