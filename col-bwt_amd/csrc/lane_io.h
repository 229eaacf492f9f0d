// lane_io.h -- per-lane input/output plumbing shared by the query kernels:
// the LDS-staged read window and the register collector of PML / col-id values.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "query_kernels.h"

namespace colbwt {

// Read bytes: 64 bytes of the lane's read (one 64-byte-aligned block of `bases`,
// 4 x uint4 from one HBM line) are staged in LDS, dword-major ([16][block] so
// consecutive lanes hit consecutive banks); each step reads its byte from there.
struct ReadWindow {
    __device__ __forceinline__ void refill(uint32_t (*s_rd)[kQueryBlock], const uint8_t *bases, uint64_t g) {
        const uint4 *src = reinterpret_cast<const uint4 *>(bases + (g & ~(uint64_t)63));
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = src[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s_rd[4 * q + 0][threadIdx.x] = v[q].x;
            s_rd[4 * q + 1][threadIdx.x] = v[q].y;
            s_rd[4 * q + 2][threadIdx.x] = v[q].z;
            s_rd[4 * q + 3][threadIdx.x] = v[q].w;
        }
    }
    __device__ __forceinline__ uint32_t get(uint32_t (*s_rd)[kQueryBlock], uint64_t g) {
        const uint32_t b = (uint32_t)g & 63u;
        return (s_rd[b >> 2][threadIdx.x] >> (8 * (b & 3u))) & 0xFFu;
    }
};

// Output collector: kFlush = 16 bases per flush, flush boundaries at global
// element indices that are multiples of 16, so a full flush is aligned vector
// stores covering whole 32-byte sectors; partial groups (read ends) go out
// element by element.
constexpr uint32_t kFlush = 16;

template <typename PmlT>
struct OutAcc;

template <>
struct OutAcc<uint16_t> {
    uint64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, c0 = 0, c1 = 0;
    uint32_t cnt = 0;
    __device__ __forceinline__ void push(uint32_t L, uint32_t cid) {
        p3 = (p3 << 16) | (p2 >> 48);
        p2 = (p2 << 16) | (p1 >> 48);
        p1 = (p1 << 16) | (p0 >> 48);
        p0 = (p0 << 16) | (uint64_t)(L & 0xFFFFu);
        c1 = (c1 << 8) | (c0 >> 56);
        c0 = (c0 << 8) | (uint64_t)cid;
        ++cnt;
    }
    __device__ __forceinline__ void flush(uint16_t *pml, uint8_t *cid, uint64_t g) {
        if (cnt == kFlush) {
            uint4 *dst = reinterpret_cast<uint4 *>(pml + g);
            dst[0] = make_uint4((uint32_t)p0, (uint32_t)(p0 >> 32), (uint32_t)p1, (uint32_t)(p1 >> 32));
            dst[1] = make_uint4((uint32_t)p2, (uint32_t)(p2 >> 32), (uint32_t)p3, (uint32_t)(p3 >> 32));
            *reinterpret_cast<uint4 *>(cid + g) =
                make_uint4((uint32_t)c0, (uint32_t)(c0 >> 32), (uint32_t)c1, (uint32_t)(c1 >> 32));
        } else {
            for (uint32_t e = 0; e < cnt; ++e) {
                pml[g + e] = (uint16_t)p0;
                cid[g + e] = (uint8_t)c0;
                p0 = (p0 >> 16) | (p1 << 48);
                p1 = (p1 >> 16) | (p2 << 48);
                p2 = (p2 >> 16) | (p3 << 48);
                p3 >>= 16;
                c0 = (c0 >> 8) | (c1 << 56);
                c1 >>= 8;
            }
        }
        cnt = 0;
    }
};

template <>
struct OutAcc<uint32_t> {  // reads longer than 65535 bases: wide PML, stored per base
    __device__ __forceinline__ void push(uint32_t, uint32_t) {}
    __device__ __forceinline__ void flush(uint32_t *, uint8_t *, uint64_t) {}
};

}  // namespace colbwt
