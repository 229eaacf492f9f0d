// lane_io.h -- per-lane input/output plumbing shared by the query kernels:
// the LDS-staged read window and the register collector of PML / col-id values.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "query_kernels.h"

namespace colbwt {

// Output collector: kFlush = 16 bases per flush, flush boundaries at global
// element indices that are multiples of 16, so a full flush is aligned vector
// stores covering whole 32-byte sectors; partial groups (read ends) go out in
// power-of-two pieces.  The group at the START of a read is flushed after the
// kernel's loop, where the wave has reconverged (inside the loop every lane
// would run it alone, in the trip its own read ends).
constexpr uint32_t kFlush = 16;

template <typename PmlT>
struct OutAcc;

struct U128 {
    uint64_t lo, hi;
};
struct Q128 {   // 16 bytes as a plain aggregate (HIP's uint4 has constructors: not packable)
    uint32_t x, y, z, w;
};
// store of a value at an address that is only aligned for the array's element type
template <typename V, typename E>
__device__ __forceinline__ void put(E *dst, V v) {
    struct __attribute__((packed, aligned(alignof(E)))) Slot {
        V v;
    };
    reinterpret_cast<Slot *>(dst)->v = v;
}

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
            // read ends: 8 + 4 + 2 + 1 elements, each piece one (unaligned) store per array
            uint16_t *dp = pml + g;
            uint8_t *dc = cid + g;
            if (cnt & 8u) {
                put(dp, U128{p0, p1});
                put(dc, c0);
                p0 = p2; p1 = p3; c0 = c1;
                dp += 8; dc += 8;
            }
            if (cnt & 4u) {
                put(dp, p0);
                put(dc, (uint32_t)c0);
                p0 = p1; c0 >>= 32;
                dp += 4; dc += 4;
            }
            if (cnt & 2u) {
                put(dp, (uint32_t)p0);
                put(dc, (uint16_t)c0);
                p0 >>= 32; c0 >>= 16;
                dp += 2; dc += 2;
            }
            if (cnt & 1u) {
                *dp = (uint16_t)p0;
                *dc = (uint8_t)c0;
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

// ---- wave-synchronous variants (sk_query.hip) --------------------------------------
// A vector-memory instruction occupies the CU's address / tag pipeline for about as long
// with one active lane as with 64, and the query kernels are bound by exactly that pipeline
// (DESIGN.md 4.1).  So both streams below touch memory only at the top of a loop trip,
// where the wave is converged: one instruction then serves every lane that needs it.

// Read bytes: a 64-byte window per lane that slides down the read.  It is refilled for ALL
// lanes of the wave whenever ANY lane is about to run out (each lane at its own position:
// 4 x 16 bytes ending just above its next base), i.e. 3-4 times per 150-base read and wave
// instead of once per lane and 64 bases.
struct SlidingWindow {
    uint32_t wb;   // low 32 bits of the global index of the window's first byte
    __device__ __forceinline__ void init(uint64_t g) { wb = (uint32_t)g + 1u; }   // nothing buffered
    // bytes buffered at and below global index g
    __device__ __forceinline__ uint32_t avail(uint64_t g) const { return (uint32_t)g - wb + 1u; }
    __device__ __forceinline__ void refill(uint32_t (*s_rd)[kQueryBlock], const uint8_t *bases, uint64_t g) {
        const uint64_t a = g >= 60 ? (g - 60) & ~(uint64_t)3 : 0;   // dword aligned; [a, a + 64) holds g
        struct __attribute__((packed, aligned(4))) Q {
            uint32_t x, y, z, w;
        };
        const Q *src = reinterpret_cast<const Q *>(bases + a);
        Q v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = src[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s_rd[4 * q + 0][threadIdx.x] = v[q].x;
            s_rd[4 * q + 1][threadIdx.x] = v[q].y;
            s_rd[4 * q + 2][threadIdx.x] = v[q].z;
            s_rd[4 * q + 3][threadIdx.x] = v[q].w;
        }
        wb = (uint32_t)a;
    }
    __device__ __forceinline__ uint32_t get(uint32_t (*s_rd)[kQueryBlock], uint64_t g) const {
        const uint32_t b = ((uint32_t)g - wb) & 63u;
        return (s_rd[b >> 2][threadIdx.x] >> (8 * (b & 3u))) & 0xFFu;
    }
    // the 8 read bytes ending at g as one little-endian word: byte 7 = base g, byte 0 = base g - 7
    // (bytes below the window are whatever the ring holds: callers cap what they use by `avail`)
    __device__ __forceinline__ uint64_t get8(uint32_t (*s_rd)[kQueryBlock], uint64_t g) const {
        const uint32_t b = ((uint32_t)g - wb) & 63u, d = b >> 2;
        const uint32_t x2 = s_rd[d][threadIdx.x], x1 = s_rd[(d - 1u) & 15u][threadIdx.x], x0 = s_rd[(d - 2u) & 15u][threadIdx.x];
        const uint32_t sh = 8u * ((b & 3u) + 1u);                     // 8 .. 32
        const uint64_t lo = (uint64_t)x0 | ((uint64_t)x1 << 32);
        return (lo >> sh) | ((uint64_t)x2 << (64u - sh));
    }
};

// Outputs: room for 18 elements, so that up to 3 values can be pushed per trip and the
// aligned 16-element group they complete is stored by the ONE flush at the top of the next
// trip (element 0 = lowest address = newest).
struct OutAcc18 {
    uint64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, c0 = 0, c1 = 0;
    uint32_t p4 = 0, c2 = 0;
    uint32_t cnt = 0;
    __device__ __forceinline__ void push(uint32_t L, uint32_t cid) {
        p4 = (p4 << 16) | (uint32_t)(p3 >> 48);
        p3 = (p3 << 16) | (p2 >> 48);
        p2 = (p2 << 16) | (p1 >> 48);
        p1 = (p1 << 16) | (p0 >> 48);
        p0 = (p0 << 16) | (uint64_t)(L & 0xFFFFu);
        c2 = (c2 << 8) | (uint32_t)(c1 >> 56);
        c1 = (c1 << 8) | (c0 >> 56);
        c0 = (c0 << 8) | (uint64_t)cid;
        ++cnt;
    }
    // n < 16 elements (q0.. / d0..) to dp / dc: 8 + 4 + 2 + 1, one store per piece and array
    static __device__ __forceinline__ void pieces(uint16_t *dp, uint8_t *dc, uint32_t n, uint64_t q0, uint64_t q1,
                                                  uint64_t q2, uint64_t q3, uint64_t d0, uint64_t d1) {
        if (n & 8u) {
            put(dp, U128{q0, q1});
            put(dc, d0);
            q0 = q2; q1 = q3; d0 = d1;
            dp += 8; dc += 8;
        }
        if (n & 4u) {
            put(dp, q0);
            put(dc, (uint32_t)d0);
            q0 = q1; d0 >>= 32;
            dp += 4; dc += 4;
        }
        if (n & 2u) {
            put(dp, (uint32_t)q0);
            put(dc, (uint16_t)d0);
            q0 >>= 32; d0 >>= 16;
            dp += 2; dc += 2;
        }
        if (n & 1u) {
            *dp = (uint16_t)q0;
            *dc = (uint8_t)d0;
        }
    }
    // gl = global index of element 0.  If the collector holds an element on a 16-boundary,
    // the group from that boundary up is complete: store it, keep the elements below it.
    __device__ __forceinline__ void flush_group(uint16_t *pml, uint8_t *cid, uint64_t gl) {
        const uint32_t extra = (0u - (uint32_t)gl) & (kFlush - 1);   // elements below the boundary (<= 2)
        if (extra >= cnt) return;
        const bool s1 = extra & 1u, s2 = extra & 2u;
        uint64_t a0 = s1 ? (p0 >> 16) | (p1 << 48) : p0;
        uint64_t a1 = s1 ? (p1 >> 16) | (p2 << 48) : p1;
        uint64_t a2 = s1 ? (p2 >> 16) | (p3 << 48) : p2;
        uint64_t a3 = s1 ? (p3 >> 16) | ((uint64_t)p4 << 48) : p3;
        const uint64_t a4 = s1 ? p4 >> 16 : p4;
        a0 = s2 ? (a0 >> 32) | (a1 << 32) : a0;
        a1 = s2 ? (a1 >> 32) | (a2 << 32) : a1;
        a2 = s2 ? (a2 >> 32) | (a3 << 32) : a2;
        a3 = s2 ? (a3 >> 32) | (a4 << 32) : a3;
        uint64_t b0 = s1 ? (c0 >> 8) | (c1 << 56) : c0;
        uint64_t b1 = s1 ? (c1 >> 8) | ((uint64_t)c2 << 56) : c1;
        const uint64_t b2 = s1 ? c2 >> 8 : c2;
        b0 = s2 ? (b0 >> 16) | (b1 << 48) : b0;
        b1 = s2 ? (b1 >> 16) | (b2 << 48) : b1;
        const uint32_t n = cnt - extra;
        const uint64_t g = gl + extra;
        if (n == kFlush) {
            uint4 *dst = reinterpret_cast<uint4 *>(pml + g);
            dst[0] = make_uint4((uint32_t)a0, (uint32_t)(a0 >> 32), (uint32_t)a1, (uint32_t)(a1 >> 32));
            dst[1] = make_uint4((uint32_t)a2, (uint32_t)(a2 >> 32), (uint32_t)a3, (uint32_t)(a3 >> 32));
            *reinterpret_cast<uint4 *>(cid + g) =
                make_uint4((uint32_t)b0, (uint32_t)(b0 >> 32), (uint32_t)b1, (uint32_t)(b1 >> 32));
        } else {
            pieces(pml + g, cid + g, n, a0, a1, a2, a3, b0, b1);   // the group at the END of the read
        }
        cnt = extra;
    }
    // what is left when the read is done (its first bases, below the last boundary)
    __device__ __forceinline__ void flush_rest(uint16_t *pml, uint8_t *cid, uint64_t gl) {
        if (cnt) pieces(pml + gl, cid + gl, cnt, p0, p1, p2, p3, c0, c1);
        cnt = 0;
    }
};

// ---- line-row kernel (fat_query.hip) ------------------------------------------------
// Cross-lane hand-over through LDS inside ONE wave: the LDS pipeline serves a wave's requests
// in order, so the only thing to enforce is that the compiler keeps the accesses on their side
// of this point (no instruction is emitted).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Every LDS-DMA (global_load_lds) this wave has issued has landed: s_waitcnt vmcnt(0), the
// other counters left alone (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8).
__device__ __forceinline__ void lds_dma_landed() {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    wave_sync();
}

// Read bytes of the line-row kernel: a 64-byte window per lane (piece q of lane L at win[q][L]),
// refilled by the lanes that run low -- four 16-byte LDS-DMA pieces (global_load_lds_dwordx4) from
// a 16-byte aligned address that ride with the trip's row fetches and have landed when they
// have, so a refill never adds a memory round trip of its own.  Persistent lanes are out of step
// with each other, so some lane of a wave refills in nearly every trip; the others take no part
// (their EXEC bit is off: the DMA leaves their window alone).  What matters is the number of
// memory REQUESTS, not of instructions (the texture addresser charges per distinct line of an
// instruction): a lane's bytes are re-read from HBM every time -- the line is long gone from the
// caches when the lane comes back for its next piece -- so 64 bytes per request cost a quarter of
// the line fills of 16-byte top-ups (measured: 94 M requests, 12 GB of HBM reads per C2 launch).
struct LaneWindow {
    uint32_t wb;        // low 32 bits of the global index of the window's first byte
    __device__ __forceinline__ void init(uint64_t g) { wb = (uint32_t)g + 1u; }   // nothing buffered
    // bytes buffered at and below global index g
    __device__ __forceinline__ uint32_t avail(uint64_t g) const {
        const uint32_t d = (uint32_t)g - wb;
        return d < 64u ? d + 1u : 0u;
    }
    // requests [a, a + 64) holding g and at least 48 bytes below it (EXEC-masked by the caller's branch)
    __device__ __forceinline__ void request(uint4 (*win)[64], const uint8_t *bases, uint64_t g) {
        const uint64_t a = g >= 48 ? (g - 48) & ~(uint64_t)15 : 0;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) __builtin_amdgcn_global_load_lds(bases + a + 16 * q, &win[q][0], 16, 0, 0);
        wb = (uint32_t)a;
    }
    __device__ __forceinline__ uint32_t dword(const uint4 (*win)[64], uint32_t lane, uint32_t d) const {
        return reinterpret_cast<const uint32_t *>(&win[(d >> 2) & 3u][lane])[d & 3u];
    }
    // the 8 read bytes ending at g as one little-endian word: byte 7 = base g, byte 0 = base g - 7
    // (bytes below the window are whatever it holds: callers cap what they use by `avail`)
    __device__ __forceinline__ uint64_t get8(const uint4 (*win)[64], uint32_t lane, uint64_t g) const {
        // bytes [e - 8, e) of the window, e = index of g + 1: they start r = e & 3 bytes into dword
        // e / 4 - 2, so each half is one v_alignbyte of two neighbouring dwords (r = 0: the upper
        // dword of each pair is not used, and e / 4 may then be 16 -- any dword will do)
        const uint32_t e = (((uint32_t)g - wb) & 63u) + 1u, d = e >> 2, r = e & 3u;
        const uint32_t x2 = dword(win, lane, d & 15u), x1 = dword(win, lane, (d - 1u) & 15u), x0 = dword(win, lane, (d - 2u) & 15u);
        const uint32_t lo = __builtin_amdgcn_alignbyte(x1, x0, r), hi = __builtin_amdgcn_alignbyte(x2, x1, r);
        return (uint64_t)lo | ((uint64_t)hi << 32);
    }
};

// Outputs of the line-row kernels: lane_out.h (OutRuns).



}  // namespace colbwt
