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
        const uint32_t b = ((uint32_t)g - wb) & 63u, d = b >> 2;
        const uint32_t x2 = dword(win, lane, d), x1 = dword(win, lane, (d - 1u) & 15u), x0 = dword(win, lane, (d - 2u) & 15u);
        const uint32_t sh = 8u * ((b & 3u) + 1u);                     // 8 .. 32
        const uint64_t l64 = (uint64_t)x0 | ((uint64_t)x1 << 32);
        return (l64 >> sh) | ((uint64_t)x2 << (64u - sh));
    }
};

// Outputs of the line-row kernel.  A trip reports a RUN of up to 8 bases: their PML values
// count up by one (col_bwt.hpp:517; a mismatch only ever opens the run, with 0), so the run is
// pushed in one go -- the collector (40 elements: 20 dwords of PML, 10 of col ids; element 0 =
// lowest address = newest) moves up by `cnt` elements through a fixed network of selects and
// byte permutes and the run is OR-ed in as a pattern -- instead of element by element, which a
// wave pays eight times per trip as soon as one lane has a run of 8.  Groups are 32 elements:
// 64 bytes of PML and 32 of col ids are whole HBM write requests (16-element groups left HBM
// 2.8 times the payload in partial writes, profiles/r02a_summary.json).  The aligned group a push
// completes is stored by the wave's flush at the end of the trip; the up to 7 elements below the
// boundary stay.  The struct is generated (tools/gen_collector.py).
struct OutAccRun {
    // generated by tools/gen_collector.py 32; scalar members on purpose: with arrays the compiler turns the
    // select networks below into dynamically indexed scratch (private memory) accesses
    static constexpr uint32_t kGroup = 32;            // elements per flushed group
    uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0, q7 = 0, q8 = 0, q9 = 0, q10 = 0, q11 = 0, q12 = 0, q13 = 0, q14 = 0, q15 = 0, q16 = 0, q17 = 0, q18 = 0, q19 = 0;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0, c8 = 0, c9 = 0;
    uint32_t cnt = 0;

    // n <= 8 elements: PML values l_new - e for element e, col ids = byte e of ids (ids_lo | ids_hi << 32)
    __device__ __forceinline__ void push_run(uint32_t n, uint32_t l_new, uint32_t ids_lo, uint32_t ids_hi) {
        // ---- PML: up by n halfwords = (n >> 1) dwords, then 16 bits
        const bool d4 = n & 8u, d2 = n & 4u, d1 = n & 2u;
        q19 = d4 ? q15 : q19;
        q18 = d4 ? q14 : q18;
        q17 = d4 ? q13 : q17;
        q16 = d4 ? q12 : q16;
        q15 = d4 ? q11 : q15;
        q14 = d4 ? q10 : q14;
        q13 = d4 ? q9 : q13;
        q12 = d4 ? q8 : q12;
        q11 = d4 ? q7 : q11;
        q10 = d4 ? q6 : q10;
        q9 = d4 ? q5 : q9;
        q8 = d4 ? q4 : q8;
        q7 = d4 ? q3 : q7;
        q6 = d4 ? q2 : q6;
        q5 = d4 ? q1 : q5;
        q4 = d4 ? q0 : q4;
        q3 = d4 ? 0u : q3;
        q2 = d4 ? 0u : q2;
        q1 = d4 ? 0u : q1;
        q0 = d4 ? 0u : q0;
        q19 = d2 ? q17 : q19;
        q18 = d2 ? q16 : q18;
        q17 = d2 ? q15 : q17;
        q16 = d2 ? q14 : q16;
        q15 = d2 ? q13 : q15;
        q14 = d2 ? q12 : q14;
        q13 = d2 ? q11 : q13;
        q12 = d2 ? q10 : q12;
        q11 = d2 ? q9 : q11;
        q10 = d2 ? q8 : q10;
        q9 = d2 ? q7 : q9;
        q8 = d2 ? q6 : q8;
        q7 = d2 ? q5 : q7;
        q6 = d2 ? q4 : q6;
        q5 = d2 ? q3 : q5;
        q4 = d2 ? q2 : q4;
        q3 = d2 ? q1 : q3;
        q2 = d2 ? q0 : q2;
        q1 = d2 ? 0u : q1;
        q0 = d2 ? 0u : q0;
        q19 = d1 ? q18 : q19;
        q18 = d1 ? q17 : q18;
        q17 = d1 ? q16 : q17;
        q16 = d1 ? q15 : q16;
        q15 = d1 ? q14 : q15;
        q14 = d1 ? q13 : q14;
        q13 = d1 ? q12 : q13;
        q12 = d1 ? q11 : q12;
        q11 = d1 ? q10 : q11;
        q10 = d1 ? q9 : q10;
        q9 = d1 ? q8 : q9;
        q8 = d1 ? q7 : q8;
        q7 = d1 ? q6 : q7;
        q6 = d1 ? q5 : q6;
        q5 = d1 ? q4 : q5;
        q4 = d1 ? q3 : q4;
        q3 = d1 ? q2 : q3;
        q2 = d1 ? q1 : q2;
        q1 = d1 ? q0 : q1;
        q0 = d1 ? 0u : q0;
        const uint32_t sel16 = (n & 1u) ? 0x05040302u : 0x07060504u;   // {hi, lo} << 16 or hi
        q19 = __builtin_amdgcn_perm(q19, q18, sel16);
        q18 = __builtin_amdgcn_perm(q18, q17, sel16);
        q17 = __builtin_amdgcn_perm(q17, q16, sel16);
        q16 = __builtin_amdgcn_perm(q16, q15, sel16);
        q15 = __builtin_amdgcn_perm(q15, q14, sel16);
        q14 = __builtin_amdgcn_perm(q14, q13, sel16);
        q13 = __builtin_amdgcn_perm(q13, q12, sel16);
        q12 = __builtin_amdgcn_perm(q12, q11, sel16);
        q11 = __builtin_amdgcn_perm(q11, q10, sel16);
        q10 = __builtin_amdgcn_perm(q10, q9, sel16);
        q9 = __builtin_amdgcn_perm(q9, q8, sel16);
        q8 = __builtin_amdgcn_perm(q8, q7, sel16);
        q7 = __builtin_amdgcn_perm(q7, q6, sel16);
        q6 = __builtin_amdgcn_perm(q6, q5, sel16);
        q5 = __builtin_amdgcn_perm(q5, q4, sel16);
        q4 = __builtin_amdgcn_perm(q4, q3, sel16);
        q3 = __builtin_amdgcn_perm(q3, q2, sel16);
        q2 = __builtin_amdgcn_perm(q2, q1, sel16);
        q1 = __builtin_amdgcn_perm(q1, q0, sel16);
        q0 = __builtin_amdgcn_perm(q0, 0u, sel16);
        const uint32_t base = (l_new & 0xFFFFu) | ((l_new - 1u) << 16);   // elements 0 and 1
        q0 |= (base - 0u * 0x00020002u) & (n >= 2u ? 0xFFFFFFFFu : (n == 1u ? 0x0000FFFFu : 0u));
        q1 |= (base - 1u * 0x00020002u) & (n >= 4u ? 0xFFFFFFFFu : (n == 3u ? 0x0000FFFFu : 0u));
        q2 |= (base - 2u * 0x00020002u) & (n >= 6u ? 0xFFFFFFFFu : (n == 5u ? 0x0000FFFFu : 0u));
        q3 |= (base - 3u * 0x00020002u) & (n >= 8u ? 0xFFFFFFFFu : (n == 7u ? 0x0000FFFFu : 0u));
        // ---- col ids: up by n bytes = (n >> 2) dwords, then (n & 3) bytes
        c9 = d4 ? c7 : c9;
        c8 = d4 ? c6 : c8;
        c7 = d4 ? c5 : c7;
        c6 = d4 ? c4 : c6;
        c5 = d4 ? c3 : c5;
        c4 = d4 ? c2 : c4;
        c3 = d4 ? c1 : c3;
        c2 = d4 ? c0 : c2;
        c1 = d4 ? 0u : c1;
        c0 = d4 ? 0u : c0;
        c9 = d2 ? c8 : c9;
        c8 = d2 ? c7 : c8;
        c7 = d2 ? c6 : c7;
        c6 = d2 ? c5 : c6;
        c5 = d2 ? c4 : c5;
        c4 = d2 ? c3 : c4;
        c3 = d2 ? c2 : c3;
        c2 = d2 ? c1 : c2;
        c1 = d2 ? c0 : c1;
        c0 = d2 ? 0u : c0;
        const uint32_t sel8 = 0x07060504u - 0x01010101u * (n & 3u);
        c9 = __builtin_amdgcn_perm(c9, c8, sel8);
        c8 = __builtin_amdgcn_perm(c8, c7, sel8);
        c7 = __builtin_amdgcn_perm(c7, c6, sel8);
        c6 = __builtin_amdgcn_perm(c6, c5, sel8);
        c5 = __builtin_amdgcn_perm(c5, c4, sel8);
        c4 = __builtin_amdgcn_perm(c4, c3, sel8);
        c3 = __builtin_amdgcn_perm(c3, c2, sel8);
        c2 = __builtin_amdgcn_perm(c2, c1, sel8);
        c1 = __builtin_amdgcn_perm(c1, c0, sel8);
        c0 = __builtin_amdgcn_perm(c0, 0u, sel8);
        const uint32_t mlo = n >= 4 ? 0xFFFFFFFFu : (1u << (8 * n)) - 1u;
        const uint32_t mhi = n >= 8 ? 0xFFFFFFFFu : (n > 4 ? (1u << (8 * (n - 4))) - 1u : 0u);
        c0 |= ids_lo & mlo;
        c1 |= ids_hi & mhi;
        cnt += n;
    }
    // n < kGroup elements (a0.. / b0..) to dp / dc in pieces of 16, .., 2, 1 elements, one (unaligned) store
    // per piece and array; consumes its arguments
    static __device__ __forceinline__ void pieces(uint16_t *dp, uint8_t *dc, uint32_t n, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7, uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t a12, uint32_t a13, uint32_t a14, uint32_t a15, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7) {
        if (n & 16u) {
            put(dp + 0, Q128{a0, a1, a2, a3});
            put(dp + 8, Q128{a4, a5, a6, a7});
            put(dc + 0, Q128{b0, b1, b2, b3});
            a0 = a8;
            a1 = a9;
            a2 = a10;
            a3 = a11;
            a4 = a12;
            a5 = a13;
            a6 = a14;
            a7 = a15;
            b0 = b4;
            b1 = b5;
            b2 = b6;
            b3 = b7;
            dp += 16; dc += 16;
        }
        if (n & 8u) {
            put(dp + 0, Q128{a0, a1, a2, a3});
            put(dc, (uint64_t)b0 | ((uint64_t)b1 << 32));
            a0 = a4;
            a1 = a5;
            a2 = a6;
            a3 = a7;
            a4 = a8;
            a5 = a9;
            a6 = a10;
            a7 = a11;
            a8 = a12;
            a9 = a13;
            a10 = a14;
            a11 = a15;
            b0 = b2;
            b1 = b3;
            b2 = b4;
            b3 = b5;
            b4 = b6;
            b5 = b7;
            dp += 8; dc += 8;
        }
        if (n & 4u) {
            put(dp, (uint64_t)a0 | ((uint64_t)a1 << 32));
            put(dc, b0);
            a0 = a2; a1 = a3; b0 = b1;
            dp += 4; dc += 4;
        }
        if (n & 2u) {
            put(dp, a0);
            put(dc, (uint16_t)b0);
            a0 = a1; b0 >>= 16;
            dp += 2; dc += 2;
        }
        if (n & 1u) {
            *dp = (uint16_t)a0;
            *dc = (uint8_t)b0;
        }
    }
    // gl = global index of element 0.  If the collector holds an element on a group boundary, the
    // group from that boundary up is complete (or is the ragged top of a chunk): store it, keep the
    // elements below it.
    __device__ __forceinline__ void flush_group(uint16_t *pml, uint8_t *cid, uint64_t gl) {
        const uint32_t extra = (0u - (uint32_t)gl) & (kGroup - 1) & 7u;   // elements below the boundary
        if (((0u - (uint32_t)gl) & (kGroup - 1)) >= 8u || extra >= cnt) return;   // no boundary within reach
        // the group from the boundary up: drop `extra` elements (halfwords of q, bytes of c)
        const bool e4 = extra & 4u, e2 = extra & 2u;
        const uint32_t t0 = e4 ? q2 : q0;
        const uint32_t t1 = e4 ? q3 : q1;
        const uint32_t t2 = e4 ? q4 : q2;
        const uint32_t t3 = e4 ? q5 : q3;
        const uint32_t t4 = e4 ? q6 : q4;
        const uint32_t t5 = e4 ? q7 : q5;
        const uint32_t t6 = e4 ? q8 : q6;
        const uint32_t t7 = e4 ? q9 : q7;
        const uint32_t t8 = e4 ? q10 : q8;
        const uint32_t t9 = e4 ? q11 : q9;
        const uint32_t t10 = e4 ? q12 : q10;
        const uint32_t t11 = e4 ? q13 : q11;
        const uint32_t t12 = e4 ? q14 : q12;
        const uint32_t t13 = e4 ? q15 : q13;
        const uint32_t t14 = e4 ? q16 : q14;
        const uint32_t t15 = e4 ? q17 : q15;
        const uint32_t t16 = e4 ? q18 : q16;
        const uint32_t t17 = e4 ? q19 : q17;
        uint32_t a0 = e2 ? t1 : t0;
        uint32_t a1 = e2 ? t2 : t1;
        uint32_t a2 = e2 ? t3 : t2;
        uint32_t a3 = e2 ? t4 : t3;
        uint32_t a4 = e2 ? t5 : t4;
        uint32_t a5 = e2 ? t6 : t5;
        uint32_t a6 = e2 ? t7 : t6;
        uint32_t a7 = e2 ? t8 : t7;
        uint32_t a8 = e2 ? t9 : t8;
        uint32_t a9 = e2 ? t10 : t9;
        uint32_t a10 = e2 ? t11 : t10;
        uint32_t a11 = e2 ? t12 : t11;
        uint32_t a12 = e2 ? t13 : t12;
        uint32_t a13 = e2 ? t14 : t13;
        uint32_t a14 = e2 ? t15 : t14;
        uint32_t a15 = e2 ? t16 : t15;
        uint32_t a16 = e2 ? t17 : t16;
        const uint32_t sel16 = (extra & 1u) ? 0x05040302u : 0x03020100u;   // {hi, lo} >> 16 or lo
        a0 = __builtin_amdgcn_perm(a1, a0, sel16);
        a1 = __builtin_amdgcn_perm(a2, a1, sel16);
        a2 = __builtin_amdgcn_perm(a3, a2, sel16);
        a3 = __builtin_amdgcn_perm(a4, a3, sel16);
        a4 = __builtin_amdgcn_perm(a5, a4, sel16);
        a5 = __builtin_amdgcn_perm(a6, a5, sel16);
        a6 = __builtin_amdgcn_perm(a7, a6, sel16);
        a7 = __builtin_amdgcn_perm(a8, a7, sel16);
        a8 = __builtin_amdgcn_perm(a9, a8, sel16);
        a9 = __builtin_amdgcn_perm(a10, a9, sel16);
        a10 = __builtin_amdgcn_perm(a11, a10, sel16);
        a11 = __builtin_amdgcn_perm(a12, a11, sel16);
        a12 = __builtin_amdgcn_perm(a13, a12, sel16);
        a13 = __builtin_amdgcn_perm(a14, a13, sel16);
        a14 = __builtin_amdgcn_perm(a15, a14, sel16);
        a15 = __builtin_amdgcn_perm(a16, a15, sel16);
        uint32_t b0 = e4 ? c1 : c0;
        uint32_t b1 = e4 ? c2 : c1;
        uint32_t b2 = e4 ? c3 : c2;
        uint32_t b3 = e4 ? c4 : c3;
        uint32_t b4 = e4 ? c5 : c4;
        uint32_t b5 = e4 ? c6 : c5;
        uint32_t b6 = e4 ? c7 : c6;
        uint32_t b7 = e4 ? c8 : c7;
        uint32_t b8 = e4 ? c9 : c8;
        const uint32_t sel8 = 0x03020100u + 0x01010101u * (extra & 3u);
        b0 = __builtin_amdgcn_perm(b1, b0, sel8);
        b1 = __builtin_amdgcn_perm(b2, b1, sel8);
        b2 = __builtin_amdgcn_perm(b3, b2, sel8);
        b3 = __builtin_amdgcn_perm(b4, b3, sel8);
        b4 = __builtin_amdgcn_perm(b5, b4, sel8);
        b5 = __builtin_amdgcn_perm(b6, b5, sel8);
        b6 = __builtin_amdgcn_perm(b7, b6, sel8);
        b7 = __builtin_amdgcn_perm(b8, b7, sel8);
        const uint32_t n = cnt - extra;
        const uint64_t g = gl + extra;
        if (n == kGroup) {
            reinterpret_cast<uint4 *>(pml + g)[0] = make_uint4(a0, a1, a2, a3);
            reinterpret_cast<uint4 *>(pml + g)[1] = make_uint4(a4, a5, a6, a7);
            reinterpret_cast<uint4 *>(pml + g)[2] = make_uint4(a8, a9, a10, a11);
            reinterpret_cast<uint4 *>(pml + g)[3] = make_uint4(a12, a13, a14, a15);
            reinterpret_cast<uint4 *>(cid + g)[0] = make_uint4(b0, b1, b2, b3);
            reinterpret_cast<uint4 *>(cid + g)[1] = make_uint4(b4, b5, b6, b7);
        } else {
            pieces(pml + g, cid + g, n, a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, b0, b1, b2, b3, b4, b5, b6, b7);
        }
        cnt = extra;
    }
    // The same flush for a whole wave at once (every lane calls it; `active` lanes take part): a
    // lane completes a group only every tenth trip or so, but some lane of the wave does in nearly
    // every trip, and a vector-memory instruction costs the texture addresser about the same with
    // 6 lanes as with 64.  So the complete groups of the trip are parked in LDS (`scratch`, 6.5 KB
    // of the wave's own) and written by all 64 lanes together, 16 bytes each: one store instruction
    // per 10 groups.  Ragged groups (the top of a chunk whose end is not aligned) go out in pieces.
    __device__ __forceinline__ void flush_group_wave(uint16_t *pml, uint8_t *cid, uint64_t gl, bool active, uint4 *scratch,
                                                     uint32_t lane) {
        const uint32_t below = (0u - (uint32_t)gl) & (kGroup - 1);   // elements below the boundary
        const uint32_t extra = below & 7u;
        const bool has = active && below < 8u && extra < cnt;
        const uint32_t n = has ? cnt - extra : 0u;
        if (!__any(has)) return;
        // the group from the boundary up: drop `extra` elements (halfwords of q, bytes of c)
        const bool e4 = extra & 4u, e2 = extra & 2u;
        const uint32_t t0 = e4 ? q2 : q0;
        const uint32_t t1 = e4 ? q3 : q1;
        const uint32_t t2 = e4 ? q4 : q2;
        const uint32_t t3 = e4 ? q5 : q3;
        const uint32_t t4 = e4 ? q6 : q4;
        const uint32_t t5 = e4 ? q7 : q5;
        const uint32_t t6 = e4 ? q8 : q6;
        const uint32_t t7 = e4 ? q9 : q7;
        const uint32_t t8 = e4 ? q10 : q8;
        const uint32_t t9 = e4 ? q11 : q9;
        const uint32_t t10 = e4 ? q12 : q10;
        const uint32_t t11 = e4 ? q13 : q11;
        const uint32_t t12 = e4 ? q14 : q12;
        const uint32_t t13 = e4 ? q15 : q13;
        const uint32_t t14 = e4 ? q16 : q14;
        const uint32_t t15 = e4 ? q17 : q15;
        const uint32_t t16 = e4 ? q18 : q16;
        const uint32_t t17 = e4 ? q19 : q17;
        uint32_t a0 = e2 ? t1 : t0;
        uint32_t a1 = e2 ? t2 : t1;
        uint32_t a2 = e2 ? t3 : t2;
        uint32_t a3 = e2 ? t4 : t3;
        uint32_t a4 = e2 ? t5 : t4;
        uint32_t a5 = e2 ? t6 : t5;
        uint32_t a6 = e2 ? t7 : t6;
        uint32_t a7 = e2 ? t8 : t7;
        uint32_t a8 = e2 ? t9 : t8;
        uint32_t a9 = e2 ? t10 : t9;
        uint32_t a10 = e2 ? t11 : t10;
        uint32_t a11 = e2 ? t12 : t11;
        uint32_t a12 = e2 ? t13 : t12;
        uint32_t a13 = e2 ? t14 : t13;
        uint32_t a14 = e2 ? t15 : t14;
        uint32_t a15 = e2 ? t16 : t15;
        uint32_t a16 = e2 ? t17 : t16;
        const uint32_t sel16 = (extra & 1u) ? 0x05040302u : 0x03020100u;   // {hi, lo} >> 16 or lo
        a0 = __builtin_amdgcn_perm(a1, a0, sel16);
        a1 = __builtin_amdgcn_perm(a2, a1, sel16);
        a2 = __builtin_amdgcn_perm(a3, a2, sel16);
        a3 = __builtin_amdgcn_perm(a4, a3, sel16);
        a4 = __builtin_amdgcn_perm(a5, a4, sel16);
        a5 = __builtin_amdgcn_perm(a6, a5, sel16);
        a6 = __builtin_amdgcn_perm(a7, a6, sel16);
        a7 = __builtin_amdgcn_perm(a8, a7, sel16);
        a8 = __builtin_amdgcn_perm(a9, a8, sel16);
        a9 = __builtin_amdgcn_perm(a10, a9, sel16);
        a10 = __builtin_amdgcn_perm(a11, a10, sel16);
        a11 = __builtin_amdgcn_perm(a12, a11, sel16);
        a12 = __builtin_amdgcn_perm(a13, a12, sel16);
        a13 = __builtin_amdgcn_perm(a14, a13, sel16);
        a14 = __builtin_amdgcn_perm(a15, a14, sel16);
        a15 = __builtin_amdgcn_perm(a16, a15, sel16);
        uint32_t b0 = e4 ? c1 : c0;
        uint32_t b1 = e4 ? c2 : c1;
        uint32_t b2 = e4 ? c3 : c2;
        uint32_t b3 = e4 ? c4 : c3;
        uint32_t b4 = e4 ? c5 : c4;
        uint32_t b5 = e4 ? c6 : c5;
        uint32_t b6 = e4 ? c7 : c6;
        uint32_t b7 = e4 ? c8 : c7;
        uint32_t b8 = e4 ? c9 : c8;
        const uint32_t sel8 = 0x03020100u + 0x01010101u * (extra & 3u);
        b0 = __builtin_amdgcn_perm(b1, b0, sel8);
        b1 = __builtin_amdgcn_perm(b2, b1, sel8);
        b2 = __builtin_amdgcn_perm(b3, b2, sel8);
        b3 = __builtin_amdgcn_perm(b4, b3, sel8);
        b4 = __builtin_amdgcn_perm(b5, b4, sel8);
        b5 = __builtin_amdgcn_perm(b6, b5, sel8);
        b6 = __builtin_amdgcn_perm(b7, b6, sel8);
        b7 = __builtin_amdgcn_perm(b8, b7, sel8);
        const uint64_t g = gl + extra;
        if (has && n != kGroup) pieces(pml + g, cid + g, n, a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, b0, b1, b2, b3, b4, b5, b6, b7);   // the group at the END of a chunk
        if (has) cnt = extra;
        const bool full = n == kGroup;
        const unsigned long long mask = __ballot(full);
        if (mask == 0) return;
        uint64_t *const where = reinterpret_cast<uint64_t *>(scratch + 6 * 64);
        if (full) {
            const uint32_t rank = (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            scratch[6 * rank + 0] = make_uint4(a0, a1, a2, a3);
            scratch[6 * rank + 1] = make_uint4(a4, a5, a6, a7);
            scratch[6 * rank + 2] = make_uint4(a8, a9, a10, a11);
            scratch[6 * rank + 3] = make_uint4(a12, a13, a14, a15);
            scratch[6 * rank + 4] = make_uint4(b0, b1, b2, b3);
            scratch[6 * rank + 5] = make_uint4(b4, b5, b6, b7);
            where[rank] = g;
        }
        wave_sync();
        const uint32_t pieces16 = 6u * (uint32_t)__builtin_popcountll(mask);
        for (uint32_t t = lane; t < pieces16; t += 64) {
            const uint32_t item = t / 6u, part = t - 6u * item;
            const uint64_t gi = where[item];
            uint8_t *dst = part >= 4u ? cid + gi + 16u * (part - 4u) : reinterpret_cast<uint8_t *>(pml + gi) + 16u * part;
            *reinterpret_cast<uint4 *>(dst) = scratch[t];
        }
        wave_sync();
    }
    // what is left when a chunk is done (its first bases, below the last boundary)
    __device__ __forceinline__ void flush_rest(uint16_t *pml, uint8_t *cid, uint64_t gl) {
        if (cnt) pieces(pml + gl, cid + gl, cnt, q0, q1, q2, q3, q4, q5, q6, q7, q8, q9, q10, q11, q12, q13, q14, q15, c0, c1, c2, c3, c4, c5, c6, c7);
        cnt = 0;
    }
};


}  // namespace colbwt
