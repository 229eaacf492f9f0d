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
// wave pays eight times per trip as soon as one lane has a run of 8.  Both arrays leave in groups
// of 64 bytes -- 32 PML values, 64 col ids -- which are whole HBM write requests: random 32-byte
// stores run at 0.7 TB/s, 64-byte ones at 3.1 (tools/scatter_bench.hip), and 16-element groups left
// HBM 2.8 times the payload in partial writes (profiles/r02a_summary.json).  So there are two
// collectors, one per array, pushed together and flushed each at its own boundaries.  The aligned group a push
// completes is stored by the wave's flush at the end of the trip; the up to 7 elements below the
// boundary stay.  The struct is generated (tools/gen_collector.py).
struct OutAccPml {
    // generated by tools/gen_collector.py; scalar members on purpose: with arrays the compiler turns the
    // select networks below into dynamically indexed scratch (private memory) accesses
    static constexpr uint32_t kGroup = 32;            // elements per flushed group (64 bytes)
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0, r8 = 0, r9 = 0, r10 = 0, r11 = 0, r12 = 0, r13 = 0, r14 = 0, r15 = 0, r16 = 0, r17 = 0, r18 = 0, r19 = 0;
    uint32_t cnt = 0;

    // n <= 8 elements: values l_new - e for element e; keep / keep1 = 0 clear elements 0-1 / 2-3 instead
    // (what a mismatch entry reports: (0, 0), or (1, 0, 0) = push_run(3, 1, ~0u, 0))
    __device__ __forceinline__ void push_run(uint32_t n, uint32_t l_new, uint32_t keep = 0xFFFFFFFFu, uint32_t keep1 = 0xFFFFFFFFu) {
        // up by n halfwords = (n >> 1) dwords, then 16 bits
        const bool d4 = n & 8u, d2 = n & 4u, d1 = n & 2u;
        r19 = d4 ? r15 : r19;
        r18 = d4 ? r14 : r18;
        r17 = d4 ? r13 : r17;
        r16 = d4 ? r12 : r16;
        r15 = d4 ? r11 : r15;
        r14 = d4 ? r10 : r14;
        r13 = d4 ? r9 : r13;
        r12 = d4 ? r8 : r12;
        r11 = d4 ? r7 : r11;
        r10 = d4 ? r6 : r10;
        r9 = d4 ? r5 : r9;
        r8 = d4 ? r4 : r8;
        r7 = d4 ? r3 : r7;
        r6 = d4 ? r2 : r6;
        r5 = d4 ? r1 : r5;
        r4 = d4 ? r0 : r4;
        r3 = d4 ? 0u : r3;
        r2 = d4 ? 0u : r2;
        r1 = d4 ? 0u : r1;
        r0 = d4 ? 0u : r0;
        r19 = d2 ? r17 : r19;
        r18 = d2 ? r16 : r18;
        r17 = d2 ? r15 : r17;
        r16 = d2 ? r14 : r16;
        r15 = d2 ? r13 : r15;
        r14 = d2 ? r12 : r14;
        r13 = d2 ? r11 : r13;
        r12 = d2 ? r10 : r12;
        r11 = d2 ? r9 : r11;
        r10 = d2 ? r8 : r10;
        r9 = d2 ? r7 : r9;
        r8 = d2 ? r6 : r8;
        r7 = d2 ? r5 : r7;
        r6 = d2 ? r4 : r6;
        r5 = d2 ? r3 : r5;
        r4 = d2 ? r2 : r4;
        r3 = d2 ? r1 : r3;
        r2 = d2 ? r0 : r2;
        r1 = d2 ? 0u : r1;
        r0 = d2 ? 0u : r0;
        r19 = d1 ? r18 : r19;
        r18 = d1 ? r17 : r18;
        r17 = d1 ? r16 : r17;
        r16 = d1 ? r15 : r16;
        r15 = d1 ? r14 : r15;
        r14 = d1 ? r13 : r14;
        r13 = d1 ? r12 : r13;
        r12 = d1 ? r11 : r12;
        r11 = d1 ? r10 : r11;
        r10 = d1 ? r9 : r10;
        r9 = d1 ? r8 : r9;
        r8 = d1 ? r7 : r8;
        r7 = d1 ? r6 : r7;
        r6 = d1 ? r5 : r6;
        r5 = d1 ? r4 : r5;
        r4 = d1 ? r3 : r4;
        r3 = d1 ? r2 : r3;
        r2 = d1 ? r1 : r2;
        r1 = d1 ? r0 : r1;
        r0 = d1 ? 0u : r0;
        const uint32_t sel16 = (n & 1u) ? 0x05040302u : 0x07060504u;   // {hi, lo} << 16 or hi
        r19 = __builtin_amdgcn_perm(r19, r18, sel16);
        r18 = __builtin_amdgcn_perm(r18, r17, sel16);
        r17 = __builtin_amdgcn_perm(r17, r16, sel16);
        r16 = __builtin_amdgcn_perm(r16, r15, sel16);
        r15 = __builtin_amdgcn_perm(r15, r14, sel16);
        r14 = __builtin_amdgcn_perm(r14, r13, sel16);
        r13 = __builtin_amdgcn_perm(r13, r12, sel16);
        r12 = __builtin_amdgcn_perm(r12, r11, sel16);
        r11 = __builtin_amdgcn_perm(r11, r10, sel16);
        r10 = __builtin_amdgcn_perm(r10, r9, sel16);
        r9 = __builtin_amdgcn_perm(r9, r8, sel16);
        r8 = __builtin_amdgcn_perm(r8, r7, sel16);
        r7 = __builtin_amdgcn_perm(r7, r6, sel16);
        r6 = __builtin_amdgcn_perm(r6, r5, sel16);
        r5 = __builtin_amdgcn_perm(r5, r4, sel16);
        r4 = __builtin_amdgcn_perm(r4, r3, sel16);
        r3 = __builtin_amdgcn_perm(r3, r2, sel16);
        r2 = __builtin_amdgcn_perm(r2, r1, sel16);
        r1 = __builtin_amdgcn_perm(r1, r0, sel16);
        r0 = __builtin_amdgcn_perm(r0, 0u, sel16);
        const uint32_t base = (l_new & 0xFFFFu) | ((l_new - 1u) << 16);   // elements 0 and 1
        r0 |= (base - 0u * 0x00020002u) & (n >= 2u ? 0xFFFFFFFFu : (n == 1u ? 0x0000FFFFu : 0u)) & keep;
        r1 |= (base - 1u * 0x00020002u) & (n >= 4u ? 0xFFFFFFFFu : (n == 3u ? 0x0000FFFFu : 0u)) & keep1;
        r2 |= (base - 2u * 0x00020002u) & (n >= 6u ? 0xFFFFFFFFu : (n == 5u ? 0x0000FFFFu : 0u));
        r3 |= (base - 3u * 0x00020002u) & (n >= 8u ? 0xFFFFFFFFu : (n == 7u ? 0x0000FFFFu : 0u));
        cnt += n;
    }
    // n < kGroup elements (a0..) to d in pieces of 16, .., 2, 1 elements, one (unaligned) store per piece; consumes
    // its arguments
    static __device__ __forceinline__ void pieces(uint16_t *d, uint32_t n, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7, uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t a12, uint32_t a13, uint32_t a14, uint32_t a15) {
        if (n & 16u) {
            put(d + 0, Q128{a0, a1, a2, a3});
            put(d + 8, Q128{a4, a5, a6, a7});
            a0 = a8;
            a1 = a9;
            a2 = a10;
            a3 = a11;
            a4 = a12;
            a5 = a13;
            a6 = a14;
            a7 = a15;
            d += 16;
        }
        if (n & 8u) {
            put(d + 0, Q128{a0, a1, a2, a3});
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
            d += 8;
        }
        if (n & 4u) {
            put(d, (uint64_t)a0 | ((uint64_t)a1 << 32));
            a0 = a2; a1 = a3;
            d += 4;
        }
        if (n & 2u) {
            put(d, a0);
            a0 = a1;
            d += 2;
        }
        if (n & 1u) *d = (uint16_t)a0;
    }
    // gl = global index of element 0.  If the collector holds an element on a group boundary, the
    // group from that boundary up is complete (or is the ragged top of a chunk): store it, keep the
    // elements below it.
    __device__ __forceinline__ void flush_group(uint16_t *out, uint64_t gl) {
        const uint32_t extra = (0u - (uint32_t)gl) & (kGroup - 1) & 7u;   // elements below the boundary
        if (((0u - (uint32_t)gl) & (kGroup - 1)) >= 8u || extra >= cnt) return;   // no boundary within reach
        // the group from the boundary up: drop `extra` elements from the bottom
        const bool e4 = extra & 4u, e2 = extra & 2u;
        const uint32_t t0 = e4 ? r2 : r0;
        const uint32_t t1 = e4 ? r3 : r1;
        const uint32_t t2 = e4 ? r4 : r2;
        const uint32_t t3 = e4 ? r5 : r3;
        const uint32_t t4 = e4 ? r6 : r4;
        const uint32_t t5 = e4 ? r7 : r5;
        const uint32_t t6 = e4 ? r8 : r6;
        const uint32_t t7 = e4 ? r9 : r7;
        const uint32_t t8 = e4 ? r10 : r8;
        const uint32_t t9 = e4 ? r11 : r9;
        const uint32_t t10 = e4 ? r12 : r10;
        const uint32_t t11 = e4 ? r13 : r11;
        const uint32_t t12 = e4 ? r14 : r12;
        const uint32_t t13 = e4 ? r15 : r13;
        const uint32_t t14 = e4 ? r16 : r14;
        const uint32_t t15 = e4 ? r17 : r15;
        const uint32_t t16 = e4 ? r18 : r16;
        const uint32_t t17 = e4 ? r19 : r17;
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
        const uint32_t sel = (extra & 1u) ? 0x05040302u : 0x03020100u;   // {hi, lo} >> 16 or lo
        a0 = __builtin_amdgcn_perm(a1, a0, sel);
        a1 = __builtin_amdgcn_perm(a2, a1, sel);
        a2 = __builtin_amdgcn_perm(a3, a2, sel);
        a3 = __builtin_amdgcn_perm(a4, a3, sel);
        a4 = __builtin_amdgcn_perm(a5, a4, sel);
        a5 = __builtin_amdgcn_perm(a6, a5, sel);
        a6 = __builtin_amdgcn_perm(a7, a6, sel);
        a7 = __builtin_amdgcn_perm(a8, a7, sel);
        a8 = __builtin_amdgcn_perm(a9, a8, sel);
        a9 = __builtin_amdgcn_perm(a10, a9, sel);
        a10 = __builtin_amdgcn_perm(a11, a10, sel);
        a11 = __builtin_amdgcn_perm(a12, a11, sel);
        a12 = __builtin_amdgcn_perm(a13, a12, sel);
        a13 = __builtin_amdgcn_perm(a14, a13, sel);
        a14 = __builtin_amdgcn_perm(a15, a14, sel);
        a15 = __builtin_amdgcn_perm(a16, a15, sel);
        const uint32_t n = cnt - extra;
        const uint64_t g = gl + extra;
        if (n == kGroup) {
            reinterpret_cast<uint4 *>(out + g)[0] = make_uint4(a0, a1, a2, a3);
            reinterpret_cast<uint4 *>(out + g)[1] = make_uint4(a4, a5, a6, a7);
            reinterpret_cast<uint4 *>(out + g)[2] = make_uint4(a8, a9, a10, a11);
            reinterpret_cast<uint4 *>(out + g)[3] = make_uint4(a12, a13, a14, a15);
        } else {
            pieces(out + g, n, a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15);
        }
        cnt = extra;
    }
    // The same flush for a whole wave at once (every lane calls it; `active` lanes take part): a
    // lane completes a group only every tenth trip or so, but some lane of the wave does in nearly
    // every trip.  The complete groups of the trip are parked in LDS (`scratch`, 4.5 KB of the wave's
    // own) and written by all 64 lanes together, 16 bytes each: one store instruction per 16 groups.
    // Ragged groups (the top of a chunk whose end is not aligned) go out in pieces.
    __device__ __forceinline__ void flush_group_wave(uint16_t *out, uint64_t gl, bool active, uint4 *scratch, uint32_t lane) {
        const uint32_t below = (0u - (uint32_t)gl) & (kGroup - 1);   // elements below the boundary
        const uint32_t extra = below & 7u;
        const bool has = active && below < 8u && extra < cnt;
        const uint32_t n = has ? cnt - extra : 0u;
        if (!__any(has)) return;
        // the group from the boundary up: drop `extra` elements from the bottom
        const bool e4 = extra & 4u, e2 = extra & 2u;
        const uint32_t t0 = e4 ? r2 : r0;
        const uint32_t t1 = e4 ? r3 : r1;
        const uint32_t t2 = e4 ? r4 : r2;
        const uint32_t t3 = e4 ? r5 : r3;
        const uint32_t t4 = e4 ? r6 : r4;
        const uint32_t t5 = e4 ? r7 : r5;
        const uint32_t t6 = e4 ? r8 : r6;
        const uint32_t t7 = e4 ? r9 : r7;
        const uint32_t t8 = e4 ? r10 : r8;
        const uint32_t t9 = e4 ? r11 : r9;
        const uint32_t t10 = e4 ? r12 : r10;
        const uint32_t t11 = e4 ? r13 : r11;
        const uint32_t t12 = e4 ? r14 : r12;
        const uint32_t t13 = e4 ? r15 : r13;
        const uint32_t t14 = e4 ? r16 : r14;
        const uint32_t t15 = e4 ? r17 : r15;
        const uint32_t t16 = e4 ? r18 : r16;
        const uint32_t t17 = e4 ? r19 : r17;
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
        const uint32_t sel = (extra & 1u) ? 0x05040302u : 0x03020100u;   // {hi, lo} >> 16 or lo
        a0 = __builtin_amdgcn_perm(a1, a0, sel);
        a1 = __builtin_amdgcn_perm(a2, a1, sel);
        a2 = __builtin_amdgcn_perm(a3, a2, sel);
        a3 = __builtin_amdgcn_perm(a4, a3, sel);
        a4 = __builtin_amdgcn_perm(a5, a4, sel);
        a5 = __builtin_amdgcn_perm(a6, a5, sel);
        a6 = __builtin_amdgcn_perm(a7, a6, sel);
        a7 = __builtin_amdgcn_perm(a8, a7, sel);
        a8 = __builtin_amdgcn_perm(a9, a8, sel);
        a9 = __builtin_amdgcn_perm(a10, a9, sel);
        a10 = __builtin_amdgcn_perm(a11, a10, sel);
        a11 = __builtin_amdgcn_perm(a12, a11, sel);
        a12 = __builtin_amdgcn_perm(a13, a12, sel);
        a13 = __builtin_amdgcn_perm(a14, a13, sel);
        a14 = __builtin_amdgcn_perm(a15, a14, sel);
        a15 = __builtin_amdgcn_perm(a16, a15, sel);
        const uint64_t g = gl + extra;
        if (has && n != kGroup) pieces(out + g, n, a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15);   // the group at the END of a chunk
        if (has) cnt = extra;
        const bool full = n == kGroup;
        const unsigned long long mask = __ballot(full);
        if (mask == 0) return;
        uint64_t *const where = reinterpret_cast<uint64_t *>(scratch + 4 * 64);
        if (full) {
            const uint32_t rank = (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            scratch[4 * rank + 0] = make_uint4(a0, a1, a2, a3);
            scratch[4 * rank + 1] = make_uint4(a4, a5, a6, a7);
            scratch[4 * rank + 2] = make_uint4(a8, a9, a10, a11);
            scratch[4 * rank + 3] = make_uint4(a12, a13, a14, a15);
            where[rank] = g;
        }
        wave_sync();
        const uint32_t pieces16 = 4u * (uint32_t)__builtin_popcountll(mask);
        for (uint32_t t = lane; t < pieces16; t += 64)
            *reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(out + where[t / 4u]) + 16u * (t % 4u)) = scratch[t];
        wave_sync();
    }
    // what is left when a chunk is done (its first bases, below the last boundary)
    __device__ __forceinline__ void flush_rest(uint16_t *out, uint64_t gl) {
        if (cnt) pieces(out + gl, cnt, r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15);
        cnt = 0;
    }
};

struct OutAccCid {
    // generated by tools/gen_collector.py; scalar members on purpose: with arrays the compiler turns the
    // select networks below into dynamically indexed scratch (private memory) accesses
    static constexpr uint32_t kGroup = 64;            // elements per flushed group (64 bytes)
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0, r8 = 0, r9 = 0, r10 = 0, r11 = 0, r12 = 0, r13 = 0, r14 = 0, r15 = 0, r16 = 0, r17 = 0;
    uint32_t cnt = 0;

    // n <= 8 elements: byte e of ids (ids_lo | ids_hi << 32) for element e
    __device__ __forceinline__ void push_run(uint32_t n, uint32_t ids_lo, uint32_t ids_hi) {
        // up by n bytes = (n >> 2) dwords, then (n & 3) bytes
        const bool d4 = n & 8u, d2 = n & 4u;
        r17 = d4 ? r15 : r17;
        r16 = d4 ? r14 : r16;
        r15 = d4 ? r13 : r15;
        r14 = d4 ? r12 : r14;
        r13 = d4 ? r11 : r13;
        r12 = d4 ? r10 : r12;
        r11 = d4 ? r9 : r11;
        r10 = d4 ? r8 : r10;
        r9 = d4 ? r7 : r9;
        r8 = d4 ? r6 : r8;
        r7 = d4 ? r5 : r7;
        r6 = d4 ? r4 : r6;
        r5 = d4 ? r3 : r5;
        r4 = d4 ? r2 : r4;
        r3 = d4 ? r1 : r3;
        r2 = d4 ? r0 : r2;
        r1 = d4 ? 0u : r1;
        r0 = d4 ? 0u : r0;
        r17 = d2 ? r16 : r17;
        r16 = d2 ? r15 : r16;
        r15 = d2 ? r14 : r15;
        r14 = d2 ? r13 : r14;
        r13 = d2 ? r12 : r13;
        r12 = d2 ? r11 : r12;
        r11 = d2 ? r10 : r11;
        r10 = d2 ? r9 : r10;
        r9 = d2 ? r8 : r9;
        r8 = d2 ? r7 : r8;
        r7 = d2 ? r6 : r7;
        r6 = d2 ? r5 : r6;
        r5 = d2 ? r4 : r5;
        r4 = d2 ? r3 : r4;
        r3 = d2 ? r2 : r3;
        r2 = d2 ? r1 : r2;
        r1 = d2 ? r0 : r1;
        r0 = d2 ? 0u : r0;
        const uint32_t sel8 = 0x07060504u - 0x01010101u * (n & 3u);
        r17 = __builtin_amdgcn_perm(r17, r16, sel8);
        r16 = __builtin_amdgcn_perm(r16, r15, sel8);
        r15 = __builtin_amdgcn_perm(r15, r14, sel8);
        r14 = __builtin_amdgcn_perm(r14, r13, sel8);
        r13 = __builtin_amdgcn_perm(r13, r12, sel8);
        r12 = __builtin_amdgcn_perm(r12, r11, sel8);
        r11 = __builtin_amdgcn_perm(r11, r10, sel8);
        r10 = __builtin_amdgcn_perm(r10, r9, sel8);
        r9 = __builtin_amdgcn_perm(r9, r8, sel8);
        r8 = __builtin_amdgcn_perm(r8, r7, sel8);
        r7 = __builtin_amdgcn_perm(r7, r6, sel8);
        r6 = __builtin_amdgcn_perm(r6, r5, sel8);
        r5 = __builtin_amdgcn_perm(r5, r4, sel8);
        r4 = __builtin_amdgcn_perm(r4, r3, sel8);
        r3 = __builtin_amdgcn_perm(r3, r2, sel8);
        r2 = __builtin_amdgcn_perm(r2, r1, sel8);
        r1 = __builtin_amdgcn_perm(r1, r0, sel8);
        r0 = __builtin_amdgcn_perm(r0, 0u, sel8);
        const uint32_t mlo = n >= 4 ? 0xFFFFFFFFu : (1u << (8 * n)) - 1u;
        const uint32_t mhi = n >= 8 ? 0xFFFFFFFFu : (n > 4 ? (1u << (8 * (n - 4))) - 1u : 0u);
        r0 |= ids_lo & mlo;
        r1 |= ids_hi & mhi;
        cnt += n;
    }
    // n < kGroup elements (a0..) to d in pieces of 32, .., 2, 1 elements, one (unaligned) store per piece; consumes
    // its arguments
    static __device__ __forceinline__ void pieces(uint8_t *d, uint32_t n, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7, uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t a12, uint32_t a13, uint32_t a14, uint32_t a15) {
        if (n & 32u) {
            put(d + 0, Q128{a0, a1, a2, a3});
            put(d + 16, Q128{a4, a5, a6, a7});
            a0 = a8;
            a1 = a9;
            a2 = a10;
            a3 = a11;
            a4 = a12;
            a5 = a13;
            a6 = a14;
            a7 = a15;
            d += 32;
        }
        if (n & 16u) {
            put(d + 0, Q128{a0, a1, a2, a3});
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
            d += 16;
        }
        if (n & 8u) {
            put(d, (uint64_t)a0 | ((uint64_t)a1 << 32));
            a0 = a2; a1 = a3;
            d += 8;
        }
        if (n & 4u) {
            put(d, a0);
            a0 = a1;
            d += 4;
        }
        if (n & 2u) {
            put(d, (uint16_t)a0);
            a0 >>= 16;
            d += 2;
        }
        if (n & 1u) *d = (uint8_t)a0;
    }
    // gl = global index of element 0.  If the collector holds an element on a group boundary, the
    // group from that boundary up is complete (or is the ragged top of a chunk): store it, keep the
    // elements below it.
    __device__ __forceinline__ void flush_group(uint8_t *out, uint64_t gl) {
        const uint32_t extra = (0u - (uint32_t)gl) & (kGroup - 1) & 7u;   // elements below the boundary
        if (((0u - (uint32_t)gl) & (kGroup - 1)) >= 8u || extra >= cnt) return;   // no boundary within reach
        // the group from the boundary up: drop `extra` elements from the bottom
        const bool e4 = extra & 4u;
        uint32_t a0 = e4 ? r1 : r0;
        uint32_t a1 = e4 ? r2 : r1;
        uint32_t a2 = e4 ? r3 : r2;
        uint32_t a3 = e4 ? r4 : r3;
        uint32_t a4 = e4 ? r5 : r4;
        uint32_t a5 = e4 ? r6 : r5;
        uint32_t a6 = e4 ? r7 : r6;
        uint32_t a7 = e4 ? r8 : r7;
        uint32_t a8 = e4 ? r9 : r8;
        uint32_t a9 = e4 ? r10 : r9;
        uint32_t a10 = e4 ? r11 : r10;
        uint32_t a11 = e4 ? r12 : r11;
        uint32_t a12 = e4 ? r13 : r12;
        uint32_t a13 = e4 ? r14 : r13;
        uint32_t a14 = e4 ? r15 : r14;
        uint32_t a15 = e4 ? r16 : r15;
        uint32_t a16 = e4 ? r17 : r16;
        const uint32_t sel = 0x03020100u + 0x01010101u * (extra & 3u);
        a0 = __builtin_amdgcn_perm(a1, a0, sel);
        a1 = __builtin_amdgcn_perm(a2, a1, sel);
        a2 = __builtin_amdgcn_perm(a3, a2, sel);
        a3 = __builtin_amdgcn_perm(a4, a3, sel);
        a4 = __builtin_amdgcn_perm(a5, a4, sel);
        a5 = __builtin_amdgcn_perm(a6, a5, sel);
        a6 = __builtin_amdgcn_perm(a7, a6, sel);
        a7 = __builtin_amdgcn_perm(a8, a7, sel);
        a8 = __builtin_amdgcn_perm(a9, a8, sel);
        a9 = __builtin_amdgcn_perm(a10, a9, sel);
        a10 = __builtin_amdgcn_perm(a11, a10, sel);
        a11 = __builtin_amdgcn_perm(a12, a11, sel);
        a12 = __builtin_amdgcn_perm(a13, a12, sel);
        a13 = __builtin_amdgcn_perm(a14, a13, sel);
        a14 = __builtin_amdgcn_perm(a15, a14, sel);
        a15 = __builtin_amdgcn_perm(a16, a15, sel);
        const uint32_t n = cnt - extra;
        const uint64_t g = gl + extra;
        if (n == kGroup) {
            reinterpret_cast<uint4 *>(out + g)[0] = make_uint4(a0, a1, a2, a3);
            reinterpret_cast<uint4 *>(out + g)[1] = make_uint4(a4, a5, a6, a7);
            reinterpret_cast<uint4 *>(out + g)[2] = make_uint4(a8, a9, a10, a11);
            reinterpret_cast<uint4 *>(out + g)[3] = make_uint4(a12, a13, a14, a15);
        } else {
            pieces(out + g, n, a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15);
        }
        cnt = extra;
    }
    // The same flush for a whole wave at once (every lane calls it; `active` lanes take part): a
    // lane completes a group only every tenth trip or so, but some lane of the wave does in nearly
    // every trip.  The complete groups of the trip are parked in LDS (`scratch`, 4.5 KB of the wave's
    // own) and written by all 64 lanes together, 16 bytes each: one store instruction per 16 groups.
    // Ragged groups (the top of a chunk whose end is not aligned) go out in pieces.
    __device__ __forceinline__ void flush_group_wave(uint8_t *out, uint64_t gl, bool active, uint4 *scratch, uint32_t lane) {
        const uint32_t below = (0u - (uint32_t)gl) & (kGroup - 1);   // elements below the boundary
        const uint32_t extra = below & 7u;
        const bool has = active && below < 8u && extra < cnt;
        const uint32_t n = has ? cnt - extra : 0u;
        if (!__any(has)) return;
        // the group from the boundary up: drop `extra` elements from the bottom
        const bool e4 = extra & 4u;
        uint32_t a0 = e4 ? r1 : r0;
        uint32_t a1 = e4 ? r2 : r1;
        uint32_t a2 = e4 ? r3 : r2;
        uint32_t a3 = e4 ? r4 : r3;
        uint32_t a4 = e4 ? r5 : r4;
        uint32_t a5 = e4 ? r6 : r5;
        uint32_t a6 = e4 ? r7 : r6;
        uint32_t a7 = e4 ? r8 : r7;
        uint32_t a8 = e4 ? r9 : r8;
        uint32_t a9 = e4 ? r10 : r9;
        uint32_t a10 = e4 ? r11 : r10;
        uint32_t a11 = e4 ? r12 : r11;
        uint32_t a12 = e4 ? r13 : r12;
        uint32_t a13 = e4 ? r14 : r13;
        uint32_t a14 = e4 ? r15 : r14;
        uint32_t a15 = e4 ? r16 : r15;
        uint32_t a16 = e4 ? r17 : r16;
        const uint32_t sel = 0x03020100u + 0x01010101u * (extra & 3u);
        a0 = __builtin_amdgcn_perm(a1, a0, sel);
        a1 = __builtin_amdgcn_perm(a2, a1, sel);
        a2 = __builtin_amdgcn_perm(a3, a2, sel);
        a3 = __builtin_amdgcn_perm(a4, a3, sel);
        a4 = __builtin_amdgcn_perm(a5, a4, sel);
        a5 = __builtin_amdgcn_perm(a6, a5, sel);
        a6 = __builtin_amdgcn_perm(a7, a6, sel);
        a7 = __builtin_amdgcn_perm(a8, a7, sel);
        a8 = __builtin_amdgcn_perm(a9, a8, sel);
        a9 = __builtin_amdgcn_perm(a10, a9, sel);
        a10 = __builtin_amdgcn_perm(a11, a10, sel);
        a11 = __builtin_amdgcn_perm(a12, a11, sel);
        a12 = __builtin_amdgcn_perm(a13, a12, sel);
        a13 = __builtin_amdgcn_perm(a14, a13, sel);
        a14 = __builtin_amdgcn_perm(a15, a14, sel);
        a15 = __builtin_amdgcn_perm(a16, a15, sel);
        const uint64_t g = gl + extra;
        if (has && n != kGroup) pieces(out + g, n, a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15);   // the group at the END of a chunk
        if (has) cnt = extra;
        const bool full = n == kGroup;
        const unsigned long long mask = __ballot(full);
        if (mask == 0) return;
        uint64_t *const where = reinterpret_cast<uint64_t *>(scratch + 4 * 64);
        if (full) {
            const uint32_t rank = (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            scratch[4 * rank + 0] = make_uint4(a0, a1, a2, a3);
            scratch[4 * rank + 1] = make_uint4(a4, a5, a6, a7);
            scratch[4 * rank + 2] = make_uint4(a8, a9, a10, a11);
            scratch[4 * rank + 3] = make_uint4(a12, a13, a14, a15);
            where[rank] = g;
        }
        wave_sync();
        const uint32_t pieces16 = 4u * (uint32_t)__builtin_popcountll(mask);
        for (uint32_t t = lane; t < pieces16; t += 64)
            *reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(out + where[t / 4u]) + 16u * (t % 4u)) = scratch[t];
        wave_sync();
    }
    // what is left when a chunk is done (its first bases, below the last boundary)
    __device__ __forceinline__ void flush_rest(uint8_t *out, uint64_t gl) {
        if (cnt) pieces(out + gl, cnt, r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15);
        cnt = 0;
    }
};


}  // namespace colbwt
