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

// Read bytes of the line-row kernel: a ring of two 16-byte blocks per lane, indexed by global
// address (block A lives in slot (A / 16) & 1, piece `slot` of lane L at ring[slot][L]), topped
// up by ONE 16-byte load per trip: the lane requests the block below the lowest one it holds
// as soon as that does not overwrite a block it still reads, i.e. when the next byte to read has
// moved into the lower block.  A trip consumes at most 8 bytes and the block requested in a trip
// has landed before the trip looks at its bytes, so at least 17 bytes are at hand whenever 8 are
// looked at (LDS is what limits the resident waves: 8 KB per workgroup instead of 16 for a
// 64-byte window).  The request rides with the trip's row fetches and has landed
// when they have -- no memory round trip of its own, and one vector-memory instruction per trip
// (the texture addresser charges a wave instruction about the same whether 3 lanes or 64 take
// part, and it is what bounds this kernel).  Persistent lanes are out of step with each other, so
// a whole-window refill as in SlidingWindow would run in nearly every trip.  The block goes
// through registers, not LDS-DMA: its slot differs from lane to lane and a DMA's LDS address is
// one base per wave.
struct DmaRing {
    uint32_t lo;        // low 32 bits of the global index of the lowest byte held (a multiple of 16)
    uint4 in;           // the block on its way
    bool in_flight = false;
    __device__ __forceinline__ void init(uint64_t g) { lo = ((uint32_t)g + 16u) & ~15u; }   // nothing held; next block = g's
    // bytes held at and below global index g (0 when g's block is not there yet)
    __device__ __forceinline__ uint32_t avail(uint64_t g) const {
        const uint32_t d = (uint32_t)g - lo;
        return d < 32u ? d + 1u : 0u;
    }
    // Requests the block below the lowest one held, unless it would take the slot of a block at
    // or above g's (g = the highest byte still to be read): block lo - 16 shares its slot with
    // block lo + 16.
    __device__ __forceinline__ void request(const uint8_t *bases, uint64_t g) {
        const int32_t d = (int32_t)((uint32_t)g - lo);       // g - lo: < 0 while nothing at or below g is held
        const uint64_t lo_full = g - (int64_t)d;
        if (d < 16 && lo_full >= 16) {
            in = *reinterpret_cast<const uint4 *>(bases + (lo_full - 16));
            in_flight = true;
        }
    }
    // after the trip's wait: the requested block joins the ring
    __device__ __forceinline__ void land(uint4 (*ring)[64], uint32_t lane) {
        if (in_flight) {
            lo -= 16u;
            ring[(lo >> 4) & 1u][lane] = in;
            in_flight = false;
        }
    }
    __device__ __forceinline__ uint32_t dword(const uint4 (*ring)[64], uint32_t lane, uint32_t byte_addr) const {
        return reinterpret_cast<const uint32_t *>(&ring[(byte_addr >> 4) & 1u][lane])[(byte_addr >> 2) & 3u];
    }
    // the 8 read bytes ending at g as one little-endian word: byte 7 = base g, byte 0 = base g - 7
    // (bytes below `lo` are whatever the ring holds: callers cap what they use by `avail`)
    __device__ __forceinline__ uint64_t get8(const uint4 (*ring)[64], uint32_t lane, uint64_t g) const {
        const uint32_t b = (uint32_t)g;
        const uint32_t x2 = dword(ring, lane, b), x1 = dword(ring, lane, b - 4u), x0 = dword(ring, lane, b - 8u);
        const uint32_t sh = 8u * ((b & 3u) + 1u);                     // 8 .. 32
        const uint64_t l64 = (uint64_t)x0 | ((uint64_t)x1 << 32);
        return (l64 >> sh) | ((uint64_t)x2 << (64u - sh));
    }
};

// Outputs of the line-row kernel.  A trip reports a RUN of up to 8 bases: their PML values
// count up by one (col_bwt.hpp:517; a mismatch only ever opens the run, with 0), so the run is
// pushed in one go -- the collector (24 elements: 12 dwords of PML, 6 of col ids; element 0 =
// lowest address = newest) moves up by `cnt` elements through a fixed network of selects and
// byte permutes and the run is OR-ed in as a pattern -- instead of element by element, which a
// wave pays eight times per trip as soon as one lane has a run of 8.  The aligned 16-element
// group a push completes is stored by the one flush at the top of the next trip; the up to 7
// elements below the boundary stay.
struct OutAccRun {
    // scalar members on purpose: with arrays the compiler turns the select networks below into
    // dynamically indexed scratch (private memory) accesses
    uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0, q7 = 0, q8 = 0, q9 = 0, q10 = 0, q11 = 0;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    uint32_t cnt = 0;

    // n <= 8 elements: PML values l_new - e for element e, col ids = byte e of ids (ids_lo | ids_hi << 32)
    __device__ __forceinline__ void push_run(uint32_t n, uint32_t l_new, uint32_t ids_lo, uint32_t ids_hi) {
        // ---- PML: up by n halfwords = (n >> 1) dwords, then 16 bits
        const bool d4 = n & 8u, d2 = n & 4u, d1 = n & 2u;
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
        c5 = d4 ? c3 : c5;
        c4 = d4 ? c2 : c4;
        c3 = d4 ? c1 : c3;
        c2 = d4 ? c0 : c2;
        c1 = d4 ? 0u : c1;
        c0 = d4 ? 0u : c0;
        c5 = d2 ? c4 : c5;
        c4 = d2 ? c3 : c4;
        c3 = d2 ? c2 : c3;
        c2 = d2 ? c1 : c2;
        c1 = d2 ? c0 : c1;
        c0 = d2 ? 0u : c0;
        const uint32_t sel8 = 0x07060504u - 0x01010101u * (n & 3u);
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
    __device__ __forceinline__ void flush_group(uint16_t *pml, uint8_t *cid, uint64_t gl) {
        const uint32_t extra = (0u - (uint32_t)gl) & (kFlush - 1);   // elements below the boundary (<= 7)
        if (extra >= cnt) return;
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
        uint32_t a0 = e2 ? t1 : t0;
        uint32_t a1 = e2 ? t2 : t1;
        uint32_t a2 = e2 ? t3 : t2;
        uint32_t a3 = e2 ? t4 : t3;
        uint32_t a4 = e2 ? t5 : t4;
        uint32_t a5 = e2 ? t6 : t5;
        uint32_t a6 = e2 ? t7 : t6;
        uint32_t a7 = e2 ? t8 : t7;
        uint32_t a8 = e2 ? t9 : t8;
        const uint32_t sel16 = (extra & 1u) ? 0x05040302u : 0x03020100u;   // {hi, lo} >> 16 or lo
        a0 = __builtin_amdgcn_perm(a1, a0, sel16);
        a1 = __builtin_amdgcn_perm(a2, a1, sel16);
        a2 = __builtin_amdgcn_perm(a3, a2, sel16);
        a3 = __builtin_amdgcn_perm(a4, a3, sel16);
        a4 = __builtin_amdgcn_perm(a5, a4, sel16);
        a5 = __builtin_amdgcn_perm(a6, a5, sel16);
        a6 = __builtin_amdgcn_perm(a7, a6, sel16);
        a7 = __builtin_amdgcn_perm(a8, a7, sel16);
        uint32_t b0 = e4 ? c1 : c0;
        uint32_t b1 = e4 ? c2 : c1;
        uint32_t b2 = e4 ? c3 : c2;
        uint32_t b3 = e4 ? c4 : c3;
        uint32_t b4 = e4 ? c5 : c4;
        const uint32_t sel8 = 0x03020100u + 0x01010101u * (extra & 3u);
        b0 = __builtin_amdgcn_perm(b1, b0, sel8);
        b1 = __builtin_amdgcn_perm(b2, b1, sel8);
        b2 = __builtin_amdgcn_perm(b3, b2, sel8);
        b3 = __builtin_amdgcn_perm(b4, b3, sel8);
        const uint32_t n = cnt - extra;
        const uint64_t g = gl + extra;
        if (n == kFlush) {
            uint4 *dst = reinterpret_cast<uint4 *>(pml + g);
            dst[0] = make_uint4(a0, a1, a2, a3);
            dst[1] = make_uint4(a4, a5, a6, a7);
            *reinterpret_cast<uint4 *>(cid + g) = make_uint4(b0, b1, b2, b3);
        } else {   // the group at the END of the read
            OutAcc18::pieces(pml + g, cid + g, n, a0 | ((uint64_t)a1 << 32), a2 | ((uint64_t)a3 << 32),
                             a4 | ((uint64_t)a5 << 32), a6 | ((uint64_t)a7 << 32), b0 | ((uint64_t)b1 << 32),
                             b2 | ((uint64_t)b3 << 32));
        }
        cnt = extra;
    }
    // The same flush for a whole wave at once (every lane calls it; `active` lanes take part): a
    // lane completes a group only every fifth trip or so, but some lane of the wave does in nearly
    // every trip, and a vector-memory instruction costs the texture addresser about the same with
    // 13 lanes as with 64.  So the complete groups of the trip are parked in LDS (`scratch`, 3.5 KB
    // of the wave's own) and written by all 64 lanes together, 16 bytes each: one store
    // instruction per 21 groups instead of three per trip.  Ragged groups (the top of a chunk whose
    // end is not 16-aligned) go out in pieces as in flush_group.
    __device__ __forceinline__ void flush_group_wave(uint16_t *pml, uint8_t *cid, uint64_t gl, bool active, uint4 *scratch,
                                                     uint32_t lane) {
        const uint32_t extra = (0u - (uint32_t)gl) & (kFlush - 1);   // elements below the boundary (<= 7)
        const bool has = active && extra < cnt;
        const uint32_t n = has ? cnt - extra : 0u;
        if (!__any(has)) return;
        const bool e4 = extra & 4u, e2 = extra & 2u;
        // the group from the boundary up: drop `extra` elements (halfwords of q, bytes of c)
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
        uint32_t a0 = e2 ? t1 : t0;
        uint32_t a1 = e2 ? t2 : t1;
        uint32_t a2 = e2 ? t3 : t2;
        uint32_t a3 = e2 ? t4 : t3;
        uint32_t a4 = e2 ? t5 : t4;
        uint32_t a5 = e2 ? t6 : t5;
        uint32_t a6 = e2 ? t7 : t6;
        uint32_t a7 = e2 ? t8 : t7;
        uint32_t a8 = e2 ? t9 : t8;
        const uint32_t sel16 = (extra & 1u) ? 0x05040302u : 0x03020100u;   // {hi, lo} >> 16 or lo
        a0 = __builtin_amdgcn_perm(a1, a0, sel16);
        a1 = __builtin_amdgcn_perm(a2, a1, sel16);
        a2 = __builtin_amdgcn_perm(a3, a2, sel16);
        a3 = __builtin_amdgcn_perm(a4, a3, sel16);
        a4 = __builtin_amdgcn_perm(a5, a4, sel16);
        a5 = __builtin_amdgcn_perm(a6, a5, sel16);
        a6 = __builtin_amdgcn_perm(a7, a6, sel16);
        a7 = __builtin_amdgcn_perm(a8, a7, sel16);
        uint32_t b0 = e4 ? c1 : c0;
        uint32_t b1 = e4 ? c2 : c1;
        uint32_t b2 = e4 ? c3 : c2;
        uint32_t b3 = e4 ? c4 : c3;
        uint32_t b4 = e4 ? c5 : c4;
        const uint32_t sel8 = 0x03020100u + 0x01010101u * (extra & 3u);
        b0 = __builtin_amdgcn_perm(b1, b0, sel8);
        b1 = __builtin_amdgcn_perm(b2, b1, sel8);
        b2 = __builtin_amdgcn_perm(b3, b2, sel8);
        b3 = __builtin_amdgcn_perm(b4, b3, sel8);
        const uint64_t g = gl + extra;
        if (has && n != kFlush)      // the group at the END of a chunk
            OutAcc18::pieces(pml + g, cid + g, n, a0 | ((uint64_t)a1 << 32), a2 | ((uint64_t)a3 << 32),
                             a4 | ((uint64_t)a5 << 32), a6 | ((uint64_t)a7 << 32), b0 | ((uint64_t)b1 << 32),
                             b2 | ((uint64_t)b3 << 32));
        if (has) cnt = extra;
        const bool full = n == kFlush;
        const unsigned long long mask = __ballot(full);
        if (mask == 0) return;
        uint64_t *const where = reinterpret_cast<uint64_t *>(scratch + 3 * 64);
        if (full) {
            const uint32_t rank = (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            scratch[3 * rank] = make_uint4(a0, a1, a2, a3);
            scratch[3 * rank + 1] = make_uint4(a4, a5, a6, a7);
            scratch[3 * rank + 2] = make_uint4(b0, b1, b2, b3);
            where[rank] = g;
        }
        wave_sync();
        const uint32_t pieces16 = 3u * (uint32_t)__builtin_popcountll(mask);
        for (uint32_t t = lane; t < pieces16; t += 64) {
            const uint32_t item = t / 3u, part = t - 3u * item;
            const uint64_t gi = where[item];
            uint8_t *dst = part == 2 ? cid + gi : reinterpret_cast<uint8_t *>(pml + gi) + 16u * part;
#ifndef COLBWT_EXPERIMENT_NO_STORES
            *reinterpret_cast<uint4 *>(dst) = scratch[t];
#else
            if (gi == 0x7FFFFFFFFFFFull) *reinterpret_cast<uint4 *>(dst) = scratch[t];
#endif
        }
        wave_sync();
    }
    // what is left when the read is done (its first bases, below the last boundary)
    __device__ __forceinline__ void flush_rest(uint16_t *pml, uint8_t *cid, uint64_t gl) {
        if (cnt)
            OutAcc18::pieces(pml + gl, cid + gl, cnt, q0 | ((uint64_t)q1 << 32), q2 | ((uint64_t)q3 << 32),
                             q4 | ((uint64_t)q5 << 32), q6 | ((uint64_t)q7 << 32), c0 | ((uint64_t)c1 << 32),
                             c2 | ((uint64_t)c3 << 32));
        cnt = 0;
    }
};

}  // namespace colbwt
