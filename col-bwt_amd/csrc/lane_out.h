// lane_out.h -- the output path of the mismatch-line kernel (fat2_query.hip): what a lane reports is
// held in a compact form and leaves ONLY at the wave's flush trips, in as few store instructions as
// the bytes need.
//
// Why: the line-row kernels' launch time drops from 12.0 to 8.7 ms when the flush of lane_io.h's
// collectors is compiled out, and not at all when the pushes are (profiles/r03h_*): what costs is
// not vector work but STORE INSTRUCTIONS -- five per wave and trip (one per array for the trip's
// finished groups, three more, one lane each, for the ragged pieces of a lane that ends a chunk),
// each worth ~0.7 ms of launch time whatever it writes.  A wave produces 700 bytes per trip; one
// store instruction can carry 1024.
//
// So:
//   * PML values are not held at all.  They are runs l, l + 1, .. that restart at 0 where a base
//     did not match and at 1 where a read starts with a match (col_bwt.hpp:503-521): two 96-bit
//     masks ("element e restarts at 0 / at 1") and the value above the collector describe them;
//     the values are made when they are stored.  Col ids are bytes in 24 registers.  Element 0 is
//     the lowest address = the base reported last.
//   * Every fourth trip the wave flushes: a lane that holds a block boundary (64 elements: 128
//     bytes of PML, 64 of col ids) gives away everything from the boundary up -- a complete block, or
//     the ragged top of its chunk -- and a lane whose chunk is reported gives away the rest as well
//     (it waits for the flush trip before it enters its next chunk: 1.5 trips on average, once per
//     chunk).  The lanes dump their registers into LDS as they are; then all 64 lanes work through
//     the 16-byte pieces of all items -- a PML piece is expanded from the masks by the lane that
//     stores it, a col-id piece is re-aligned from the dumped bytes -- one store instruction per 64
//     pieces, and the partial pieces at the ragged ends go out byte by byte, 64 bytes per
//     instruction.  About six store instructions per flush, 1.5 per trip.
// Capacity: a lane keeps fewer than 64 elements after a flush and adds at most 8 per trip: 96.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lane_io.h"

namespace colbwt {

struct OutRuns {
    static constexpr uint32_t kBlock = 64;            // elements per block
    static constexpr uint32_t kPeriod = 4;            // trips between flushes
    static constexpr uint32_t kSlots = 32;            // lanes whose registers one pass of the flush parks
    // LDS of a flush pass (uint4 units): kSlots x 8 (the dumped registers), then 64 items x 2
    static constexpr uint32_t kItemBase = kSlots * 8, kLdsQ = kItemBase + 64 * 2;

    uint32_t z0a = 0, z0b = 0, z0c = 0, z1a = 0, z1b = 0, z1c = 0;   // bit e: element e restarts at 0 / at 1
    uint32_t ltop = 0;                                               // value of the element above the collector's highest
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0, c8 = 0, c9 = 0, c10 = 0, c11 = 0, c12 = 0, c13 = 0, c14 = 0, c15 = 0, c16 = 0, c17 = 0, c18 = 0, c19 = 0, c20 = 0, c21 = 0, c22 = 0, c23 = 0;
    uint32_t cnt = 0;

    // n <= 8 elements: PML values l_new - e for element e (keep / keep1 = 0: elements 0-1 / 2-3 are 0
    // instead: what a mismatch entry reports), col ids = byte e of ids_lo | ids_hi << 32
    __device__ __forceinline__ void push_run(uint32_t n, uint32_t l_new, uint32_t keep, uint32_t keep1, uint32_t ids_lo,
                                             uint32_t ids_hi) {
        const uint32_t in_run = (1u << n) - 1u;
        uint32_t p0 = l_new < n ? 1u << l_new : 0u;                  // the element whose value is 0
        p0 |= (keep ? 0u : 3u) & in_run;
        p0 |= (keep1 ? 0u : 0xCu) & in_run;
        const uint32_t top = (1u << n) >> 1;                         // the run's oldest element; none for an empty run
        const uint32_t p1 = (keep & keep1) != 0u && l_new == n ? top : 0u;   // ... has value 1 (marking a 1 after a 0 is harmless)
        z0c = (uint32_t)(((((uint64_t)z0c << 32) | z0b) << n) >> 32);
        z0b = (uint32_t)(((((uint64_t)z0b << 32) | z0a) << n) >> 32);
        z0a = (z0a << n) | p0;
        z1c = (uint32_t)(((((uint64_t)z1c << 32) | z1b) << n) >> 32);
        z1b = (uint32_t)(((((uint64_t)z1b << 32) | z1a) << n) >> 32);
        z1a = (z1a << n) | p1;
        // ---- col ids: up by n bytes = (n >> 2) dwords, then (n & 3) bytes
        const bool d4 = n & 8u, d2 = n & 4u;
        c23 = d4 ? c21 : c23;
        c22 = d4 ? c20 : c22;
        c21 = d4 ? c19 : c21;
        c20 = d4 ? c18 : c20;
        c19 = d4 ? c17 : c19;
        c18 = d4 ? c16 : c18;
        c17 = d4 ? c15 : c17;
        c16 = d4 ? c14 : c16;
        c15 = d4 ? c13 : c15;
        c14 = d4 ? c12 : c14;
        c13 = d4 ? c11 : c13;
        c12 = d4 ? c10 : c12;
        c11 = d4 ? c9 : c11;
        c10 = d4 ? c8 : c10;
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
        c23 = d2 ? c22 : c23;
        c22 = d2 ? c21 : c22;
        c21 = d2 ? c20 : c21;
        c20 = d2 ? c19 : c20;
        c19 = d2 ? c18 : c19;
        c18 = d2 ? c17 : c18;
        c17 = d2 ? c16 : c17;
        c16 = d2 ? c15 : c16;
        c15 = d2 ? c14 : c15;
        c14 = d2 ? c13 : c14;
        c13 = d2 ? c12 : c13;
        c12 = d2 ? c11 : c12;
        c11 = d2 ? c10 : c11;
        c10 = d2 ? c9 : c10;
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
        c23 = __builtin_amdgcn_perm(c23, c22, sel8);
        c22 = __builtin_amdgcn_perm(c22, c21, sel8);
        c21 = __builtin_amdgcn_perm(c21, c20, sel8);
        c20 = __builtin_amdgcn_perm(c20, c19, sel8);
        c19 = __builtin_amdgcn_perm(c19, c18, sel8);
        c18 = __builtin_amdgcn_perm(c18, c17, sel8);
        c17 = __builtin_amdgcn_perm(c17, c16, sel8);
        c16 = __builtin_amdgcn_perm(c16, c15, sel8);
        c15 = __builtin_amdgcn_perm(c15, c14, sel8);
        c14 = __builtin_amdgcn_perm(c14, c13, sel8);
        c13 = __builtin_amdgcn_perm(c13, c12, sel8);
        c12 = __builtin_amdgcn_perm(c12, c11, sel8);
        c11 = __builtin_amdgcn_perm(c11, c10, sel8);
        c10 = __builtin_amdgcn_perm(c10, c9, sel8);
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

    // ---- 96-bit helpers (lo: bits 0-63, hi: bits 64-95)
    static __device__ __forceinline__ void shr96(uint64_t &lo, uint32_t &hi, uint32_t s) {   // s < 96
        if (s >= 64) {
            lo = (uint64_t)(hi >> (s - 64));
            hi = 0;
        } else if (s) {
            lo = (lo >> s) | ((uint64_t)hi << (64 - s));         // the bits of hi that stay there are dropped on the way
            hi = s < 32 ? hi >> s : 0u;
        }
    }
    // value of element e given the masks shifted down to e (zz = z0 | z1, yy = z1), the value above the
    // collector and the elements from e to its top
    static __device__ __forceinline__ uint32_t value_at(uint64_t zz_lo, uint32_t zz_hi, uint64_t yy_lo, uint32_t yy_hi, uint32_t ltop,
                                                        uint32_t above) {
        if (zz_lo == 0 && zz_hi == 0) return ltop + above;           // the run comes from above the collector
        const uint32_t p = zz_lo ? (uint32_t)__builtin_ctzll(zz_lo) : 64u + (uint32_t)__builtin_ctz(zz_hi);
        const uint32_t one = p < 64 ? (uint32_t)(yy_lo >> p) & 1u : (yy_hi >> (p - 64)) & 1u;
        return p + one;
    }
    __device__ __forceinline__ uint32_t value(uint32_t e) const {   // e < cnt
        uint64_t zz = ((uint64_t)(z0b | z1b) << 32) | (z0a | z1a), yy = ((uint64_t)z1b << 32) | z1a;
        uint32_t zh = z0c | z1c, yh = z1c;
        shr96(zz, zh, e);
        shr96(yy, yh, e);
        return value_at(zz, zh, yy, yh, ltop, cnt - e);
    }

    // The flush of a whole wave (every lane calls it).  gl: global index of this lane's element 0;
    // active: the lane has a chunk; final: its chunk is reported, everything leaves.  lds: kLdsQ uint4
    // of the wave's own (the staged lines: they are read by now).
    __device__ __forceinline__ void flush_wave(uint16_t *pml, uint8_t *cid, uint64_t gl, bool active, bool final, uint4 *lds,
                                               uint32_t lane) {
        const uint32_t below = (0u - (uint32_t)gl) & (kBlock - 1);   // elements below the next block boundary
        const bool has_a = active && below < cnt;                    // from the boundary up: a block, or the chunk's ragged top
        const uint32_t rest = below < cnt ? below : cnt;
        const bool has_b = active && final && rest != 0;             // the elements below it, when the chunk is done
        const unsigned long long emit = __ballot(has_a || has_b);
        if (emit == 0) return;
        const uint32_t n_a = has_a ? cnt - below : 0u;
        const unsigned long long lt = (1ull << lane) - 1ull;
        const uint32_t rank = (uint32_t)__builtin_popcountll(emit & lt), lanes = (uint32_t)__builtin_popcountll(emit);
        for (uint32_t first = 0; first < lanes; first += kSlots) {
            const bool mine = (has_a || has_b) && rank >= first && rank < first + kSlots;
            // ---- items of this pass: ragged ones first (A items that are not whole blocks, all B items), then whole blocks
            const bool rag_a = mine && has_a && n_a != kBlock, rag_b = mine && has_b, full_a = mine && has_a && n_a == kBlock;
            const unsigned long long m_ra = __ballot(rag_a), m_rb = __ballot(rag_b), m_fa = __ballot(full_a);
            const uint32_t n_ra = (uint32_t)__builtin_popcountll(m_ra), n_rb = (uint32_t)__builtin_popcountll(m_rb);
            const uint32_t n_rag = n_ra + n_rb, n_items = n_rag + (uint32_t)__builtin_popcountll(m_fa);
            if (mine) {
                uint4 *slot = lds + 8 * (rank - first);
        slot[0] = make_uint4(c0, c1, c2, c3);
        slot[1] = make_uint4(c4, c5, c6, c7);
        slot[2] = make_uint4(c8, c9, c10, c11);
        slot[3] = make_uint4(c12, c13, c14, c15);
        slot[4] = make_uint4(c16, c17, c18, c19);
        slot[5] = make_uint4(c20, c21, c22, c23);
                slot[6] = make_uint4(z0a, z0b, z0c, z1a);
                slot[7] = make_uint4(z1b, z1c, ltop, cnt);
                if (has_a) {
                    const uint32_t at = rag_a ? (uint32_t)__builtin_popcountll(m_ra & lt) : n_rag + (uint32_t)__builtin_popcountll(m_fa & lt);
                    const uint64_t a = gl + below;
                    lds[kItemBase + 2 * at] = make_uint4(rank - first, below, n_a, (uint32_t)a);
                    lds[kItemBase + 2 * at + 1] = make_uint4((uint32_t)(a >> 32), 0u, 0u, 0u);
                }
                if (has_b) {
                    const uint32_t at = n_ra + (uint32_t)__builtin_popcountll(m_rb & lt);
                    lds[kItemBase + 2 * at] = make_uint4(rank - first, 0u, rest, (uint32_t)gl);
                    lds[kItemBase + 2 * at + 1] = make_uint4((uint32_t)(gl >> 32), 0u, 0u, 0u);
                }
            }
            wave_sync();
            // ---- whole 16-byte pieces: 8 of PML and 4 of col ids per item (PML tasks first, so that the
            // lanes of an iteration mostly do the same thing); ONE store instruction per iteration
            const uint32_t t_pml = 8u * n_items, t_all = 12u * n_items;
            for (uint32_t t = lane; t < t_all; t += 64) {
                const bool is_pml = t < t_pml;
                const uint32_t it = is_pml ? t >> 3 : (t - t_pml) >> 2, piece = is_pml ? t & 7u : (t - t_pml) & 3u;
                const uint4 h = lds[kItemBase + 2 * it];
                const uint64_t addr = (uint64_t)h.w | ((uint64_t)lds[kItemBase + 2 * it + 1].x << 32), end = addr + h.z;
                const uint32_t per = is_pml ? 8u : 16u;
                const uint64_t w0 = (addr & ~(uint64_t)(kBlock - 1)) + (uint64_t)piece * per;   // the piece's first element (global)
                if (w0 < addr || w0 + per > end) continue;                                   // not wholly inside the item
                const uint32_t e = h.y + (uint32_t)(w0 - addr);                             // its first element in the collector
                const uint4 *slot = lds + 8 * h.x;
                uint8_t *dst;
                uint4 val;
                if (is_pml) {
                    const uint4 ma = slot[6], mb = slot[7];
                    uint64_t zz = ((uint64_t)(ma.y | mb.x) << 32) | (ma.x | ma.w), yy = ((uint64_t)mb.x << 32) | ma.w;
                    uint32_t zh = ma.z | mb.y, yh = mb.y;
                    shr96(zz, zh, e);
                    shr96(yy, yh, e);
                    // element e + 7 first, then down: a restart says its value, else one more than the element above
                    uint64_t z7 = zz, y7 = yy;
                    uint32_t z7h = zh, y7h = yh;
                    shr96(z7, z7h, 7);
                    shr96(y7, y7h, 7);
                    uint32_t v = value_at(z7, z7h, y7, y7h, mb.z, mb.w - e - 7);
                    const uint32_t z1m = (uint32_t)yy, z0m = (uint32_t)zz & ~z1m;            // restarts at 1 / at 0 among the piece's 8
                    uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int i = 7; i >= 0; --i) {
                        if (i != 7) v = (z1m >> i) & 1u ? 1u : ((z0m >> i) & 1u ? 0u : v + 1u);
                        out[i >> 1] |= (v & 0xFFFFu) << (16 * (i & 1));
                    }
                    dst = reinterpret_cast<uint8_t *>(pml + w0);
                    val = make_uint4(out[0], out[1], out[2], out[3]);
                } else {
                    const uint32_t *bytes = reinterpret_cast<const uint32_t *>(slot);
                    const uint32_t d = e >> 2, sel = 0x03020100u + 0x01010101u * (e & 3u);
                    const uint32_t x0 = bytes[d], x1 = bytes[d + 1], x2 = bytes[d + 2], x3 = bytes[d + 3], x4 = bytes[d + 4];
                    dst = cid + w0;
                    val = make_uint4(__builtin_amdgcn_perm(x1, x0, sel), __builtin_amdgcn_perm(x2, x1, sel),
                                     __builtin_amdgcn_perm(x3, x2, sel), __builtin_amdgcn_perm(x4, x3, sel));
                }
                *reinterpret_cast<uint4 *>(dst) = val;
            }
            // ---- the partial pieces of the ragged items, byte by byte: an item covers at most two pieces
            // of an array partly, the one its first element is in and the one its end is in (16 bytes of
            // PML = 8 elements, 16 bytes of col ids each): 64 byte tasks per item, one store instruction
            // per 64 of them
            for (uint32_t t = lane; t < 64u * n_rag; t += 64) {
                const uint32_t it = t >> 6, q = t & 31u;
                const bool tail = (t >> 5) & 1u, is_pml = q < 16u;
                const uint4 h = lds[kItemBase + 2 * it];
                const uint64_t addr = (uint64_t)h.w | ((uint64_t)lds[kItemBase + 2 * it + 1].x << 32), end = addr + h.z;
                const uint64_t mask = is_pml ? 7u : 15u;
                const uint64_t w_head = addr & ~mask, w_tail = end & ~mask;
                if (tail ? ((end & mask) == 0 || (w_tail == w_head && (addr & mask) != 0)) : (addr & mask) == 0) continue;
                const uint64_t el = (tail ? w_tail : w_head) + (is_pml ? q >> 1 : q - 16u);   // this byte's element (global)
                if (el < addr || el >= end) continue;
                const uint32_t e = h.y + (uint32_t)(el - addr);
                const uint4 *slot = lds + 8 * h.x;
                uint8_t *dst;
                uint32_t byte;
                if (is_pml) {
                    const uint4 ma = slot[6], mb = slot[7];
                    uint64_t zz = ((uint64_t)(ma.y | mb.x) << 32) | (ma.x | ma.w), yy = ((uint64_t)mb.x << 32) | ma.w;
                    uint32_t zh = ma.z | mb.y, yh = mb.y;
                    shr96(zz, zh, e);
                    shr96(yy, yh, e);
                    const uint32_t v = value_at(zz, zh, yy, yh, mb.z, mb.w - e);
                    dst = reinterpret_cast<uint8_t *>(pml + el) + (q & 1u);
                    byte = v >> (8u * (q & 1u));
                } else {
                    dst = cid + el;
                    byte = reinterpret_cast<const uint8_t *>(slot)[e];
                }
                *dst = (uint8_t)byte;
            }
            wave_sync();
        }
        // ---- what the lane keeps
        if (has_a || has_b) {
            if (final) {
                z0a = z0b = z0c = z1a = z1b = z1c = 0;
                ltop = 0;
                cnt = 0;
            } else {
                ltop = value(below);                                 // the lowest element that left
                const uint64_t keep_lo = below >= 64 ? ~0ull : (1ull << below) - 1ull;   // below < 64
                z0a &= (uint32_t)keep_lo; z0b &= (uint32_t)(keep_lo >> 32); z0c = 0;
                z1a &= (uint32_t)keep_lo; z1b &= (uint32_t)(keep_lo >> 32); z1c = 0;
                cnt = below;
            }
        }
    }
};

}  // namespace colbwt
