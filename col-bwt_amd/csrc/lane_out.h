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

// timing experiments only (tools/ab_parts.sh): -DCOLBWT_ABL_NO_STORES keeps the flush and drops its stores
#ifdef COLBWT_ABL_NO_STORES
#define COLBWT_ABL_IF if (ltop == 0xFFFFFFF1u)
#else
#define COLBWT_ABL_IF
#endif

namespace colbwt {

struct OutRuns {
    static constexpr uint32_t kBlock = 64;            // elements per block
    static constexpr uint32_t kPeriod = 4;            // trips between flushes
    // LDS of a flush pass (uint4 units): kSlots x 6 (the dumped col-id registers of the lanes one pass
    // parks), then 2 * kSlots items x 4 (a lane has at most two)
    static constexpr uint32_t kSlotQ = 6, kItemQ = 4;
    static constexpr uint32_t lds_q(uint32_t slots) { return slots * kSlotQ + 2 * slots * kItemQ; }

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
    // The low 64 bits of a 96-bit mask shifted down by s < 96
    static __device__ __forceinline__ uint64_t low64_from(uint32_t a, uint32_t b, uint32_t c, uint32_t s) {
        uint64_t lo = ((uint64_t)b << 32) | a;
        uint32_t hi = c;
        shr96(lo, hi, s);
        return lo;
    }
    // Parks one item -- `count` <= 64 elements from collector element e_lo on, going to global element
    // address `addr` (all inside one 64-element block), `lt` = the value of the element above its top
    // one -- for the lanes that will store it: a header and, per 8-element piece p of the block, one
    // dword { value above the piece's highest element of the item, its restart-at-0 bits, its
    // restart-at-1 bits }: whoever stores a piece makes its values with 8-bit arithmetic alone.
    // Returns the value of the item's lowest element.
    __device__ __forceinline__ uint32_t park_item(uint4 *item, uint32_t slot, uint32_t e_lo, uint32_t count, uint64_t addr, uint32_t lt) const {
        const uint32_t o_lo = (uint32_t)addr & (kBlock - 1), o_hi = o_lo + count;
        const uint64_t in_item = count >= 64 ? ~0ull : (1ull << count) - 1ull;
        const uint64_t Z0 = (low64_from(z0a, z0b, z0c, e_lo) & in_item) << o_lo;      // block-aligned restart masks of the item
        const uint64_t Z1 = (low64_from(z1a, z1b, z1c, e_lo) & in_item) << o_lo;
        const uint32_t z0w[2] = {(uint32_t)Z0, (uint32_t)(Z0 >> 32)}, z1w[2] = {(uint32_t)Z1, (uint32_t)(Z1 >> 32)};
        uint32_t pm[8];
        uint32_t v = lt;
#pragma unroll
        for (int p = 7; p >= 0; --p) {
            const uint32_t b0 = (z0w[p >> 2] >> (8 * (p & 3))) & 0xFFu, b1 = (z1w[p >> 2] >> (8 * (p & 3))) & 0xFFu;
            pm[p] = (v & 0xFFFFu) | (b0 << 16) | (b1 << 24);
            const uint32_t lo_p = o_lo > 8u * p ? o_lo : 8u * p, hi_p = o_hi < 8u * p + 8u ? o_hi : 8u * p + 8u;
            if (hi_p > lo_p) {                                       // the value of the piece's lowest element of the item
                const uint32_t bits = b0 | b1;
                if (bits) {
                    const uint32_t q = (uint32_t)__builtin_ctz(bits);
                    v = q - (lo_p - 8u * p) + ((b1 >> q) & 1u);
                } else {
                    v += hi_p - lo_p;
                }
            }
        }
        item[0] = make_uint4(slot, e_lo, count, (uint32_t)addr);
        item[1] = make_uint4((uint32_t)(addr >> 32), 0u, 0u, 0u);
        item[2] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        item[3] = make_uint4(pm[4], pm[5], pm[6], pm[7]);
        return v;
    }

    // The flush of a whole wave (every lane calls it).  gl: global index of this lane's element 0;
    // active: the lane has a chunk; final: its chunk is reported, everything leaves.  lds: lds_q(kSlots) uint4
    // of the wave's own (the staged lines: they are read by now).
    template <uint32_t kSlots = 32>
    __device__ __forceinline__ void flush_wave(uint16_t *pml, uint8_t *cid, uint64_t gl, bool active, bool final, uint4 *lds,
                                               uint32_t lane) {
        constexpr uint32_t kItemBase = kSlots * kSlotQ;
        const uint32_t below = (0u - (uint32_t)gl) & (kBlock - 1);   // elements below the next block boundary
        const bool has_a = active && below < cnt;                    // from the boundary up: a block, or the chunk's ragged top
        const uint32_t rest = below < cnt ? below : cnt;
        const bool has_b = active && final && rest != 0;             // the elements below it, when the chunk is done
        const unsigned long long emit = __ballot(has_a || has_b);
        if (emit == 0) return;
        const uint32_t n_a = has_a ? cnt - below : 0u;
        uint32_t v_cut = ltop;                                       // the value just above what stays / above item B: the walk over item A gives it
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
                uint4 *slot = lds + kSlotQ * (rank - first);
                slot[0] = make_uint4(c0, c1, c2, c3);
                slot[1] = make_uint4(c4, c5, c6, c7);
                slot[2] = make_uint4(c8, c9, c10, c11);
                slot[3] = make_uint4(c12, c13, c14, c15);
                slot[4] = make_uint4(c16, c17, c18, c19);
                slot[5] = make_uint4(c20, c21, c22, c23);
            }
            // (one copy of park_item's code for both kinds of item: inlined twice it costs 50 registers)
#pragma unroll 1
            for (uint32_t which = 0; which < 2; ++which) {
                const bool b_item = which != 0;
                if (!__any(mine && (b_item ? has_b : has_a))) continue;
                if (mine && (b_item ? has_b : has_a)) {
                    const uint32_t at = b_item ? n_ra + (uint32_t)__builtin_popcountll(m_rb & lt)
                                               : (rag_a ? (uint32_t)__builtin_popcountll(m_ra & lt) : n_rag + (uint32_t)__builtin_popcountll(m_fa & lt));
                    const uint32_t v_low = park_item(lds + kItemBase + kItemQ * at, rank - first, b_item ? 0u : below, b_item ? rest : n_a,
                                                     b_item ? gl : gl + below, b_item ? v_cut : ltop);
                    if (!b_item) v_cut = v_low;
                }
            }
            wave_sync();
            // ---- whole 16-byte pieces of PML: 8 per item, one store instruction per 64 of them
            for (uint32_t t = lane; t < 8u * n_items; t += 64) {
                const uint32_t it = t >> 3, piece = t & 7u;
                const uint4 *item = lds + kItemBase + kItemQ * it;
                const uint4 h = item[0];
                const uint64_t addr = (uint64_t)h.w | ((uint64_t)item[1].x << 32), end = addr + h.z;
                const uint64_t w0 = (addr & ~(uint64_t)(kBlock - 1)) + 8u * piece;            // the piece's first element (global)
                if (w0 < addr || w0 + 8u > end) continue;                                    // not wholly inside the item
                const uint32_t d = reinterpret_cast<const uint32_t *>(item + 2)[piece];
                const uint32_t b0 = (d >> 16) & 0xFFu, b1 = d >> 24;
                uint32_t v = d & 0xFFFFu;                                                    // the value above the piece
                uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
                for (int i = 7; i >= 0; --i) {                       // a restart says its value, else one more than the element above
                    v = (b1 >> i) & 1u ? 1u : ((b0 >> i) & 1u ? 0u : v + 1u);
                    out[i >> 1] |= (v & 0xFFFFu) << (16 * (i & 1));
                }
                COLBWT_ABL_IF *reinterpret_cast<uint4 *>(pml + w0) = make_uint4(out[0], out[1], out[2], out[3]);
            }
            // ---- whole 16-byte pieces of col ids: 4 per item, re-aligned from the dumped registers
            for (uint32_t t = lane; t < 4u * n_items; t += 64) {
                const uint32_t it = t >> 2, piece = t & 3u;
                const uint4 *item = lds + kItemBase + kItemQ * it;
                const uint4 h = item[0];
                const uint64_t addr = (uint64_t)h.w | ((uint64_t)item[1].x << 32), end = addr + h.z;
                const uint64_t w0 = (addr & ~(uint64_t)(kBlock - 1)) + 16u * piece;
                if (w0 < addr || w0 + 16u > end) continue;
                const uint32_t e = h.y + (uint32_t)(w0 - addr);                              // its first element in the collector
                const uint32_t *bytes = reinterpret_cast<const uint32_t *>(lds + kSlotQ * h.x);
                const uint32_t dq = e >> 2, sel = 0x03020100u + 0x01010101u * (e & 3u);
                const uint32_t x0 = bytes[dq], x1 = bytes[dq + 1], x2 = bytes[dq + 2], x3 = bytes[dq + 3], x4 = bytes[dq + 4 < 24u ? dq + 4 : 23u];   // beyond the registers only when it is not used
                COLBWT_ABL_IF *reinterpret_cast<uint4 *>(cid + w0) =
                    make_uint4(__builtin_amdgcn_perm(x1, x0, sel), __builtin_amdgcn_perm(x2, x1, sel), __builtin_amdgcn_perm(x3, x2, sel),
                               __builtin_amdgcn_perm(x4, x3, sel));
            }
            // ---- the partial pieces of the ragged items: an item covers at most two pieces of an array
            // partly, the one its first element is in and the one its end is in.  PML: 2 x 8 elements,
            // a halfword store each; col ids: 2 x 16 elements, a byte store each -- one store
            // instruction per 64 elements.
            for (uint32_t t = lane; t < 16u * n_rag; t += 64) {
                const uint32_t it = t >> 4, i = t & 7u;
                const bool tail = (t >> 3) & 1u;
                const uint4 *item = lds + kItemBase + kItemQ * it;
                const uint4 h = item[0];
                const uint64_t addr = (uint64_t)h.w | ((uint64_t)item[1].x << 32), end = addr + h.z;
                const uint64_t w_head = addr & ~7ull, w_tail = end & ~7ull;
                if (tail ? ((end & 7u) == 0 || (w_tail == w_head && (addr & 7u) != 0)) : (addr & 7u) == 0) continue;
                const uint64_t w0 = tail ? w_tail : w_head, el = w0 + i;
                if (el < addr || el >= end) continue;
                const uint32_t piece = (uint32_t)(w0 >> 3) & 7u;
                const uint32_t d = reinterpret_cast<const uint32_t *>(item + 2)[piece];
                const uint32_t b0 = (d >> 16) & 0xFFu, b1 = d >> 24, bits = (b0 | b1) >> i;   // restarts at or above this element (item's only)
                const uint32_t hi = end - w0 < 8u ? (uint32_t)(end - w0) : 8u;                // the piece's elements of the item end here
                uint32_t v;
                if (bits) {
                    const uint32_t q = (uint32_t)__builtin_ctz(bits);
                    v = q + ((b1 >> (i + q)) & 1u);
                } else {
                    v = (d & 0xFFFFu) + hi - i;
                }
                COLBWT_ABL_IF pml[el] = (uint16_t)v;
            }
            for (uint32_t t = lane; t < 32u * n_rag; t += 64) {
                const uint32_t it = t >> 5, i = t & 15u;
                const bool tail = (t >> 4) & 1u;
                const uint4 *item = lds + kItemBase + kItemQ * it;
                const uint4 h = item[0];
                const uint64_t addr = (uint64_t)h.w | ((uint64_t)item[1].x << 32), end = addr + h.z;
                const uint64_t w_head = addr & ~15ull, w_tail = end & ~15ull;
                if (tail ? ((end & 15u) == 0 || (w_tail == w_head && (addr & 15u) != 0)) : (addr & 15u) == 0) continue;
                const uint64_t el = (tail ? w_tail : w_head) + i;
                if (el < addr || el >= end) continue;
                COLBWT_ABL_IF cid[el] = reinterpret_cast<const uint8_t *>(lds + kSlotQ * h.x)[h.y + (uint32_t)(el - addr)];
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
                ltop = v_cut;                                        // the lowest element that left
                const uint64_t keep_lo = (1ull << below) - 1ull;     // below < 64
                z0a &= (uint32_t)keep_lo; z0b &= (uint32_t)(keep_lo >> 32); z0c = 0;
                z1a &= (uint32_t)keep_lo; z1b &= (uint32_t)(keep_lo >> 32); z1c = 0;
                cnt = below;
            }
        }
    }
};

}  // namespace colbwt
