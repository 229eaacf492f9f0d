// fat_build.hip -- builds the line-row layout (fat_layout.h) on the device, once per index.
//
//   levels  level 2 is refined from the one-step tables, level L+1 from level L (refine.h:
//           count / scan, then emit + link below), kept in a plain SoA form: idx, the landing
//           (row, offset) of LF(first position), char | col id, threshold;
//   chars   byte-per-row character array of level K and its per-block jump tables;
//   pack    one thread per level-K row chases LF through the plain level K: K steps from the
//           row's first position (landings, characters, col ids, cuts) and two steps from each
//           of the three mismatch targets.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "../../include/colbwt.h"
#include "dev_mem.h"
#include "device_layout.h"
#include "fat_layout.h"
#include "jump_tables.h"
#include "lf_device.h"
#include "query_kernels.h"
#include "read_sampler.h"
#include "refine.h"

namespace colbwt {

namespace {

struct PlainLevel {
    const uint64_t *idx;    // r + 1 (idx[r] = n)
    const uint32_t *I;      // r : row of this level holding LF(first position of the row)
    const uint16_t *O;      // r : offset of that image inside I (< 65535: exact, no fast-forward left)
    const uint16_t *meta;   // r : char | col id << 8
    const uint64_t *thr;    // r : threshold of the BWT run
    const uint8_t *org;     // r : 1 where an origin row starts (fat_layout.h, mismatch lines)
    uint64_t n;
    uint32_t r;
};

struct PlainBuffers {
    DevPtr idx, I, O, meta, thr, org;
    void release() {
        for (DevPtr *p : {&idx, &I, &O, &meta, &thr, &org}) p->reset();
    }
};

struct SrcPlain {   // a plain level as the source of the next one (the interface of refine.h)
    PlainLevel T;
    __device__ __forceinline__ uint32_t cuts(uint32_t, uint64_t (&)[kHintSlots]) const { return 0; }   // cut in the first pass
    __device__ __forceinline__ bool origin_start(uint32_t i, uint64_t b) const { return b == T.idx[i] && T.org[i]; }
    __device__ __forceinline__ uint32_t rows() const { return T.r; }
    __device__ __forceinline__ uint64_t n() const { return T.n; }
    __device__ __forceinline__ uint64_t idx(uint32_t j) const { return T.idx[j]; }
    __device__ __forceinline__ uint64_t len(uint32_t j) const { return T.idx[(uint64_t)j + 1] - T.idx[j]; }
    __device__ __forceinline__ uint64_t thr(uint32_t j) const { return T.thr[j]; }
    __device__ __forceinline__ void lf(uint32_t j, int, uint32_t &dj, uint64_t &dt) const {
        dj = T.I[j];
        dt = T.O[j];
    }
    __device__ __forceinline__ uint32_t ch_at(uint32_t j, int) const { return T.meta[j] & 0xFFu; }
    __device__ __forceinline__ uint32_t cid_at(uint32_t j, int) const { return T.meta[j] >> 8; }
};

// New rows of one refinement pass; the image of a row's first position is parked as
// (source row in I, BWT position in `park`) until the link pass knows the new row numbers.
template <class Src>
__global__ __launch_bounds__(256) void plain_emit_kernel(Src S, const uint32_t *__restrict__ first,
                                                         uint64_t *__restrict__ idx_new, uint32_t *__restrict__ I_new,
                                                         uint16_t *__restrict__ meta_new, uint64_t *__restrict__ thr_new,
                                                         uint8_t *__restrict__ org_new, uint64_t *__restrict__ park, uint32_t r_new) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= S.rows()) return;
    const uint32_t meta = S.ch_at((uint32_t)i, 1) | (S.cid_at((uint32_t)i, 1) << 8);
    const uint64_t thr = S.thr((uint32_t)i);
    uint32_t out = first[i];
    for_each_piece(S, (uint32_t)i, [&](uint64_t b, uint32_t, uint32_t j, uint64_t t) {
        idx_new[out] = b;
        I_new[out] = j;
        park[out] = S.idx(j) + t;
        meta_new[out] = (uint16_t)meta;
        thr_new[out] = thr;
        org_new[out] = S.origin_start((uint32_t)i, b) ? 1 : 0;
        ++out;
    });
    if (i + 1 == S.rows()) idx_new[r_new] = S.n();   // sentinel
}

__global__ __launch_bounds__(256) void plain_link_kernel(const uint32_t *__restrict__ first, const uint64_t *__restrict__ idx_new,
                                                         const uint64_t *__restrict__ park, uint32_t *__restrict__ I_new,
                                                         uint16_t *__restrict__ O_new, uint32_t r_new) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= r_new) return;
    const uint64_t pos = park[i];
    const uint32_t dst = sk_find(idx_new, first, I_new[i], pos);
    I_new[i] = dst;
    O_new[i] = (uint16_t)(pos - idx_new[dst]);
}

__global__ __launch_bounds__(256) void plain_chars_kernel(const uint16_t *__restrict__ meta, uint32_t r, uint8_t *__restrict__ chr) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < r) chr[i] = (uint8_t)(meta[i] & 0xFFu);
}

// One wave per 256-row block: first / last row of the block holding each present character.
__global__ __launch_bounds__(256) void chr_block_first_last_kernel(const uint8_t *__restrict__ chr, uint32_t r, uint32_t nblk,
                                                                   uint32_t sigma, const uint8_t *__restrict__ cmap,
                                                                   uint32_t *__restrict__ first, uint32_t *__restrict__ last) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nblk) return;
    const uint64_t base = (uint64_t)b << kBlockShift;
    uint32_t cx[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint64_t row = base + (uint64_t)s * 64 + lane;
        cx[s] = kNone;
        if (row < r) cx[s] = cmap[chr[row]];
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

// LF_table::LF (LF_table.hpp:251-262) inside a plain level: from offset t of row j.
__device__ __forceinline__ void plain_lf(const PlainLevel &P, uint32_t &j, uint64_t &t) {
    uint32_t nj = P.I[j];
    uint64_t nt = (uint64_t)P.O[j] + t;
    uint64_t len = P.idx[(uint64_t)nj + 1] - P.idx[nj];
    while (nt >= len && nj < P.r - 1) {
        nt -= len;
        ++nj;
        len = P.idx[(uint64_t)nj + 1] - P.idx[nj];
    }
    j = nj;
    t = nt;
}

__device__ __forceinline__ void put_byte(uint32_t *w, uint32_t off, uint32_t v) { w[off >> 2] |= (v & 0xFFu) << (8 * (off & 3u)); }
__device__ __forceinline__ void put_half(uint32_t *w, uint32_t off, uint32_t v) { w[off >> 2] |= (v & 0xFFFFu) << (8 * (off & 2u)); }

// col_pml::threshold_step (col_bwt.hpp:531-574) resolved for a whole row: where a mismatch on the
// character with dense index cidx takes EVERY position of level-K row i -- the head of the
// succeeding run of that character or the tail of the preceding one, decided by the row's place
// relative to the threshold.  false: there is no such run, or the threshold lies inside the row
// (the query decides at run time).
__device__ __forceinline__ bool fat_slot_target(const PlainLevel &P, const FatTable &T, uint32_t i, uint32_t cidx, uint32_t &tj,
                                                uint64_t &to) {
    const uint64_t lo = P.idx[i], len = P.idx[(uint64_t)i + 1] - lo;
    const uint32_t c = (T.top4 >> (8 * cidx)) & 0xFFu;
    const uint32_t s = fat_succ_char(T, i, c, cidx);                 // :548
    const uint64_t thr = s != kNone ? P.thr[s] : P.n;                // :535 / :553
    tj = kNone;
    to = 0;
    if (lo + len - 1 < thr) {                                        // :560 true for the whole row
        const uint32_t q = fat_pred_char(T, i, c, cidx);             // :562
        if (q != kNone) { tj = q; to = P.idx[(uint64_t)q + 1] - P.idx[q] - 1; }   // :565-569
        else if (s != kNone) { tj = s; to = 0; }                     // :552-557
    } else if (lo >= thr) {                                          // false for the whole row
        tj = s;                                                      // exists: thr < n
        to = 0;
    }                                                                // else: the threshold is inside the row
    return tj != kNone;
}

// ---- origin rows (fat_layout.h, mismatch lines): rho[i] = origin row of level-K row i, rho_first
// its first level-K row.  `rho` arrives holding the exclusive prefix sum of the origin-start flags.
__global__ __launch_bounds__(256) void org_flags_kernel(const uint8_t *__restrict__ org, uint32_t r, uint32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < r) out[i] = org[i] ? 1u : 0u;
}

__global__ __launch_bounds__(256) void rho_kernel(const uint8_t *__restrict__ org, uint32_t r, uint32_t *__restrict__ rho,
                                                  uint32_t *__restrict__ rho_first) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= r) return;
    const uint32_t flag = org[i] ? 1u : 0u;
    const uint32_t id = rho[i] + flag - 1u;      // row 0 starts an origin row
    rho[i] = id;
    if (flag) rho_first[id] = (uint32_t)i;
}

// Per level-K row: index of its character among the four most frequent << 4 (7 = other) | bit s =
// mismatch slot s is decided for the row AND takes it where it takes the first row of its origin
// row -- the entry of (origin row, slot) is packed from that first row, and every row that points
// a lane to it must mean the same position.  (Rows are cut at the thresholds of the slotted
// characters, so the two agree whenever both are decided; the comparison is the guarantee.)
__global__ __launch_bounds__(256) void fat_slot_flags_kernel(PlainLevel P, FatTable T, const uint32_t *__restrict__ rho,
                                                             const uint32_t *__restrict__ rho_first, uint8_t *__restrict__ flags) {
    const uint64_t i64 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i64 >= P.r) return;
    const uint32_t i = (uint32_t)i64;
    const uint32_t a_dense = T.cmap[P.meta[i] & 0xFFu];
    uint32_t f = (a_dense < 4 ? a_dense : kFatOwnOther) << 4;
    const uint32_t head = rho_first[rho[i]];
    const uint32_t top = T.sigma < 4 ? T.sigma : 4;
    for (uint32_t cidx = 0; cidx < top; ++cidx) {
        if (cidx == a_dense) continue;
        const uint32_t slot = hint_slot(cidx, a_dense);
        if (slot >= kFatSlots) continue;
        uint32_t tj, hj;
        uint64_t to, ho;
        if (!fat_slot_target(P, T, i, cidx, tj, to)) continue;
        if (head != i && (!fat_slot_target(P, T, head, cidx, hj, ho) || hj != tj || ho != to)) continue;
        f |= 1u << slot;
    }
    flags[i] = (uint8_t)f;
}

// kMis = false: rows with in-row mismatch slots; true: rows of the mismatch-line variant (rho,
// sflags: the arrays of the two kernels above).
template <int K, bool kMis>
__global__ __launch_bounds__(256) void fat_pack_kernel(PlainLevel P, FatTable T, const uint32_t *__restrict__ rho,
                                                       const uint8_t *__restrict__ sflags, uint8_t *__restrict__ lines) {
    const uint64_t i64 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i64 >= P.r) return;
    const uint32_t i = (uint32_t)i64;
    uint32_t w[kFatRowBytes / 4];
#pragma unroll
    for (uint32_t q = 0; q < kFatRowBytes / 4; ++q) w[q] = 0;
    const uint64_t len = P.idx[(uint64_t)i + 1] - P.idx[i];
    put_half(w, kFatLen, (uint32_t)len);

    // ---- the row's own K steps
    uint32_t j = i;
    uint64_t t = 0;
    uint32_t own_ch = 0;
#pragma unroll
    for (int s = 1; s <= K; ++s) {
        const uint32_t meta = P.meta[j];
        if (s == 1) own_ch = meta & 0xFFu;
        put_byte(w, kFatCh + (8 - s), meta & 0xFFu);
        put_byte(w, kFatCid + (8 - s), meta >> 8);
        if constexpr (kMis) {
            // The row met after s - 1 steps: its origin row, which of its entries exist.  A level-K
            // row stays inside ONE row of level K - s + 1 for s - 1 steps, and from level 2 up those
            // are cut at the thresholds -- but after K - 1 steps all that is left is "one row of the
            // file", whose threshold cuts the image may straddle: then the positions of this row
            // are in different origin rows at that depth and no single entry is theirs.
            bool one_origin = true;
            if (s == K) {
                const uint64_t end = P.idx[j] + t + len;     // image = [idx[j] + t, end)
                for (uint64_t q = (uint64_t)j + 1; q < P.r && P.idx[q] < end; ++q) one_origin = one_origin && !P.org[q];
            }
            w[kFatRho / 4 + (s - 1)] = rho[j];
            if (one_origin) w[kFatVal / 4] |= (uint32_t)(sflags[j] & 7u) << (3 * (s - 1));   // 24 bits of one dword
        }
        plain_lf(P, j, t);                                   // (j, t) = LF^s(first position)
        w[kFatI / 4 + (s - 1)] = j;
        put_half(w, kFatO + 2 * (s - 1), (uint32_t)t);
        // where the image of the row leaves row j and row j + 1 (sk_layout.h)
        uint32_t cut = kSKCutNone, len_b = kSKCutNone;
        if ((uint64_t)j + 1 < P.r) {
            const uint64_t c = P.idx[(uint64_t)j + 1] - P.idx[j] - t;
            if (c < kSKCutNone) {
                cut = (uint32_t)c;
                if ((uint64_t)j + 2 < P.r) {
                    const uint64_t l = P.idx[(uint64_t)j + 2] - P.idx[(uint64_t)j + 1];
                    if (l < kSKCutNone) len_b = (uint32_t)l;
                }
            }
        }
        put_byte(w, kFatCut + (s - 1), cut | (len_b << 4));
    }

    if constexpr (kMis) {
        put_byte(w, kFatFlags, sflags[i]);
    } else {
        // ---- the mismatch slots: col_pml::threshold_step (col_bwt.hpp:531-574) resolved per row
        const uint32_t a_dense = T.cmap[own_ch];
        const uint32_t aidx = a_dense < 4 ? a_dense : kFatOwnOther;
        uint32_t flags = aidx << 4;
        const uint32_t top = T.sigma < 4 ? T.sigma : 4;
        for (uint32_t cidx = 0; cidx < top; ++cidx) {
            if (cidx == a_dense) continue;
            const uint32_t slot = hint_slot(cidx, a_dense);
            if (slot >= kFatSlots) continue;
            uint32_t tj;
            uint64_t to;
            if (!fat_slot_target(P, T, i, cidx, tj, to)) continue;       // the query decides at run time
            flags |= 1u << slot;
            const uint32_t sb = kFatSlot0 + slot * kFatSlotBytes;
            plain_lf(P, tj, to);                                         // LF(p_c)
            w[sb >> 2] = tj;
            put_half(w, sb + kFatSlotP, (uint32_t)to);
            const uint32_t meta = P.meta[tj];                            // met after one step
            put_byte(w, sb + kFatSlotCh2, meta & 0xFFu);
            put_byte(w, sb + kFatSlotCid2, meta >> 8);
            plain_lf(P, tj, to);                                         // LF^2(p_c)
            w[(sb >> 2) + 1] = tj;
            put_half(w, sb + kFatSlotP + 2, (uint32_t)to);
        }
        put_byte(w, kFatFlags, flags);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(lines + (uint64_t)i * kFatRowBytes);
#pragma unroll
    for (uint32_t q = 0; q < kFatRowBytes / 16; ++q) dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

// One thread per mismatch entry e = 3 * origin row + slot (fat_layout.h): p_c of the origin row,
// q1 = LF(p_c), and the landing after the NEXT base for each thing that base can do.
template <bool kDeep>
__global__ __launch_bounds__(256) void fat_mis_pack_kernel(PlainLevel P, FatTable T, const uint32_t *__restrict__ rho,
                                                           const uint32_t *__restrict__ rho_first,
                                                           const uint8_t *__restrict__ sflags, uint8_t *__restrict__ entries) {
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (uint64_t)T.n_rho * kFatSlots) return;
    constexpr uint32_t kDwords = (kDeep ? 2 * kMisBytes : kMisBytes) / 4;
    uint32_t w[kDwords];
#pragma unroll
    for (uint32_t q = 0; q < kDwords; ++q) w[q] = 0;
    const uint32_t i = rho_first[e / kFatSlots], slot = (uint32_t)(e % kFatSlots);
    if ((sflags[i] >> slot) & 1u) {
        const uint32_t a_dense = T.cmap[P.meta[i] & 0xFFu];
        uint32_t tj;
        uint64_t to;
        fat_slot_target(P, T, i, slot < a_dense ? slot : slot + 1, tj, to);      // exists: the flag says so
        plain_lf(P, tj, to);                                                      // q1 = LF(p_c)
        const uint32_t J1 = tj;
        const uint64_t P1 = to;
        const uint32_t m1 = P.meta[J1], a1 = T.cmap[m1 & 0xFFu], v1 = sflags[J1] & 7u;
        w[kMisJ1] = J1;
        w[kMisP1] = (uint32_t)P1 | ((m1 & 0xFFu) << 16) | ((m1 >> 8) << 24);
        w[kMisVal] = v1;
        for (uint32_t o = 0; o < 4; ++o) {
            uint32_t j = J1;
            uint64_t t = P1;
            if (o) {                                                              // a mismatch on slot o - 1 of q1's origin row
                if (!((v1 >> (o - 1)) & 1u)) continue;
                fat_slot_target(P, T, J1, (o - 1) < a1 ? o - 1 : o, j, t);
            }
            plain_lf(P, j, t);
            const uint32_t m = P.meta[j];
            w[kMisJ + o] = j;
            w[kMisRho + o] = rho[j];
            put_half(w, 4 * kMisP + 2 * o, (uint32_t)t);
            put_byte(w, 4 * kMisCh + o, m & 0xFFu);
            put_byte(w, 4 * kMisCid + o, m >> 8);
            w[kMisVal] |= (uint32_t)(sflags[j] & 7u) << (4 + 4 * o);
            if constexpr (kDeep) {                                                // one more step, should the next base match
                plain_lf(P, j, t);
                const uint32_t m2 = P.meta[j];
                w[kMisDeep + kMisJ - 2 + o] = j;
                w[kMisDeep + kMisRho - 2 + o] = rho[j];
                put_half(w, 4 * (kMisDeep + kMisP - 2) + 2 * o, (uint32_t)t);
                put_byte(w, 4 * (kMisDeep + kMisCh - 2) + o, m2 & 0xFFu);
                put_byte(w, 4 * (kMisDeep + kMisCid - 2) + o, m2 >> 8);
                w[kMisDeep + kMisVal - 2] |= (uint32_t)(sflags[j] & 7u) << (4 * o);
            }
        }
    }
    uint4 *dst = reinterpret_cast<uint4 *>(entries + e * (kDwords * 4));
#pragma unroll
    for (uint32_t q = 0; q < kDwords / 4; ++q) dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

// One refinement pass into the plain form.  COLBWT_OK / COLBWT_ERR_NOMEM (HBM, row limit) /
// COLBWT_ERR_HIP; nothing stays allocated on failure (DevPtr).
template <class Src>
int refine_plain(const Src &S, uint64_t src_rows, uint64_t n, PlainLevel &out, PlainBuffers &buf, std::string &err) {
    DevPtr first, park;
    uint64_t total = 0;
    {
        const int rc = count_and_scan(S, src_rows, first, total, err);
        if (rc != COLBWT_OK) return rc;
    }
    if (total > 0xFFFFFFFEull) {
        err = "line-row layout needs " + std::to_string(total) + " rows at one of its levels (> 2^32-2)";
        return COLBWT_ERR_NOMEM;
    }
    // The final level has at least as many rows as this one, each a 128-byte line: when that alone
    // is beyond what can still be allocated the build cannot succeed -- say so after the counting
    // pass, before tens of GB are allocated and filled only to be given back.
    if (total * kFatRowBytes > dev_available_bytes()) {
        if (getenv("COLBWT_ALLOC_LOG"))
            fprintf(stderr, "[colbwt alloc] ? %llu rows x 128 bytes = %.2f GB against %.2f GB available\n", (unsigned long long)total,
                    total * 128e-9, dev_available_bytes() * 1e-9);
        err = "line-row layout: " + std::to_string(total) + " rows at a refinement level need more HBM than is available";
        return COLBWT_ERR_NOMEM;
    }
    const uint32_t r_new = (uint32_t)total;
    SK_TRY(buf.idx.alloc(((uint64_t)r_new + 2) * sizeof(uint64_t)));
    SK_TRY(buf.I.alloc(((uint64_t)r_new + 1) * sizeof(uint32_t)));
    SK_TRY(buf.O.alloc(((uint64_t)r_new + 1) * sizeof(uint16_t)));
    SK_TRY(buf.meta.alloc(((uint64_t)r_new + 1) * sizeof(uint16_t)));
    SK_TRY(buf.thr.alloc(((uint64_t)r_new + 1) * sizeof(uint64_t)));
    SK_TRY(buf.org.alloc((uint64_t)r_new + 1));
    SK_TRY(park.alloc(((uint64_t)r_new + 1) * sizeof(uint64_t)));
    uint64_t *const d_idx = buf.idx.as<uint64_t>(), *const d_thr = buf.thr.as<uint64_t>(), *const d_park = park.as<uint64_t>();
    uint32_t *const d_I = buf.I.as<uint32_t>(), *const d_first = first.as<uint32_t>();
    uint16_t *const d_O = buf.O.as<uint16_t>(), *const d_meta = buf.meta.as<uint16_t>();
    uint8_t *const d_org = buf.org.as<uint8_t>();
    const uint32_t rblocks = (uint32_t)((src_rows + 255) / 256);
    hipLaunchKernelGGL(plain_emit_kernel<Src>, dim3(rblocks), dim3(256), 0, 0, S, (const uint32_t *)d_first, d_idx, d_I, d_meta,
                       d_thr, d_org, d_park, r_new);
    SK_TRY(hipGetLastError());
    SK_TRY(hipStreamSynchronize(0));
    const uint32_t nblocks = (uint32_t)(((uint64_t)r_new + 255) / 256);
    hipLaunchKernelGGL(plain_link_kernel, dim3(nblocks), dim3(256), 0, 0, (const uint32_t *)d_first, (const uint64_t *)d_idx,
                       (const uint64_t *)d_park, d_I, d_O, r_new);
    SK_TRY(hipGetLastError());
    SK_TRY(hipStreamSynchronize(0));
    out.idx = d_idx;
    out.I = d_I;
    out.O = d_O;
    out.meta = d_meta;
    out.thr = d_thr;
    out.org = d_org;
    out.n = n;
    out.r = r_new;
    return COLBWT_OK;
}

// HBM an AUTO open leaves free (of what it could still allocate) when it chooses deep mismatch entries.
constexpr uint64_t kDeepReserve = 32ull << 30;

template <int K>
int build_fat_steps(const DevTable &T1, const HintChars &chars, int mismatch_lines, FatTable &out, FatBuffers &buf, std::string &err,
                    const std::function<void()> &source_done, int &failed_level) {
    failed_level = 2;
    const uint8_t *cmap = T1.cmap;
    const uint32_t sigma = T1.sigma;
    PlainLevel cur{};
    PlainBuffers cur_buf;
    LoadClock clock;
    {
        const SrcL1 s1{T1, chars};
        const int rc = refine_plain(s1, T1.r, T1.n, cur, cur_buf, err);    // level 2 (cuts at thresholds too)
        if (rc != COLBWT_OK) return rc;
    }
    clock.lap("  refinement level 2");
    source_done();
    clock.lap("  one-step tables freed");
    for (int level = 3; level <= K; ++level) {
        failed_level = level;
        PlainLevel next{};
        PlainBuffers next_buf;
        const SrcPlain sp{cur};
        const int rc = refine_plain(sp, cur.r, cur.n, next, next_buf, err);
        if (rc != COLBWT_OK) return rc;
        cur_buf.release();
        cur = next;
        cur_buf.idx = std::move(next_buf.idx);
        cur_buf.I = std::move(next_buf.I);
        cur_buf.O = std::move(next_buf.O);
        cur_buf.meta = std::move(next_buf.meta);
        cur_buf.thr = std::move(next_buf.thr);
        cur_buf.org = std::move(next_buf.org);
        clock.lap("  refinement level (3..K)");
    }

    failed_level = K + 1;                 // from here on it is the size of the final tables that may not fit
    const uint32_t r = cur.r;
    out = FatTable{};
    out.n = cur.n;
    out.r = r;
    out.sigma = sigma;
    out.cmap = cmap;
    out.nblk = (uint32_t)(((uint64_t)r + (1u << kBlockShift) - 1) >> kBlockShift);
    out.steps = K;
    out.top4 = 0;
    for (uint32_t k = 0; k < 4; ++k) out.top4 |= (uint32_t)chars.c[k < sigma ? k : 0] << (8 * k);

    SK_TRY(buf.chr.alloc((uint64_t)r + 64));
    uint8_t *const d_chr = buf.chr.as<uint8_t>();
    const uint32_t nblocks = (uint32_t)(((uint64_t)r + 255) / 256);
    hipLaunchKernelGGL(plain_chars_kernel, dim3(nblocks), dim3(256), 0, 0, cur.meta, r, d_chr);
    SK_TRY(hipGetLastError());
    const uint64_t entries = (uint64_t)out.nblk * sigma;
    SK_TRY(buf.next.alloc((entries ? entries : 1) * sizeof(uint32_t)));
    SK_TRY(buf.prev.alloc((entries ? entries : 1) * sizeof(uint32_t)));
    uint32_t *const d_next = buf.next.as<uint32_t>(), *const d_prev = buf.prev.as<uint32_t>();
    hipLaunchKernelGGL(chr_block_first_last_kernel, dim3((out.nblk + 3) / 4), dim3(256), 0, 0, (const uint8_t *)d_chr, r, out.nblk,
                       sigma, cmap, d_next, d_prev);
    SK_TRY(hipGetLastError());
    SK_TRY(hipStreamSynchronize(0));
    {
        std::vector<uint32_t> first(entries), last(entries), next, prev;
        SK_TRY(hipMemcpy(first.data(), d_next, entries * sizeof(uint32_t), hipMemcpyDeviceToHost));
        SK_TRY(hipMemcpy(last.data(), d_prev, entries * sizeof(uint32_t), hipMemcpyDeviceToHost));
        finish_jump_tables(first, last, out.nblk, sigma, next, prev);
        SK_TRY(hipMemcpy(d_next, next.data(), entries * sizeof(uint32_t), hipMemcpyHostToDevice));
        SK_TRY(hipMemcpy(d_prev, prev.data(), entries * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    out.chr = d_chr;
    out.next_tbl = d_next;
    out.prev_tbl = d_prev;
    out.idx = cur.idx;
    out.thr = cur.thr;

    clock.lap("  characters, jump tables");
    // ---- origin rows and which of their mismatch entries exist (mismatch-line variant)
    DevPtr rho_buf, rho_first_buf, sflags_buf;
    uint64_t mis_lines = 0;
    uint32_t *d_rho = nullptr, *d_rho_first = nullptr;     // raw pointers for the launches
    uint8_t *d_sflags = nullptr;
    if (mismatch_lines) {
        SK_TRY(rho_buf.alloc(((uint64_t)r + 1) * sizeof(uint32_t)));
        d_rho = rho_buf.as<uint32_t>();
        const uint8_t *const d_org = cur.org;
        hipLaunchKernelGGL(org_flags_kernel, dim3(nblocks), dim3(256), 0, 0, d_org, r, d_rho);
        SK_TRY(hipGetLastError());
        uint64_t n_rho = 0;
        {
            const int rc = exclusive_scan_u32(d_rho, r, n_rho, err);
            if (rc != COLBWT_OK) return rc;
        }
        if (mismatch_lines == 3) {
            // AUTO: deep entries (6 % faster on C2, profiles/r03y_deep_vs_plain_entries_last_build.jsonl)
            // when the table then still leaves kDeepReserve of what can be allocated now -- room for
            // read batches and their results; else the 64-byte entries (half the entry table)
            const uint64_t deep_bytes = ((uint64_t)r + 1 + n_rho * kFatSlots) * kFatRowBytes;
            mismatch_lines = deep_bytes + kDeepReserve <= dev_available_bytes() ? 2 : 1;
        }
        const uint32_t entry_shift = mismatch_lines == 2 ? 7u : 6u;           // 2: deep entries, a line each
        mis_lines = ((n_rho * kFatSlots << entry_shift) + kFatRowBytes - 1) / kFatRowBytes;
        if ((uint64_t)r + 1 + mis_lines > 0xFFFFFFFEull) {
            err = "line rows + mismatch lines need " + std::to_string((uint64_t)r + 1 + mis_lines) + " lines (> 2^32-2)";
            return COLBWT_ERR_NOMEM;
        }
        out.n_rho = (uint32_t)n_rho;
        out.slot_line0 = r + 1;
        out.entry_shift = entry_shift;
        SK_TRY(rho_first_buf.alloc((n_rho + 1) * sizeof(uint32_t)));
        SK_TRY(sflags_buf.alloc((uint64_t)r + 1));
        d_rho_first = rho_first_buf.as<uint32_t>();
        d_sflags = sflags_buf.as<uint8_t>();
        hipLaunchKernelGGL(rho_kernel, dim3(nblocks), dim3(256), 0, 0, d_org, r, d_rho, d_rho_first);
        SK_TRY(hipGetLastError());
        hipLaunchKernelGGL(fat_slot_flags_kernel, dim3(nblocks), dim3(256), 0, 0, cur, out, (const uint32_t *)d_rho,
                           (const uint32_t *)d_rho_first, d_sflags);
        SK_TRY(hipGetLastError());
        SK_TRY(hipStreamSynchronize(0));
        clock.lap("  origin rows, entry flags");
    }
    {
        PlainAllocScope whole;             // the table the query fetches from: one hipMalloc block (dev_mem.h)
        SK_TRY(buf.lines.alloc(((uint64_t)r + 1 + mis_lines) * kFatRowBytes));
    }
    clock.lap("  lines allocated");
    uint8_t *const d_lines = buf.lines.as<uint8_t>();
    SK_TRY(hipMemset(d_lines + (uint64_t)r * kFatRowBytes, 0, kFatRowBytes));
    if (mismatch_lines) {
        hipLaunchKernelGGL((fat_pack_kernel<K, true>), dim3(nblocks), dim3(256), 0, 0, cur, out, (const uint32_t *)d_rho,
                           (const uint8_t *)d_sflags, d_lines);
        SK_TRY(hipGetLastError());
        uint8_t *const d_entries = d_lines + ((uint64_t)r + 1) * kFatRowBytes;
        if (mis_lines) SK_TRY(hipMemset(d_entries + (mis_lines - 1) * kFatRowBytes, 0, kFatRowBytes));   // the odd half of the last line
        const uint64_t n_entries = (uint64_t)out.n_rho * kFatSlots;
        if (mismatch_lines == 2)
            hipLaunchKernelGGL(fat_mis_pack_kernel<true>, dim3((uint32_t)((n_entries + 255) / 256)), dim3(256), 0, 0, cur, out,
                               (const uint32_t *)d_rho, (const uint32_t *)d_rho_first, (const uint8_t *)d_sflags, d_entries);
        else
            hipLaunchKernelGGL(fat_mis_pack_kernel<false>, dim3((uint32_t)((n_entries + 255) / 256)), dim3(256), 0, 0, cur, out,
                               (const uint32_t *)d_rho, (const uint32_t *)d_rho_first, (const uint8_t *)d_sflags, d_entries);
    } else {
        hipLaunchKernelGGL((fat_pack_kernel<K, false>), dim3(nblocks), dim3(256), 0, 0, cur, out, (const uint32_t *)nullptr,
                           (const uint8_t *)nullptr, d_lines);
    }
    SK_TRY(hipGetLastError());
    SK_TRY(hipStreamSynchronize(0));
    out.lines = d_lines;
    clock.lap("  lines packed");
    buf.idx = std::move(cur_buf.idx);
    buf.thr = std::move(cur_buf.thr);
    cur_buf.release();          // I, O, meta: only the pack pass read them
    clock.lap("  level arrays freed");
    return COLBWT_OK;
}

// ---- read sampler over line rows (read_sampler.h) ---------------------------------------
struct FatView {
    FatTable T;
    struct Row {
        uint32_t i1, o1_len, ch;
    };
    __device__ __forceinline__ uint64_t n() const { return T.n; }
    __device__ __forceinline__ uint32_t rows() const { return T.r; }
    __device__ __forceinline__ uint64_t idx(uint32_t j) const { return T.idx[j]; }
    __device__ __forceinline__ Row load(uint32_t j) const {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(T.lines + (uint64_t)j * kFatRowBytes);
        Row w;
        w.i1 = p[kFatI / 4];
        w.o1_len = fat_half(p, kFatO) | (fat_half(p, kFatLen) << 16);
        w.ch = fat_byte(p, kFatCh + 7);
        return w;
    }
    __device__ __forceinline__ uint32_t ch(const Row &w) const { return w.ch; }
    __device__ __forceinline__ uint32_t lf_row(const Row &w) const { return w.i1; }
    __device__ __forceinline__ uint32_t lf_off(const Row &w) const { return w.o1_len & 0xFFFFu; }
    __device__ __forceinline__ uint64_t len(uint32_t, const Row &w) const { return w.o1_len >> 16; }
};

__global__ __launch_bounds__(256) void fat_synth_reads_kernel(FatView V, uint64_t n_reads, uint32_t m, uint32_t sub_permille,
                                                              uint64_t seed, uint8_t *__restrict__ bases,
                                                              uint64_t *__restrict__ read_off) {
    const uint64_t rd = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (rd > n_reads) return;
    read_off[rd] = rd * m;
    if (rd == n_reads) return;
    sample_read(V, rd, m, sub_permille, seed, bases + rd * m);
}

}  // namespace

void FatBuffers::release() {
    for (DevPtr *p : {&lines, &chr, &idx, &thr, &next, &prev}) p->reset();
}

uint64_t FatBuffers::bytes() const {
    return lines.bytes() + chr.bytes() + idx.bytes() + thr.bytes() + next.bytes() + prev.bytes();
}

bool fat_steps_supported(int steps) {
#define X(K) if (steps == K) return true;
    COLBWT_FAT_STEPS(X)
#undef X
    return false;
}

// Builds the line-row layout with `steps` own steps from the one-step tables.  Same contract as
// build_sk.
int build_fat(const DevTable &T, const HintChars &chars, int steps, int mismatch_lines, FatTable &out, FatBuffers &buf,
              std::string &err, const std::function<void()> &source_done, int *failed_level) {
    int rc = COLBWT_ERR_ARG, level = 0;
    err = "unsupported number of line-row steps";
#define X(K) if (steps == K) rc = build_fat_steps<K>(T, chars, mismatch_lines, out, buf, err, source_done, level);
    if (failed_level) *failed_level = 0;
    COLBWT_FAT_STEPS(X)
#undef X
    if (rc != COLBWT_OK) buf.release();
    if (rc != COLBWT_OK && failed_level) *failed_level = level;
    return rc;
}

void launch_fat_synth_reads(const FatTable &T, uint64_t n_reads, uint32_t read_len, uint32_t sub_permille, uint64_t seed,
                            uint8_t *d_bases, uint64_t *d_read_off, hipStream_t stream) {
    const uint32_t blocks = (uint32_t)((n_reads + 1 + 255) / 256);
    hipLaunchKernelGGL(fat_synth_reads_kernel, dim3(blocks), dim3(256), 0, stream, FatView{T}, n_reads, read_len, sub_permille,
                       seed, d_bases, d_read_off);
}

}  // namespace colbwt
