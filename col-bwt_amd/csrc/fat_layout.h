// fat_layout.h -- "line rows": one 128-byte HBM line per row of the refined run table.
//
// What a random row costs on MI355X is the 128-byte line fill, not the bytes used
// (DESIGN.md 4.1), and a line fetched lane-cooperatively -- eight lanes, 16 bytes each, ONE
// instruction -- costs the texture addresser what a single 16-byte load costs
// (tools/gather_bench modes 15 / 16: 48-51 G whole lines/s on 16-128 GiB tables, against
// 39.5 G/s for a 32-byte row fetched with two per-lane loads and 10 G/s for a line fetched
// with eight).  So the row takes the whole line and spends it on everything that saves a
// later, dependent line fill:
//
//   * K-step look-ahead (sk_layout.h) with K up to 8: level-K rows are the level-(K-1) rows
//     split at the pre-images of their boundaries, so all positions of a row walk through the
//     same original rows for K-1 LF steps; the characters / col ids met on the way and the
//     landings of LF .. LF^K are constants of the row.
//   * for each of the three most frequent OTHER characters c (hint slots, device_layout.h):
//     where a mismatch on c re-orients to (col_bwt.hpp:531-574) is ONE position p_c for the
//     whole row -- head of the succeeding c-run or tail of the preceding one, decided by the
//     row's place relative to the threshold (rows are cut at thresholds, sk_build.hip) -- so
//     the row also stores the landings of LF and LF^2 from p_c and the character / col id
//     met on the way.  A mismatch then costs no line fill of its own: the lane leaves the
//     row it mismatched in directly for where the reference is two steps later.
//
// Row bytes (little endian; K <= 8 own steps, 2 steps per mismatch slot), laid out for how the
// query reads them -- two 16-byte pieces always, then one dword, one halfword and one slot:
//   [  0,   8)  CH    ch[a] at byte 8 - a: the character met after a - 1 LF steps (ch[1] = the
//                     row's own); compared with the next 8 read bases in one 64-bit XOR
//   [  8,  16)  CID   cid[a] at byte 8 - a: the col id reported there (col_bwt.hpp:513)
//   [ 16,  18)  len   row length (<= 65534)
//   [ 18]       flags bit s: mismatch slot s is valid; bits 4-6: dense index of the row's
//                     character among the four most frequent (7 = other)
//   [ 20,  28)  cut[1..8]  cut_a | len_b << 4 for the jump LF^s (sk_layout.h): where the image
//                     leaves row I[s] and row I[s] + 1; 15 = none
//   [ 32,  64)  I[1..8]    level-K row holding LF^s(first position of the row)
//   [ 64,  80)  O[1..8]    offset of that image inside I[s]
//   [ 80, 128)  3 slots of 16 bytes: J[1..2] dwords, P[1..2] halfwords: exact landing (row,
//                     offset) of LF^a(p_c), fast-forward included (one position: nothing left to
//                     cut); tch2, tcid2: character / col id met after one LF step from p_c
//   entries beyond K are zero.
// Side arrays (cold: rare characters, read sampler, load-time kernels):
//   chr[r] u8, idx[r+1] u64 (idx[r] = n), thr[r] u64, next_tbl / prev_tbl per 256-row block.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"

namespace colbwt {

// the numbers of own steps K compiled in
#define COLBWT_FAT_STEPS(X) X(4) X(5) X(6) X(7) X(8)

constexpr uint32_t kFatRowBytes = 128;
constexpr uint32_t kFatSlots = 3;
constexpr uint32_t kFatSlotSteps = 2;
constexpr uint32_t kFatOwnOther = 7;     // flags bits 4-6: the row's character is not one of the top four
constexpr uint32_t kFatCh = 0, kFatCid = 8, kFatLen = 16, kFatFlags = 18, kFatCut = 20, kFatI = 32, kFatO = 64,
                   kFatSlot0 = 80, kFatSlotBytes = 16;
// inside a slot
constexpr uint32_t kFatSlotP = 8, kFatSlotCh2 = 12, kFatSlotCid2 = 13;

struct FatTable {
    const uint8_t *lines;     // r rows of 128 bytes (+ one zero row)
    const uint8_t *chr;       // r
    const uint64_t *idx;      // r + 1
    const uint64_t *thr;      // r
    const uint32_t *next_tbl; // nblk * sigma : first row >= b * 256 holding c
    const uint32_t *prev_tbl; // nblk * sigma : last row < b * 256 holding c
    const uint8_t *cmap;      // 256
    uint64_t n;
    uint32_t r;
    uint32_t sigma;
    uint32_t nblk;
    uint32_t steps;           // K
    uint32_t top4;            // the four most frequent characters, byte k = dense index k
    uint32_t *claim;          // kFatClaimSets x kFatClaimBlocks chunk counters of the query's persistent workgroups
};
constexpr uint32_t kFatClaimSets = 16, kFatClaimBlocks = 4096;

// byte / halfword / dword k of a row image held as dwords
__device__ __forceinline__ uint32_t fat_byte(const uint32_t *w, uint32_t off) { return (w[off >> 2] >> (8 * (off & 3u))) & 0xFFu; }
__device__ __forceinline__ uint32_t fat_half(const uint32_t *w, uint32_t off) { return (w[off >> 2] >> (8 * (off & 2u))) & 0xFFFFu; }

// Dense index (0..3) of byte c among the four most frequent characters, 4 = none of them.
__device__ __forceinline__ uint32_t fat_top_index(uint32_t top4, uint32_t c) {
    uint32_t x = 4;
    x = ((top4 >> 24) & 0xFFu) == c ? 3u : x;
    x = ((top4 >> 16) & 0xFFu) == c ? 2u : x;
    x = ((top4 >> 8) & 0xFFu) == c ? 1u : x;
    x = (top4 & 0xFFu) == c ? 0u : x;
    return x;
}

// succ_char / pred_char (LF_table.hpp:286-298 / 271-283) over the byte-per-row character
// array: the rest of the 256-row block, then one jump-table lookup.  Rare path (characters
// beyond the mismatch slots) and load-time kernels.
__device__ __forceinline__ uint32_t fat_succ_char(const FatTable &T, uint32_t i, uint32_t c, uint32_t cidx) {
    const uint32_t blk = i >> kBlockShift;
    const uint64_t lim = (((uint64_t)blk + 1) << kBlockShift);
    const uint64_t end = lim < T.r ? lim : T.r;
    for (uint64_t s = (uint64_t)i + 1; s < end; ++s)
        if (T.chr[s] == c) return (uint32_t)s;
    if (blk + 1 < T.nblk) return T.next_tbl[(uint64_t)(blk + 1) * T.sigma + cidx];
    return kNone;
}
__device__ __forceinline__ uint32_t fat_pred_char(const FatTable &T, uint32_t i, uint32_t c, uint32_t cidx) {
    const uint32_t blk = i >> kBlockShift;
    const uint32_t first = blk << kBlockShift;
    for (uint32_t q = i; q > first;) {
        --q;
        if (T.chr[q] == c) return q;
    }
    if (blk > 0) return T.prev_tbl[(uint64_t)blk * T.sigma + cidx];
    return kNone;
}

}  // namespace colbwt
