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
//
// ---- line rows with MISMATCH LINES (COLBWT_LAYOUT_LINE_ROWS_MISMATCH_LINES; slot_line0 != 0) ----
// Where threshold_step goes on a mismatch is ONE position p_c for every position of an ORIGINAL
// row (a row of the file, cut at the thresholds that fall inside it: call it an origin row) and a
// character c -- the in-row slots above store that fact once per REFINED row (5.25 per origin row
// at K = 8), which is why they have room for one outcome only: "the base after the mismatch
// matches".  When it does not (60 % of the mismatches on the C2 workload) the lane lands on the
// next row only to find another mismatch: in a stretch of consecutive mismatches every base costs
// a line fill.  Here the fact is stored once per (origin row, character) in a table of its own,
// and each entry has room for EVERY outcome of the base after the mismatch:
//   * rows keep [0, 80) as above (the read sampler and the side arrays are shared) and hold, in
//     place of the slots, RHO[1..8] = the origin row met after a - 1 LF steps and VAL = which of
//     its three mismatch entries exist; a read base that differs from CH[a] sends the lane to the
//     entry (RHO[a], slot) directly -- the next trip fetches the entry's line, not a row;
//   * a mismatch entry (64 bytes, two per line, behind the rows in the same allocation: line
//     slot_line0 + e / 2, e = 3 * rho + slot) resolves the mismatching base AND the base after it
//     whatever it is: q1 = LF(p_c), the character / col id met there, and for each of the four
//     things the next base can do -- match, or mismatch on one of the three slot characters of
//     q1's origin row -- the exact landing after that base, the character and col id met there,
//     its origin row and which of ITS entries exist.  If the base after those two mismatches as
//     well, the lane goes from entry to entry: a stretch of mismatches costs a line fill per two
//     bases, and a trip never ends on a row just to learn that it mismatches.
// Entry dwords: [0] J1 | [1] P1 | t1 << 16 | d1 << 24 | [2..5] J[0..3] | [6..9] rho[0..3] |
//   [10..11] P[0..3] halfwords | [12] ch[0..3] bytes | [13] cid[0..3] bytes |
//   [14] v1 | vo[0] << 4 | vo[1] << 8 | vo[2] << 12 | vo[3] << 16 (3 valid bits each) | [15] 0.
//   outcome 0 = the next base matches t1; outcome 1 + s = it is slot character s of q1's origin row.
// DEEP entries (COLBWT_LAYOUT_MISMATCH_LINES_DEEP; entry_shift = 7): a whole line per entry, the first
// 64 bytes as above, the second half one more step for every outcome -- if the base after the two
// MATCHES the character the outcome's landing meets, it is resolved too: dwords [16..19] J2[0..3]
// (landing after it) | [20..23] rho2[0..3] | [24..25] P2[0..3] halfwords | [26] ch2[0..3] |
// [27] cid2[0..3] | [28] v2[o] << 4 o: what is met THERE, for the base after that.  Between two
// mismatches of a stretch there is often exactly one matching base; with it resolved in the entry
// the lane goes from entry to entry without the row visit in between.
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
// rows of the mismatch-line variant: valid bits of the origin row met after a - 1 steps at bits
// 3 (a - 1) of the three bytes at kFatVal; RHO[a] dwords at kFatRho
constexpr uint32_t kFatVal = 28, kFatRho = 80;
// a mismatch entry
constexpr uint32_t kMisBytes = 64, kMisJ1 = 0, kMisP1 = 1, kMisJ = 2, kMisRho = 6, kMisP = 10, kMisCh = 12, kMisCid = 13, kMisVal = 14;
constexpr uint32_t kMisDeep = 16;        // dword offset of the second half of a deep entry: the same fields one step further

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
    uint32_t slot_line0;      // mismatch-line variant: line number (128-byte units of `lines`) of entry 0; 0 = in-row slots
    uint32_t n_rho;           // origin rows (entries: 3 per origin row)
    uint32_t entry_shift;     // log2 of the bytes per mismatch entry: 6, or 7 = deep entries
};

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
