// fat2_query.hip -- the PML / col-ID query over line rows with MISMATCH LINES (fat_layout.h).
//
// Same per-base semantics as every other layout (col_bwt.hpp:498-574, LF_table.hpp:251-298), the
// same machinery as fat_query.hip -- one lane per read at a time, persistent lanes claiming chunks
// of reads, ONE 128-byte line per lane and trip fetched lane-cooperatively into LDS, two output
// collectors flushed by the whole wave -- but a trip's line is one of two kinds:
//
//   * a ROW: up to K <= 8 matching bases are consumed with one 64-bit compare, as before.  What is
//     new is what happens when the compare stops at a base that differs: the row knows the origin
//     row met at that depth and which of its mismatch entries exist, so the lane leaves for the
//     ENTRY (origin row, character) -- it never lands on a row only to find that it mismatches;
//   * a mismatch ENTRY (64 bytes, half a line): resolves the mismatching base (length 0, the col
//     id the row it happened in carried) AND the base after it whatever it is -- a match, or a
//     mismatch on one of the three slot characters there, each with its exact landing -- and knows
//     the character met after that: if the third base differs too the lane goes on to the next
//     entry, else to the landing row.
//
// In a stretch of mismatching bases (after a substitution the walk is somewhere else in the BWT and
// mismatches every second or third base: two thirds of all trips on the C2 workload) that is one
// line fill per two bases instead of one per base.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "device_layout.h"
#include "fat_cursor.h"
#include "fat_layout.h"
#include "lane_io.h"
#include "lane_out.h"
#include "query_kernels.h"

namespace colbwt {

namespace {

constexpr uint32_t kOffMis0 = 0xFFFFFFFCu;      // the lane's line is a mismatch entry: first half of the line
constexpr uint32_t kOffMis1 = 0xFFFFFFFBu;      // ... second half.  In this state L holds the col id to report for the mismatching base.

#ifdef COLBWT_COUNT_TRIPS
#define FAT2_STAT(k) (++stat[k])
#else
#define FAT2_STAT(k) ((void)0)
#endif

// slot (0..2) of read byte c among the mismatch entries of an origin row whose character is `own`,
// 3 = none (c or too many characters beyond the four most frequent)
// top: byte -> its index among the four most frequent characters, 4 = none of them (a table in LDS:
// the three uses per trip were 60 vector instructions of compares and selects)
__device__ __forceinline__ uint32_t mis_slot(const uint8_t *top, uint32_t own, uint32_t c) {
    const uint32_t aidx = top[own], cidx = top[c];
    const uint32_t slot = cidx < aidx ? cidx : cidx - 1;
    return cidx < 4 && slot < kFatSlots ? slot : kFatSlots;
}

template <int K, typename PmlT, bool kDeep>
__global__ __launch_bounds__(kQueryBlock)
void fat2_query_kernel(FatTable T, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ read_off,
                       uint64_t n_reads, uint32_t big_reads, uint32_t tail_permille,
                       PmlT *__restrict__ pml, uint8_t *__restrict__ cid) {
    constexpr bool kWide = sizeof(PmlT) == 4;
    __shared__ uint4 s_stage[kWaves][8][64];       // per wave: instruction q's 64 x 16 bytes
    __shared__ uint4 s_win[kWaves][4][64];         // read bytes (lane_io.h LaneWindow)
    __shared__ uint32_t s_claim;                   // the workgroup's chunk counter (ChunkPlan)
    __shared__ uint8_t s_top[256];                 // byte -> index among the four most frequent characters (mis_slot)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, g8 = lane & ~7u, p = lane & 7u;
    uint32_t *const s_jx = reinterpret_cast<uint32_t *>(&s_stage[wave][7][48]);   // 64 dwords
    ChunkPlan plan;
    plan.init(n_reads, big_reads, tail_permille);
    if (threadIdx.x == 0) s_claim = 0;
    s_top[threadIdx.x] = (uint8_t)fat_top_index(T.top4, threadIdx.x);
    __syncthreads();
    uint32_t *const claim = &s_claim;
    ReadCursor rc;
    bool done;
    rc.c_next = threadIdx.x;                          // the first chunk is the lane's own number
    rc.request_chunk(plan, read_off);
    rc.commit();
    done = !rc.enter_chunk(plan, claim);

    OutRuns acc;                                    // what the lane has reported and not yet stored (lane_out.h)
    uint32_t trip = 0;                              // the wave flushes every OutRuns::kPeriod-th trip
    LaneWindow win;
    win.init(rc.off + rc.k - 1);
    uint4 (*const my_win)[64] = s_win[wave];

    // col_bwt.hpp:503-508: pos = n - 1 = the last position of the last row, expressed as an
    // arrival at the last row that clamps to len - 1.
    uint32_t j = done ? 0u : T.r - 1;               // the LINE the lane wants next: a row, or the line of an entry
    uint32_t o = kOffLastPos;
    uint32_t L = 0;
    // this lane's line, piece x: s_stage[wave][p][g8 + (x ^ p)]
    const uint4 *const my_row = &s_stage[wave][p][g8];

#ifdef COLBWT_COUNT_TRIPS
    unsigned long long stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    while (__any(!done)) {
        // ---- (1) where the lane stands (registers and LDS only, fat_query.hip).  A lane whose chunk is
        // reported enters its next one only once the wave's flush has taken what it still holds: the
        // collector describes ONE stretch of addresses (1.5 idle trips on average, once per chunk).
        bool step_back = false;
        if (!done && rc.k == 0 && !rc.next_in_flight && (rc.r != rc.r_lo || rc.nc_ready)) {
            if (rc.r != rc.r_lo) {
                rc.r -= 1;
                rc.k = rc.off - rc.next_off;
                rc.off = rc.next_off;
                step_back = rc.r > rc.r_lo;
            } else if (kWide || acc.cnt == 0) {
                done = !rc.enter_chunk(plan, claim);
                if (!done) win.init(rc.off + rc.k - 1);
            }
            j = done ? 0u : T.r - 1;
            o = kOffLastPos;
            L = 0;
        }
        const bool live = !done && rc.k != 0;                // an empty read idles for one trip
        uint64_t &k = rc.k;
        const uint64_t g = rc.off + k - 1;                   // :512 pattern[m-i-1] is the next base

        // ---- (2) the wave's 64 lines, 128 bytes each, into LDS
        s_jx[lane] = j;
        wave_sync();
        {
            const uint4 ja = *reinterpret_cast<const uint4 *>(&s_jx[g8]);
            const uint4 jb = *reinterpret_cast<const uint4 *>(&s_jx[g8 + 4]);
            const uint32_t jq[8] = {ja.x, ja.y, ja.z, ja.w, jb.x, jb.y, jb.z, jb.w};
            wave_sync();
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q)
                __builtin_amdgcn_global_load_lds(T.lines + (uint64_t)jq[q] * kFatRowBytes + ((p ^ q) << 4), &s_stage[wave][q][0],
                                                 16, 0, 0);
        }
        // ---- (3) the trip's other memory traffic, behind the lines
        if (live && win.avail(g) < (k < 8u ? (uint32_t)k : 8u)) win.request(my_win, bases, g);
        if (step_back) { rc.in_next = read_off[rc.r - 1]; rc.next_in_flight = true; }
        if (!done && rc.fetch_pending) rc.request_chunk(plan, read_off);
        lds_dma_landed();
        rc.commit();

        const uint32_t have = live ? win.avail(g) : 0u;      // read bytes at hand
        // what the trip reports: `consumed` bases, lengths l_new - e for element e (keep = 0: all 0),
        // col ids byte e of `ids`; pushed once below, whatever kind of line the lane had
        uint32_t consumed = 0, l_new = 0, keep = 0xFFFFFFFFu, keep1 = 0xFFFFFFFFu;
        uint64_t ids = 0;
        if (!live || have == 0) FAT2_STAT(5);
        if (live && have != 0) {
            const uint64_t W = win.get8(my_win, lane, g);    // byte 7 = the next base
            const uint32_t left = k < have ? (uint32_t)k : have;   // bases this trip may consume (<= 64: 32-bit compares below)
            if (o == kOffMis0 || o == kOffMis1) {
                // ---- a mismatch entry: the next base did not match where the lane came from
                // (col_bwt.hpp:520-523: length 0, threshold_step, LF), the one after it is open
                FAT2_STAT(2);
                const uint32_t h4 = !kDeep && o == kOffMis1 ? 4u : 0u;      // a deep entry is the whole line
                const uint4 e0 = my_row[(h4 + 0u) ^ p], e1 = my_row[(h4 + 1u) ^ p], e2 = my_row[(h4 + 2u) ^ p],
                            e3 = my_row[(h4 + 3u) ^ p];
                const uint32_t carry = L;                    // col id of the row the mismatch happened in (:513)
                const uint32_t t1 = (e0.y >> 16) & 0xFFu, d1 = e0.y >> 24, v1 = e3.z & 7u;
                uint32_t oc = 4;                             // which outcome the second base is (4: not resolved here)
                if (left >= 2) {
                    const uint32_t c2 = (uint32_t)(W >> 48) & 0xFFu;
                    const uint32_t s2 = mis_slot(s_top, t1, c2);
                    if (c2 == t1) oc = 0;                    // :516 one step later
                    else if (s2 < kFatSlots && ((v1 >> s2) & 1u)) oc = 1 + s2;
                }
                if (oc < 4) {
                    const uint32_t J = oc == 0 ? e0.z : oc == 1 ? e0.w : oc == 2 ? e1.x : e1.y;
                    const uint32_t rho = oc == 0 ? e1.z : oc == 1 ? e1.w : oc == 2 ? e2.x : e2.y;
                    const uint32_t pw = oc < 2 ? e2.z : e2.w, P = (oc & 1u) ? pw >> 16 : pw & 0xFFFFu;
                    const uint32_t ch = (e3.x >> (8 * oc)) & 0xFFu, cd = (e3.y >> (8 * oc)) & 0xFFu;
                    const uint32_t vo = (e3.z >> (4 + 4 * oc)) & 7u;
                    consumed = 2;
                    l_new = 1;                               // (1, 0) when the second base matches (:517) ...
                    keep = oc == 0 ? 0xFFFFFFFFu : 0u;       // ... (0, 0) when it is a mismatch of its own
                    ids = (uint64_t)(d1 | (carry << 8));
                    L = oc == 0 ? 1u : 0u;
                    // what the lane stands on after the bases it consumed, and what it knows about the next one
                    uint32_t nJ = J, nP = P, nrho = rho, nch = ch, ncd = cd, nv = vo, seen = 2;
                    if constexpr (kDeep) {
                        if (left >= 3 && ((uint32_t)(W >> 40) & 0xFFu) == ch) {
                            // the base after the two matches what the landing meets (:516-517): resolved here too
                            const uint4 f0 = my_row[4u ^ p], f1 = my_row[5u ^ p], f2 = my_row[6u ^ p], f3 = my_row[7u ^ p];
                            nJ = oc == 0 ? f0.x : oc == 1 ? f0.y : oc == 2 ? f0.z : f0.w;
                            nrho = oc == 0 ? f1.x : oc == 1 ? f1.y : oc == 2 ? f1.z : f1.w;
                            const uint32_t pw2 = oc < 2 ? f2.x : f2.y;
                            nP = (oc & 1u) ? pw2 >> 16 : pw2 & 0xFFFFu;
                            nch = (f2.z >> (8 * oc)) & 0xFFu;
                            ncd = (f2.w >> (8 * oc)) & 0xFFu;
                            nv = (f3.x >> (4 * oc)) & 7u;
                            consumed = 3;
                            seen = 3;
                            ids = (uint64_t)(cd | (d1 << 8) | (carry << 16));
                            if (oc == 0) { l_new = 2; keep = 0xFFFFFFFFu; }            // (2, 1, 0)
                            else { l_new = 1; keep = 0xFFFFFFFFu; keep1 = 0u; }        // (1, 0, 0)
                            L += 1;
                        }
                    }
                    if (k > consumed) {
                        // the base after those: does it match where the lane lands?
                        uint32_t sn = kFatSlots;
                        if (left > seen) {
                            const uint32_t cn = (uint32_t)(W >> (8u * (7u - seen))) & 0xFFu;
                            if (cn != nch) sn = mis_slot(s_top, nch, cn);
                        }
                        if (sn < kFatSlots && ((nv >> sn) & 1u)) {
                            const uint32_t e = nrho * kFatSlots + sn;    // straight on to the next entry
                            j = T.slot_line0 + (kDeep ? e : e >> 1);
                            o = kDeep ? kOffMis0 : kOffMis0 - (e & 1u);
                            L = ncd;
                        } else {
                            j = nJ;                          // exact: one position, fast-forward included
                            o = nP;
                        }
                    }
                } else {
                    // the second base is beyond this trip (read end, window) or a character without an
                    // entry: the lane lands where the reference is after the mismatching base alone
                    consumed = 1;
                    l_new = 0;
                    ids = (uint64_t)carry;
                    L = 0;
                    j = e0.x;
                    o = e0.y & 0xFFFFu;
                }
            } else {
                FAT2_STAT(0);
                const uint4 r0 = my_row[0 ^ p];              // CH, CID
                const uint4 r1 = my_row[1 ^ p];              // len | flags << 16, cuts, valid bits
                const uint64_t CH = (uint64_t)r0.x | ((uint64_t)r0.y << 32);
                const uint64_t CID = (uint64_t)r0.z | ((uint64_t)r0.w << 32);
                const uint32_t len = r1.x & 0xFFFFu;
                // ---- how the lane arrives
                uint32_t skip = 0;                           // 1: the row's own character was consumed on the way here
                bool stay = false;                           // this trip only moves on to another row
                if (o == kOffPred) { o = len - 1; skip = 1; }            // LF_table.hpp:282
                else if (o == kOffSucc) { o = 0; skip = 1; }             // LF_table.hpp:296
                else if (o >= len && j < T.r - 1) {
                    o -= len;                                // fast-forward of LF_table::LF (LF_table.hpp:256-259)
                    j += 1;
                    stay = true;
                    FAT2_STAT(1);
                } else {
                    o = o < len ? o : len - 1;
                }
                if (!stay) {
                    // ---- the next 8 read bases against the 8 characters the row's positions meet
                    const uint64_t X = skip ? ((W >> 8) ^ CH) & 0x00FFFFFFFFFFFFFFull : W ^ CH;
                    const uint32_t run = matching_top_bytes(X);          // LF steps that match, the skipped one included
                    const uint32_t cap = left + skip < (uint32_t)K ? left + skip : (uint32_t)K;
                    uint32_t steps = run < cap ? run : cap;
                    bool own_jump = true, to_entry = false;
                    uint32_t e_next = 0, carry = 0;
                    if (run < cap) {
                        // a base that is there differs from the character met after `run` steps
                        // (:516 / :520): depth d = run + 1 <= K, the base after the consumed ones
                        const uint32_t d = run + 1u, sh = 8u * (8u - d);
                        const uint32_t c = (uint32_t)(W >> (sh + 8u * skip)) & 0xFFu;
                        const uint32_t s = mis_slot(s_top, (uint32_t)(CH >> sh) & 0xFFu, c);
                        if (s < kFatSlots && ((r1.w >> (3u * run + s)) & 1u)) {
                            const uint32_t rho = reinterpret_cast<const uint32_t *>(&my_row[((kFatRho / 16) + (run >> 2)) ^ p])[run & 3u];
                            e_next = rho * kFatSlots + s;
                            carry = (uint32_t)(CID >> sh) & 0xFFu;
                            to_entry = true;
                            own_jump = false;
                        }
                    }
                    if (run == 0 && !to_entry) {
                        // :520-523 with no entry to go to: threshold_step at run time
                        const uint32_t c = (uint32_t)(W >> 56);
                        const uint32_t cidx = T.cmap[c];
                        bool moved = false;
                        if (cidx != kAbsent) moved = fat_threshold_scan(T, j, o, c, cidx);
                        FAT2_STAT(moved ? 3 : 4);
                        if (moved) {                         // the target row is the next load
                            steps = 1;                       // one id to report: the row's own
                            consumed = 1;
                            l_new = 0;
                            own_jump = false;
                        } else {
                            // c occurs nowhere: (interval, offset) unchanged (:533-534), LF proceeds
                            // from this row; length restarts at 0
                            const uint32_t st = matching_top_bytes(X & 0x00FFFFFFFFFFFFFFull);
                            const uint32_t cap0 = left < (uint32_t)K ? left : (uint32_t)K;
                            steps = st < cap0 ? st : cap0;
                            consumed = steps;
                            l_new = steps - 1;
                        }
                    } else {
                        consumed = steps - skip;
                        l_new = L + consumed;                // :517 ++length per matching base
                    }
                    L = l_new;
                    ids = steps ? CID >> (8u * (8u - steps)) : 0ull;     // element e <-> step steps - e
                    if (k == consumed) {
                        // the read is done: its last LF (:527) has no observable effect
                    } else if (to_entry) {
                        FAT2_STAT(6);
                        j = T.slot_line0 + (kDeep ? e_next : e_next >> 1);
                        o = kDeep ? kOffMis0 : kOffMis0 - (e_next & 1u);
                        L = carry;
                    } else if (own_jump) {
                        // LF^steps lands at (I, O + o) ... unless the cuts say it is already further on
                        const uint32_t e = steps - 1;
                        const uint32_t I = reinterpret_cast<const uint32_t *>(&my_row[((kFatI / 16) + (e >> 2)) ^ p])[e & 3u];
                        const uint32_t Oh = reinterpret_cast<const uint16_t *>(&my_row[((kFatO / 16) + (e >> 3)) ^ p])[e & 7u];
                        const uint64_t cuts = (uint64_t)r1.y | ((uint64_t)r1.z << 32);
                        const uint32_t cut = (uint32_t)(cuts >> (8 * e)) & 0xFFu, cut_a = cut & 0xFu, len_b = cut >> 4;
                        j = I;
                        if (cut_a != kSKCutNone && o >= cut_a) {
                            j += 1;
                            o -= cut_a;
                            if (len_b != kSKCutNone && o >= len_b) {
                                j += 1;
                                o -= len_b;
                            }
                        } else {
                            o += Oh;
                        }
                    }
                }
            }
            // ---- report the run (:525): element e is the base at g - consumed + 1 + e
            if constexpr (kWide) {
                for (uint32_t e = 0; e < consumed; ++e) {
                    pml[g - consumed + 1 + e] = (PmlT)((e < 2 ? keep : keep1) ? l_new - e : 0u);
                    cid[g - consumed + 1 + e] = (uint8_t)(ids >> (8 * e));
                }
            } else {
                acc.push_run(consumed, l_new, keep, keep1, (uint32_t)ids, (uint32_t)(ids >> 32));
            }
            k -= consumed;
        }
        // ---- (5) every fourth trip: what the lanes hold from a block boundary up, and all of what the
        // lanes hold whose chunk is reported, leaves -- the staged lines are read, their LDS is the
        // flush's working area
        wave_sync();
        if constexpr (!kWide) {
#ifdef COLBWT_ABL_NO_FLUSH         // timing experiments only
            if ((trip & (OutRuns::kPeriod - 1)) == OutRuns::kPeriod - 1) acc.cnt = acc.cnt > 90 ? 1 : 0;
#else
            if ((trip & (OutRuns::kPeriod - 1)) == OutRuns::kPeriod - 1)
                acc.flush_wave((uint16_t *)pml, cid, rc.off + rc.k, !done, rc.k == 0 && rc.r == rc.r_lo, &s_stage[wave][0][0], lane);
#endif
        }
        ++trip;
        wave_sync();   // the next trip overwrites s_jx and the staged lines
    }
#ifdef COLBWT_COUNT_TRIPS
    for (int q = 0; q < 8; ++q) atomicAdd(&g_fat_stats[q], stat[q]);
#endif
}

// Blocks that are resident at once on the device (LDS-bound: 3 per CU): the persistent grid.
template <int K, typename PmlT, bool kDeep>
uint32_t resident_blocks2() {
    static uint32_t cached[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (cached[dev] == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fat2_query_kernel<K, PmlT, kDeep>, kQueryBlock, 0) != hipSuccess || per_cu < 1)
            per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 1;
        (void)hipGetLastError();
        cached[dev] = (uint32_t)per_cu * (uint32_t)cus;
    }
    return cached[dev];
}

template <int K, typename PmlT, bool kDeep>
void launch_typed2(const FatTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, uint64_t n_bases,
                   PmlT *d_pml, uint8_t *d_cid, hipStream_t stream) {
    const uint64_t want_blocks = (n_reads + kQueryBlock - 1) / kQueryBlock;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(want_blocks, resident_blocks2<K, PmlT, kDeep>());
    // chunk sizes as in fat_query.hip (launch_typed)
    const uint64_t lanes = (uint64_t)blocks * kQueryBlock;
    const uint64_t avg_len = std::max<uint64_t>(n_bases / std::max<uint64_t>(n_reads, 1), 1);
    uint32_t big = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(n_bases / lanes / 6 / avg_len, 1), 8);
    uint32_t tail_permille = 100;
    if (const char *e = getenv("COLBWT_LINE_ROWS_CHUNK")) {   // experiments: "<big>[,<tail permille>]"
        const int v = atoi(e);
        if (v >= 1 && v <= 1024) big = (uint32_t)v;
        if (const char *c = strchr(e, ',')) tail_permille = (uint32_t)std::min(1000, std::max(0, atoi(c + 1)));
    }
    hipLaunchKernelGGL((fat2_query_kernel<K, PmlT, kDeep>), dim3(blocks), dim3(kQueryBlock), 0, stream, T, d_bases, d_read_off, n_reads,
                       big, tail_permille, d_pml, d_cid);
}

template <int K>
void launch_steps2(const FatTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, uint64_t n_bases,
                   void *d_pml, int pml_bytes, uint8_t *d_cid, hipStream_t stream) {
    const bool deep = T.entry_shift == 7;
    if (pml_bytes == 2 && deep) launch_typed2<K, uint16_t, true>(T, d_bases, d_read_off, n_reads, n_bases, (uint16_t *)d_pml, d_cid, stream);
    else if (pml_bytes == 2) launch_typed2<K, uint16_t, false>(T, d_bases, d_read_off, n_reads, n_bases, (uint16_t *)d_pml, d_cid, stream);
    else if (deep) launch_typed2<K, uint32_t, true>(T, d_bases, d_read_off, n_reads, n_bases, (uint32_t *)d_pml, d_cid, stream);
    else launch_typed2<K, uint32_t, false>(T, d_bases, d_read_off, n_reads, n_bases, (uint32_t *)d_pml, d_cid, stream);
}

}  // namespace

void launch_fat2_query(const FatTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, uint64_t n_bases,
                       void *d_pml, int pml_bytes, uint8_t *d_cid, hipStream_t stream) {
    if (n_reads == 0) return;
#define X(K) \
    if (T.steps == K) launch_steps2<K>(T, d_bases, d_read_off, n_reads, n_bases, d_pml, pml_bytes, d_cid, stream);
    COLBWT_FAT_STEPS(X)
#undef X
}

#ifdef COLBWT_COUNT_TRIPS
extern "C" int colbwt_debug_fat2_stats(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_fat_stats), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fat_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

}  // namespace colbwt
