// sk3_query.hip -- the PML / col-ID query over THREE-STEP rows (sk_layout.h, K = 3) with the
// machinery of the line-row kernels: what indexes too large for line rows get (3e8 .. 1.3e9 rows
// of the file on 288 GB).
//
// Same rows and the same per-base semantics as sk_query.hip (col_bwt.hpp:498-574,
// LF_table.hpp:251-298; one row load per loop trip, up to three bases per trip).  What changed is
// everything around the row:
//   * the 32-byte row is fetched by lane PAIRS: instruction A brings the even lanes' rows, B the
//     odd lanes', each lane 16 bytes (LDS-DMA, no VGPR staging) -- 32 distinct lines per instruction
//     instead of 64, and the texture addresser charges per distinct line of an instruction
//     (tools/gather_bench mode 12: 48.0 against 39.4 G rows/s for two per-lane loads);
//   * lanes are persistent and claim chunks of consecutive reads (fat_cursor.h), so a wave does
//     not run for the slowest of its 64 reads and outputs of consecutive reads run on;
//   * read bytes come through the 64-byte LDS-DMA window, results leave through the two collectors
//     in whole 64-byte groups written by the wave together (lane_io.h) -- the round-1 kernel
//     stored 32-byte PML and 16-byte col-id pieces, the expensive kind (tools/scatter_bench).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "device_layout.h"
#include "fat_cursor.h"
#include "lane_io.h"
#include "lane_out.h"
#include "lf_device.h"
#include "query_kernels.h"
#include "sk_layout.h"

namespace colbwt {

namespace {

constexpr int K3 = 3;
// Lanes one pass of the wave's flush parks (lane_out.h).  16 (3.5 KB per wave) and a register cap of
// 128 give FOUR workgroups per CU: measured slower, 23.5 against 22.3 ms on the 1e9-row index
// (profiles/r03n_*) -- the kernel wants fewer passes per flush, not more waves.
constexpr uint32_t kFlushSlots = 32;
constexpr uint32_t kStagePieces = OutRuns::lds_q(kFlushSlots);   // per wave: 2 x 64 row pieces; then the flush's working area (7 KB)

// col_pml::threshold_step (col_bwt.hpp:531-574) for a mismatch the row cannot turn into "the
// target is d rows away": scans + (when the hint says so) the position compare.  Returns true
// when (j, o) moved: j is the target row, o the arrival code.
__device__ __forceinline__ bool sk3_threshold_scan(const SKTable &T, uint32_t &j, uint32_t &o, uint32_t c, uint32_t cidx, uint32_t hint) {
    SKRow<K3> t;
    if (hint == kHintPred) {
        const uint32_t q = sk_pred_char<K3>(T, j, c, cidx, t);      // :562
        if (q != kNone) { j = q; o = kOffPred; return true; }       // :565-569
        const uint32_t s = sk_succ_char<K3>(T, j, c, cidx, t);      // :548
        if (s != kNone) { j = s; o = kOffSucc; return true; }       // :552-557
        return false;
    }
    if (hint == kHintSucc) {
        const uint32_t s = sk_succ_char<K3>(T, j, c, cidx, t);
        if (s != kNone) { j = s; o = kOffSucc; return true; }
        return false;
    }
    const uint64_t pos = T.idx[j] + o;    // LF_table::to_idx (LF_table.hpp:214-217)
    uint64_t thr = T.n;                   // :535
    uint32_t nj = j, no = o;
    bool moved = false;
    const uint32_t s = sk_succ_char<K3>(T, j, c, cidx, t);          // :548
    if (s != kNone) { thr = T.thr[s]; nj = s; no = kOffSucc; moved = true; }   // :552-557
    if (pos < thr) {                                                // :560
        const uint32_t q = sk_pred_char<K3>(T, j, c, cidx, t);      // :562
        if (q != kNone) { nj = q; no = kOffPred; moved = true; }    // :565-569
    }
    j = nj;                                                         // :572-573
    o = no;
    return moved;
}

template <typename PmlT>
__global__ __launch_bounds__(kQueryBlock)
void sk3_query_kernel(SKTable T, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ read_off,
                      uint64_t n_reads, uint32_t big_reads, uint32_t tail_permille,
                      PmlT *__restrict__ pml, uint8_t *__restrict__ cid) {
    constexpr bool kWide = sizeof(PmlT) == 4;
    __shared__ uint4 s_stage[kWaves][kStagePieces];   // per wave: [q][lane] pieces of the trip's rows; then the flush's parking area
    __shared__ uint4 s_win[kWaves][4][64];            // read bytes (lane_io.h LaneWindow)
    __shared__ uint32_t s_jx[kWaves][64];             // the rows the lanes want, for their pair partners
    __shared__ uint8_t s_cmap[256];
    __shared__ uint32_t s_claim;                      // the workgroup's chunk counter (ChunkPlan)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, g2 = lane & ~1u, sub = lane & 1u;
    ChunkPlan plan;
    plan.init(n_reads, big_reads, tail_permille);
    if (threadIdx.x == 0) s_claim = 0;
    s_cmap[threadIdx.x] = T.cmap[threadIdx.x];
    __syncthreads();
    uint32_t *const claim = &s_claim;
    ReadCursor rc;
    bool done;
    rc.c_next = threadIdx.x;                          // the first chunk is the lane's own number
    rc.request_chunk(plan, read_off);
    rc.commit();
    done = !rc.enter_chunk(plan, claim);

    OutRuns acc;                                      // what the lane has reported and not yet stored (lane_out.h)
    uint32_t trip = 0;                                // the wave flushes every OutRuns::kPeriod-th trip
    LaneWindow win;
    win.init(rc.off + rc.k - 1);
    uint4 (*const my_win)[64] = s_win[wave];
    uint4 *const stage = s_stage[wave];

    // col_bwt.hpp:503-508: pos = n - 1, expressed as an arrival at the last row that clamps to len - 1
    uint32_t j = done ? 0u : T.r - 1;
    uint32_t o = kOffLastPos;
    uint32_t L = 0;

    while (__any(!done)) {
        // ---- (1) where the lane stands (registers and LDS only; fat_query.hip)
        bool step_back = false;
        if (!done && rc.k == 0 && !rc.next_in_flight && (rc.r != rc.r_lo || rc.nc_ready)) {
            if (rc.r != rc.r_lo) {
                rc.r -= 1;
                rc.k = rc.off - rc.next_off;
                rc.off = rc.next_off;
                step_back = rc.r > rc.r_lo;
            } else if (kWide || acc.cnt == 0) {              // the next chunk, once the wave's flush has taken what the lane holds
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

        // ---- (2) the wave's 64 rows, 32 bytes each, into LDS: instruction q serves, in every lane
        // pair, the row of the pair's lane q; lane (pair, s) brings piece s ^ q
        s_jx[wave][lane] = j;
        wave_sync();
        {
            const uint2 jp = *reinterpret_cast<const uint2 *>(&s_jx[wave][g2]);
            const uint32_t jq[2] = {jp.x, jp.y};
#pragma unroll
            for (uint32_t q = 0; q < 2; ++q)
                __builtin_amdgcn_global_load_lds(T.lines + (uint64_t)jq[q] * 32u + ((sub ^ q) << 4), &stage[q * 64], 16, 0, 0);
        }
        // ---- (3) the trip's other memory traffic, behind the rows
        if (live && win.avail(g) < (k < 8u ? (uint32_t)k : 8u)) win.request(my_win, bases, g);
        if (step_back) { rc.in_next = read_off[rc.r - 1]; rc.next_in_flight = true; }
        if (!done && rc.fetch_pending) rc.request_chunk(plan, read_off);
        lds_dma_landed();
        rc.commit();

        const uint32_t have = live ? win.avail(g) : 0u;      // read bytes at hand
        uint32_t consumed = 0, l_new = 0;
        uint64_t ids = 0;
        if (live && have != 0) {
            // this lane's row: piece x was brought by the pair's lane x ^ s into stage[s][g2 + (x ^ s)]
            SKRow<K3> w;
            {
                const uint4 a = stage[sub * 64 + g2 + (0u ^ sub)], b = stage[sub * 64 + g2 + (1u ^ sub)];
                w.d[0] = a.x; w.d[1] = a.y; w.d[2] = a.z; w.d[3] = a.w; w.d[4] = b.x; w.d[5] = b.y; w.d[6] = b.z; w.d[7] = b.w;
            }
            const uint64_t W = win.get8(my_win, lane, g);    // byte 7 = the next base
            const uint64_t left = k < have ? k : have;       // bases this trip may consume (>= 1)
            const uint32_t len = sk_len<K3>(w);
            bool stay = false;                               // j already names the next row to load
            uint32_t own = 0;                                // 1: the row's own character takes the next base in this trip
            if (o == kOffPred) o = len - 1;                  // LF_table.hpp:282: the row's own character went with the mismatching base
            else if (o == kOffSucc) o = 0;                   // LF_table.hpp:296
            else if (o >= len && j < T.r - 1) {
                o -= len;                                    // fast-forward of LF_table::LF (LF_table.hpp:256-259), one row per trip
                j += 1;
                stay = true;
            } else {
                o = o < len ? o : len - 1;
                own = 1;
            }
            uint32_t la_from = 1;                            // read byte (7 - la_from ..) meets the row's second character
            if (own) {
                const uint32_t c = (uint32_t)(W >> 56);      // raw byte
                ids = sk_cid<K3>(w);                         // :513 before any re-orientation
                consumed = 1;
                if (sk_char<K3>(w) == c) {                   // :516
                    l_new = L + 1;                           // :517
                } else {                                     // :520-523
                    l_new = 0;
                    const uint32_t cidx = s_cmap[c];
                    if (cidx != kAbsent) {                   // else (interval, offset) unchanged (:533-534)
                        uint32_t hint = kHintCompare, dist = kSKDistFar;
                        const uint32_t hs = hint_slot(cidx, s_cmap[sk_char<K3>(w)]);
                        if (hs < kHintSlots) {
                            hint = (sk_hints<K3>(w) >> (2 * hs)) & 3u;
                            dist = sk_dist<K3>(w, hs);
                        }
                        if (dist != kSKDistFar && hint != kHintCompare) {   // decided and close: the target is the next load
                            j = hint == kHintPred ? j - dist : j + dist;
                            o = hint == kHintPred ? kOffPred : kOffSucc;
                            stay = true;
                        } else if (sk3_threshold_scan(T, j, o, c, cidx, hint)) {
                            stay = true;
                        }
                    }
                }
            } else {
                la_from = 0;                                 // arrival at a threshold target: the next base meets the second character
                l_new = L;
            }
            if (!stay) {
                // After a - 1 LF steps (:527) every position of this row is in one original row whose
                // character / col id are char_a / cid_a: while the next base matches it, that base is
                // ++length with that col id (:513-517) and the jump grows by one LF step.
                uint32_t steps = 1;
                const uint32_t c2 = (uint32_t)(W >> (8u * (7u - la_from))) & 0xFFu, c3 = (uint32_t)(W >> (8u * (6u - la_from))) & 0xFFu;
                if (left > consumed && c2 == sk_char_at<K3, 2>(w)) {
                    ids = (ids << 8) | sk_cid_at<K3, 2>(w);
                    ++consumed;
                    ++l_new;
                    steps = 2;
                    if (left > consumed && c3 == sk_char_at<K3, 3>(w)) {
                        ids = (ids << 8) | sk_cid_at<K3, 3>(w);
                        ++consumed;
                        ++l_new;
                        steps = 3;
                    }
                }
                if (k != consumed) {
                    j = sk_I<K3>(w, steps);                  // LF^steps lands at (I_s, O_s + o) ...
                    const uint32_t cut = sk_cut_a<K3>(w);
                    if (steps == (uint32_t)K3 && cut != kSKCutNone && o >= cut) {
                        j += 1;                              // ... which is already in the next row
                        o -= cut;
                        const uint32_t lb = sk_len_b<K3>(w);
                        if (lb != kSKCutNone && o >= lb) {   // ... or in the one after
                            j += 1;
                            o -= lb;
                        }
                    } else {
                        o += sk_O<K3>(w, steps);
                    }
                }
            }
            L = l_new;
            // ---- report the run (:525): element e is the base at g - consumed + 1 + e, the newest first
            if constexpr (kWide) {
                for (uint32_t e = 0; e < consumed; ++e) {
                    pml[g - consumed + 1 + e] = (PmlT)(l_new - e);
                    cid[g - consumed + 1 + e] = (uint8_t)(ids >> (8 * e));
                }
            } else {
                acc.push_run(consumed, l_new, 0xFFFFFFFFu, 0xFFFFFFFFu, (uint32_t)ids, 0u);
            }
            k -= consumed;
        }
        // ---- (5) the output groups the trip completed, all lanes' at once
        wave_sync();
        if constexpr (!kWide) {
            if ((trip & (OutRuns::kPeriod - 1)) == OutRuns::kPeriod - 1)
                acc.template flush_wave<kFlushSlots>((uint16_t *)pml, cid, rc.off + rc.k, !done, rc.k == 0 && rc.r == rc.r_lo, stage, lane);
        }
        ++trip;
        wave_sync();   // the next trip overwrites s_jx and the staged rows
    }
}

template <typename PmlT>
uint32_t resident_blocks3() {
    static uint32_t cached[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (cached[dev] == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sk3_query_kernel<PmlT>, kQueryBlock, 0) != hipSuccess || per_cu < 1)
            per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 1;
        (void)hipGetLastError();
        cached[dev] = (uint32_t)per_cu * (uint32_t)cus;
    }
    return cached[dev];
}

template <typename PmlT>
void launch_typed3(const SKTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, uint64_t n_bases,
                   PmlT *d_pml, uint8_t *d_cid, hipStream_t stream) {
    const uint64_t want_blocks = (n_reads + kQueryBlock - 1) / kQueryBlock;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(want_blocks, resident_blocks3<PmlT>());
    const uint64_t lanes = (uint64_t)blocks * kQueryBlock;     // chunk sizes as in fat_query.hip (launch_typed)
    const uint64_t avg_len = std::max<uint64_t>(n_bases / std::max<uint64_t>(n_reads, 1), 1);
    const uint32_t big = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(n_bases / lanes / 6 / avg_len, 1), 8);
    hipLaunchKernelGGL((sk3_query_kernel<PmlT>), dim3(blocks), dim3(kQueryBlock), 0, stream, T, d_bases, d_read_off, n_reads, big,
                       100u, d_pml, d_cid);
}

}  // namespace

// The three-step query with persistent lanes and pair-fetched rows.  n_bases = read_off[n_reads] - read_off[0].
void launch_sk3_query(const SKTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, uint64_t n_bases,
                      void *d_pml, int pml_bytes, uint8_t *d_cid, hipStream_t stream) {
    if (n_reads == 0) return;
    if (pml_bytes == 2) launch_typed3<uint16_t>(T, d_bases, d_read_off, n_reads, n_bases, (uint16_t *)d_pml, d_cid, stream);
    else launch_typed3<uint32_t>(T, d_bases, d_read_off, n_reads, n_bases, (uint32_t *)d_pml, d_cid, stream);
}

}  // namespace colbwt
