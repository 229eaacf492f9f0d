// fat_query.hip -- the PML / col-ID query over line rows (fat_layout.h).
//
// Same per-base semantics as query_kernels.hip / sk_query.hip (col_bwt.hpp:498-574,
// LF_table.hpp:251-298), one lane per read, one row load per loop trip.  What differs is the
// row and how it reaches the lane:
//
//   * the row is a whole 128-byte line.  The wave fetches its 64 rows with EIGHT instructions:
//     instruction q serves, in every group of eight lanes, the row wanted by the group's lane
//     q, each lane contributing 16 bytes (LDS-DMA, global_load_lds_dwordx4) -- eight distinct
//     lines per instruction, every byte of them used.  The texture addresser charges per
//     distinct line of an instruction, so 64 whole lines cost what 64 single 16-byte loads
//     cost (tools/gather_bench, modes 15 / 16);
//   * a trip consumes up to K <= 8 bases of a matching stretch (the look-ahead of sk_query.hip)
//     -- found with one 64-bit XOR of the next 8 read bases against the 8 characters the row
//     keeps, reported as one run -- and a mismatch costs no trip of its own: the row holds, for
//     the three most frequent other characters, where threshold_step (col_bwt.hpp:531-574) goes
//     and what the next step meets from there, so the trip consumes the mismatching base and
//     one more and lands where the reference is after those steps.
//
// Lanes are persistent and take chunk after chunk of consecutive reads (ReadCursor); all lanes
// of a wave stay in the loop until the wave's last lane is out of reads: a finished lane still
// fetches its share of the other lanes' rows.
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

template <int K, typename PmlT>
__global__ __launch_bounds__(kQueryBlock)
void fat_query_kernel(FatTable T, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ read_off,
                      uint64_t n_reads, uint32_t big_reads, uint32_t tail_permille,
                      PmlT *__restrict__ pml, uint8_t *__restrict__ cid) {
    constexpr bool kWide = sizeof(PmlT) == 4;
    __shared__ uint4 s_stage[kWaves][8][64];       // per wave: instruction q's 64 x 16 bytes
    __shared__ uint4 s_win[kWaves][4][64];         // read bytes (lane_io.h LaneWindow)
    __shared__ uint32_t s_claim;                   // the workgroup's chunk counter (ChunkPlan)
    // 48 KB (+ the counter) in all: three workgroups (12 waves) per CU -- the registers of the
    // collector allow no more.  The rows the lanes want are handed to their groups through the last
    // 256 bytes of the wave's own stage area, which the trip's last DMA (q = 7) fills only after
    // every lane has read them.  The chunk counter lives in LDS and is zeroed by the workgroup
    // itself: a launch owns no state outside its arguments, so any number of launches may overlap
    // on one index (other streams, other host threads).
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, g8 = lane & ~7u, p = lane & 7u;
    uint32_t *const s_jx = reinterpret_cast<uint32_t *>(&s_stage[wave][7][48]);   // 64 dwords
    ChunkPlan plan;
    plan.init(n_reads, big_reads, tail_permille);
    if (threadIdx.x == 0) s_claim = 0;
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
    uint32_t j = done ? 0u : T.r - 1;
    uint32_t o = kOffLastPos;
    uint32_t L = 0;
    // this lane's row, piece x: s_stage[wave][p][g8 + (x ^ p)]
    const uint4 *const my_row = &s_stage[wave][p][g8];

#ifdef COLBWT_COUNT_TRIPS
    unsigned long long stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = clock64();
#endif
    while (__any(!done)) {
        // ---- (1) where the lane stands.  Registers and LDS only: nothing goes to global memory
        // before the rows are requested (a store or load issued here would have to be waited for
        // at the LDS hand-over below, a memory round trip ahead of the one that matters).
        bool step_back = false;
        if (!done && rc.k == 0 && !rc.next_in_flight && (rc.r != rc.r_lo || rc.nc_ready)) {
            // the read is reported (or empty): a new query starts at the read before it
            // (col_bwt.hpp:503-508), or the chunk is finished -- the lane enters its next one once the
            // wave's flush has taken what it still holds (lane_out.h)
            if (rc.r != rc.r_lo) {
                rc.r -= 1;
                rc.k = rc.off - rc.next_off;
                rc.off = rc.next_off;
                step_back = rc.r > rc.r_lo;                  // the offset after this one is requested below
            } else if (kWide || acc.cnt == 0) {
                FAT_STAT(6);
                done = !rc.enter_chunk(plan, claim);
                if (!done) win.init(rc.off + rc.k - 1);
            }
            j = done ? 0u : T.r - 1;
            o = kOffLastPos;
            L = 0;
        }
        FAT_CLOCK(4);
        const bool live = !done && rc.k != 0;                // an empty read idles for one trip
        uint64_t &k = rc.k;
        const uint64_t g = rc.off + k - 1;                   // :512 pattern[m-i-1] is the next base

        // ---- (2) the wave's 64 rows, 128 bytes each, into LDS
        s_jx[lane] = j;
        wave_sync();
        {
            const uint4 ja = *reinterpret_cast<const uint4 *>(&s_jx[g8]);
            const uint4 jb = *reinterpret_cast<const uint4 *>(&s_jx[g8 + 4]);
            const uint32_t jq[8] = {ja.x, ja.y, ja.z, ja.w, jb.x, jb.y, jb.z, jb.w};
            wave_sync();                       // every lane has its eight rows: the hand-over area may be overwritten
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q)   // lane (g, p) brings piece p ^ q: the read-back below is bank-conflict free
                __builtin_amdgcn_global_load_lds(T.lines + (uint64_t)jq[q] * kFatRowBytes + ((p ^ q) << 4), &s_stage[wave][q][0],
                                                 16, 0, 0);
        }
        FAT_CLOCK(5);
        // ---- (3) the trip's other memory traffic, behind the rows: finished output groups, read
        // bytes for the lanes that run low (a trip looks at 8 bases), offsets for later boundaries.
        // All of it has landed at the one wait below.
        FAT_CLOCK(6);
        if (live && win.avail(g) < (k < 8u ? (uint32_t)k : 8u)) win.request(my_win, bases, g);
        if (step_back) { rc.in_next = read_off[rc.r - 1]; rc.next_in_flight = true; }
        if (!done && rc.fetch_pending) rc.request_chunk(plan, read_off);
        FAT_CLOCK(1);
        lds_dma_landed();
        rc.commit();
        FAT_CLOCK(2);

        const uint32_t have = live ? win.avail(g) : 0u;      // read bytes at hand (a fresh chunk starts with 1 .. 16)
        if (!live || have == 0) FAT_STAT(5);
        if (live && have != 0) {
            FAT_STAT(0);
            const uint4 r0 = my_row[0 ^ p];                  // CH, CID
            const uint4 r1 = my_row[1 ^ p];                  // len | flags << 16, cuts
            const uint64_t CH = (uint64_t)r0.x | ((uint64_t)r0.y << 32);
            const uint32_t len = r1.x & 0xFFFFu, flags = (r1.x >> 16) & 0xFFu;
            // ---- how the lane arrives
            uint32_t skip = 0;                               // 1: the row's own character was consumed on the way here
            bool stay = false;                               // this trip only moves on to another row
            if (o == kOffPred) { o = len - 1; skip = 1; FAT_STAT(7); }   // LF_table.hpp:282
            else if (o == kOffSucc) { o = 0; skip = 1; FAT_STAT(7); }    // LF_table.hpp:296
            else if (o >= len && j < T.r - 1) {
                // fast-forward of LF_table::LF (LF_table.hpp:256-259): one row per trip (the first
                // steps of most were taken at the jump, see the cuts)
                o -= len;
                j += 1;
                stay = true;
                FAT_STAT(1);
            } else {
                o = o < len ? o : len - 1;
            }
            if (!stay) {
                // ---- the next 8 read bases against the 8 characters the row's positions meet
                // (col_bwt.hpp:516, one iteration per LF step): byte 7 <-> the next base <-> the
                // row's own character.  After a threshold target (skip) the own character is past.
                const uint64_t W = win.get8(my_win, lane, g);
                const uint64_t X = skip ? ((W >> 8) ^ CH) & 0x00FFFFFFFFFFFFFFull : W ^ CH;
                uint32_t steps = matching_top_bytes(X);      // LF steps of the jump, the skipped one included
                const uint64_t left = k < have ? k : have;   // bases this trip may consume
                const uint32_t cap = left + skip < (uint64_t)K ? (uint32_t)left + skip : (uint32_t)K;
                steps = steps < cap ? steps : cap;
                uint32_t ids_lo = r0.z, ids_hi = r0.w;       // col ids of the steps, cid[a] at byte 8 - a
                uint32_t consumed, l_new;
                bool use_slot = false, own_jump = true;
                uint4 sl = make_uint4(0, 0, 0, 0);
                if (steps == 0) {                            // :520-523 the next base does not match (skip == 0)
                    const uint32_t c = (uint32_t)(W >> 56);
                    const uint32_t aidx = flags >> 4, cidx4 = fat_top_index(T.top4, c);
                    const uint32_t slot = cidx4 < aidx ? cidx4 : cidx4 - 1;
                    if (cidx4 < 4 && cidx4 != aidx && slot < kFatSlots && ((flags >> slot) & 1u)) {
                        // threshold_step decided when the index was built: the slot says where the
                        // reference is after this base and the next
                        sl = my_row[(kFatSlot0 / 16 + slot) ^ p];
                        const uint32_t tch2 = sl.w & 0xFFu, tcid2 = (sl.w >> 8) & 0xFFu;
                        const bool two = left >= 2 && ((uint32_t)(W >> 48) & 0xFFu) == tch2;   // :516 one step later
                        consumed = two ? 2u : 1u;
                        l_new = consumed - 1;                // :521 length = 0, then :517
                        ids_hi = (ids_hi & 0xFF000000u) | (tcid2 << 16);
                        use_slot = true;
                        own_jump = false;
                        FAT_STAT(2);
                    } else {
                        const uint32_t cidx = T.cmap[c];
                        bool moved = false;
                        if (cidx != kAbsent) moved = fat_threshold_scan(T, j, o, c, cidx);
                        FAT_STAT(moved ? 3 : 4);
                        if (moved) {                         // the target row is the next load
                            steps = 1;                       // one id to report: the row's own
                            consumed = 1;
                            l_new = 0;
                            own_jump = false;
                        } else {
                            // c occurs nowhere: (interval, offset) unchanged (:533-534), LF proceeds
                            // from this row; length restarts at 0
                            uint32_t st = matching_top_bytes(X & 0x00FFFFFFFFFFFFFFull);
                            const uint32_t cap0 = left < (uint64_t)K ? (uint32_t)left : (uint32_t)K;
                            steps = st < cap0 ? st : cap0;
                            consumed = steps;
                            l_new = steps - 1;
                        }
                    }
                } else {
                    consumed = steps - skip;
                    l_new = L + consumed;                    // :517 ++length per matching base
                }
                L = l_new;
                // ---- report the run (:525): element e (address g - consumed + 1 + e) <-> step steps - e
                {
                    const uint32_t top = use_slot ? consumed : steps;      // steps whose ids are reported
                    const uint32_t sh = 8u * (8u - top);                   // 0 .. 56 (top >= 1 when consumed >= 1)
                    const uint64_t ids = ((uint64_t)ids_lo | ((uint64_t)ids_hi << 32)) >> sh;
                    if constexpr (kWide) {
                        for (uint32_t e = 0; e < consumed; ++e) {
                            pml[g - consumed + 1 + e] = (PmlT)(l_new - e);
                            cid[g - consumed + 1 + e] = (uint8_t)(ids >> (8 * e));
                        }
                    } else {
                        acc.push_run(consumed, l_new, 0xFFFFFFFFu, 0xFFFFFFFFu, (uint32_t)ids, (uint32_t)(ids >> 32));
                    }
                }
                k -= consumed;
                if (k == 0) {
                    // the read is done: its last LF (:527) has no observable effect
                } else if (use_slot) {
                    j = consumed == 2 ? sl.y : sl.x;         // exact: one position, fast-forward included
                    o = consumed == 2 ? sl.z >> 16 : sl.z & 0xFFFFu;
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
        // ---- (5) the output groups the trip completed, all lanes' at once; the staged rows are read,
        // their LDS serves as the parking area
        wave_sync();
        if constexpr (!kWide) {
            if ((trip & (OutRuns::kPeriod - 1)) == OutRuns::kPeriod - 1)
                acc.flush_wave((uint16_t *)pml, cid, rc.off + rc.k, !done, rc.k == 0 && rc.r == rc.r_lo, &s_stage[wave][0][0], lane);
        }
        ++trip;
        wave_sync();   // the next trip overwrites s_jx and the staged rows
        FAT_CLOCK(3);
#ifdef COLBWT_COUNT_TRIPS
        ++clk[0];
#endif
    }
#ifdef COLBWT_COUNT_TRIPS
    for (int q = 0; q < 8; ++q) atomicAdd(&g_fat_stats[q], stat[q]);
    if (lane == 0)
        for (int q = 0; q < 8; ++q) atomicAdd(&g_fat_clocks[q], clk[q]);
#endif
}

// Blocks that are resident at once on the device (LDS-bound: 3 per CU): the persistent grid.
template <int K, typename PmlT>
uint32_t resident_blocks() {
    static uint32_t cached[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (cached[dev] == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fat_query_kernel<K, PmlT>, kQueryBlock, 0) != hipSuccess || per_cu < 1)
            per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 1;
        (void)hipGetLastError();
        cached[dev] = (uint32_t)per_cu * (uint32_t)cus;
    }
    return cached[dev];
}

template <int K, typename PmlT>
void launch_typed(const FatTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, uint64_t n_bases,
                  PmlT *d_pml, uint8_t *d_cid, hipStream_t stream) {
    const uint64_t want_blocks = (n_reads + kQueryBlock - 1) / kQueryBlock;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(want_blocks, resident_blocks<K, PmlT>());
    // Reads per bulk chunk: a chunk ends with a ragged flush of the collector, so it should hold a
    // few reads -- but no more than a sixth of a lane's share of the BASES, or a few lanes end up
    // with most of a workgroup's work (1 M reads of 10 kbp are five reads per lane: chunks of eight
    // took 1.6 times as long as single reads).  The last tenth of a workgroup's share (at least two
    // reads per lane) goes out read by read.
    const uint64_t lanes = (uint64_t)blocks * kQueryBlock;
    const uint64_t avg_len = std::max<uint64_t>(n_bases / std::max<uint64_t>(n_reads, 1), 1);
    uint32_t big = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(n_bases / lanes / 6 / avg_len, 1), 8);
    uint32_t tail_permille = 100;
    if (const char *e = getenv("COLBWT_LINE_ROWS_CHUNK")) {   // experiments: "<big>[,<tail permille>]"
        const int v = atoi(e);
        if (v >= 1 && v <= 1024) big = (uint32_t)v;
        if (const char *c = strchr(e, ',')) tail_permille = (uint32_t)std::min(1000, std::max(0, atoi(c + 1)));
    }
    hipLaunchKernelGGL((fat_query_kernel<K, PmlT>), dim3(blocks), dim3(kQueryBlock), 0, stream, T, d_bases, d_read_off, n_reads,
                       big, tail_permille, d_pml, d_cid);
}

template <int K>
void launch_steps(const FatTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, uint64_t n_bases,
                  void *d_pml, int pml_bytes, uint8_t *d_cid, hipStream_t stream) {
    if (pml_bytes == 2) launch_typed<K, uint16_t>(T, d_bases, d_read_off, n_reads, n_bases, (uint16_t *)d_pml, d_cid, stream);
    else launch_typed<K, uint32_t>(T, d_bases, d_read_off, n_reads, n_bases, (uint32_t *)d_pml, d_cid, stream);
}

}  // namespace

// d_order (the length-sorted lane assignment of the other layouts) is not used: persistent lanes
// claiming chunks of consecutive reads balance ragged batches by themselves.
void launch_fat_query(const FatTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, uint64_t n_bases,
                      void *d_pml, int pml_bytes, uint8_t *d_cid, const uint32_t *, hipStream_t stream) {
    if (n_reads == 0) return;
    if (T.slot_line0) {   // the variant with mismatch lines has a kernel of its own
        launch_fat2_query(T, d_bases, d_read_off, n_reads, n_bases, d_pml, pml_bytes, d_cid, stream);
        return;
    }
#define X(K) \
    if (T.steps == K) launch_steps<K>(T, d_bases, d_read_off, n_reads, n_bases, d_pml, pml_bytes, d_cid, stream);
    COLBWT_FAT_STEPS(X)
#undef X
}

#ifdef COLBWT_COUNT_TRIPS
extern "C" int colbwt_debug_fat_stats(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fat_stats), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out16 + 8, HIP_SYMBOL(g_fat_clocks), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fat_stats), z, sizeof(z)) != hipSuccess) return -1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fat_clocks), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

}  // namespace colbwt
