// fat_cursor.h -- what the two line-row query kernels (fat_query.hip: in-row mismatch slots,
// fat2_query.hip: mismatch lines) share: arrival codes, the run-time threshold_step for characters
// without a slot, and the persistent lanes' chunk plan / read cursor.  Everything lives in an
// unnamed namespace: each translation unit gets its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"
#include "fat_layout.h"
#include "lane_io.h"
#include "query_kernels.h"

namespace colbwt {

namespace {

constexpr uint32_t kOffLastPos = 0xFFFFFFFFu;   // LF-style arrival, offset clamped to the row's last position
constexpr uint32_t kOffPred = 0xFFFFFFFEu;      // threshold target reached from below: offset = len - 1
constexpr uint32_t kOffSucc = 0xFFFFFFFDu;      // threshold target reached from above: offset = 0
constexpr uint32_t kWaves = kQueryBlock / 64;

// col_pml::threshold_step (col_bwt.hpp:531-574) at run time, for characters without a mismatch
// slot in the row (beyond the four most frequent, or a threshold inside the row): scans over
// the byte-per-row character array, position compare.  Returns true when (j, o) moved: j is
// the target row, o the arrival code.
__device__ __forceinline__ bool fat_threshold_scan(const FatTable &T, uint32_t &j, uint32_t &o, uint32_t c, uint32_t cidx) {
    const uint64_t pos = T.idx[j] + o;                     // LF_table::to_idx (LF_table.hpp:214-217)
    uint64_t thr = T.n;                                    // :535
    uint32_t nj = j, no = o;
    bool moved = false;
    const uint32_t s = fat_succ_char(T, j, c, cidx);      // :548
    if (s != kNone) { thr = T.thr[s]; nj = s; no = kOffSucc; moved = true; }   // :552-557
    if (pos < thr) {                                       // :560
        const uint32_t q = fat_pred_char(T, j, c, cidx);  // :562
        if (q != kNone) { nj = q; no = kOffPred; moved = true; }               // :565-569
    }
    j = nj;                                                // :572-573
    o = no;
    return moved;
}

#ifdef COLBWT_COUNT_TRIPS
// Experiment builds only (make variant VFLAGS=-DCOLBWT_COUNT_TRIPS): what the lanes' trips were spent on.
__device__ unsigned long long g_fat_stats[8];   // live trips, fast-forward, slot, scan, absent, idle (done / empty), chunk ends, skip arrivals
__device__ unsigned long long g_fat_clocks[8];  // per wave (lane 0): trips, cycles top -> loads issued, -> landed, -> trip end
#define FAT_STAT(k) (++stat[k])
#define FAT_CLOCK(k) do { const unsigned long long t_ = clock64(); clk[k] += t_ - t_prev; t_prev = t_; } while (0)
#else
#define FAT_STAT(k) ((void)0)
#define FAT_CLOCK(k) ((void)0)
#endif

// 64-bit helpers the trip is written with
__device__ __forceinline__ uint32_t matching_top_bytes(uint64_t x) {   // bytes 7, 6, .. that are zero
    return x ? (uint32_t)__builtin_clzll(x) >> 3 : 8u;
}

// The reads a lane walks.  Lanes are persistent: the grid just fills the chip and every lane
// takes chunk after chunk of consecutive reads until none is left, so neither a wave (which with
// one read per lane runs for the maximum of its 64 reads' trips, about 1.6 times their mean on
// 150 bp reads) nor the launch (which ends with its slowest lane) is held up by slow reads.
// Chunks are CLAIMED: every workgroup owns an equal share of the batch (a share is thousands of
// reads: shares differ by a fraction of a percent in work), and inside it a lane's first chunk is
// its own number and every further one comes from the workgroup's counter in LDS -- a counter in HBM shared by
// the whole grid serialises (150 M claims/s measured: the 10 M single-read claims of a C2 batch
// took longer than the query).  The claim is made one chunk ahead of need, so the offsets of the
// claimed chunk are in registers when the lane gets there.  The bulk of a share goes out in
// chunks of `big` reads, its tail in single reads, so a workgroup ends within about one read's
// time of the moment its counter runs out.  Inside a chunk the reads are taken from the last to
// the first: their bases and outputs are contiguous, so the read window and the output collector
// simply run on across read boundaries.
struct ChunkPlan {
    uint64_t read_lo;                    // first read of the workgroup's share
    uint64_t n_big, n_chunks;            // chunks [0, n_big) hold `big` reads each, the rest one read each
    uint32_t big;
    __device__ __forceinline__ void init(uint64_t n_reads, uint32_t big_reads, uint32_t tail_permille) {
        read_lo = n_reads / gridDim.x * blockIdx.x + (n_reads % gridDim.x < blockIdx.x ? n_reads % gridDim.x : blockIdx.x);
        const uint64_t n = n_reads / gridDim.x + (blockIdx.x < n_reads % gridDim.x ? 1 : 0);
        uint64_t tail = n * tail_permille / 1000;
        tail = tail < 2 * kQueryBlock ? 2 * kQueryBlock : tail;
        tail = tail < n ? tail : n;
        big = big_reads;
        n_big = (n - tail) / big;
        n_chunks = n_big + (n - n_big * big);
    }
    __device__ __forceinline__ uint64_t first_read(uint64_t c) const {
        return read_lo + (c < n_big ? c * big : n_big * big + (c - n_big));
    }
    __device__ __forceinline__ uint64_t last_read(uint64_t c) const {
        return read_lo + (c < n_big ? (c + 1) * big - 1 : n_big * big + (c - n_big));
    }
};

struct ReadCursor {
    uint64_t off = 0;        // read_off[r]: global index of the current read's first base
    uint64_t k = 0;          // bases of the current read not yet reported
    uint64_t next_off = 0;   // read_off[r - 1] (valid while r > r_lo)
    uint64_t r = 0, r_lo = 0;
    uint64_t c_next = 0;     // the chunk claimed for later
    uint64_t nc_end = 0, nc_off = 0, nc_next = 0;   // of that chunk: read_off[r_hi + 1], [r_hi], [r_hi - 1]
    // Offsets on their way from HBM.  They are requested at the top of a trip, before the trip's
    // rows, and moved into the fields above (commit) after the trip's one wait, when they have
    // landed with the rows: the compiler copies a loaded value out of its destination register as
    // soon as control flow merges, so consuming them any earlier puts a full memory round trip
    // into nearly every trip (some lane of a wave crosses a read boundary in most trips).
    uint64_t in_next = 0, in_end = 0, in_off = 0, in_nx = 0;
    bool fetch_pending = false;   // c_next is claimed, its offsets not yet requested
    bool next_in_flight = false, chunk_in_flight = false;
    bool nc_ready = false;        // nc_* describe chunk c_next

    __device__ __forceinline__ void request_chunk(const ChunkPlan &P, const uint64_t *__restrict__ read_off) {
        fetch_pending = false;
        if (c_next < P.n_chunks) {
            const uint64_t hi = P.last_read(c_next);
            in_end = read_off[hi + 1];
            in_off = read_off[hi];
            in_nx = read_off[hi > P.first_read(c_next) ? hi - 1 : hi];
            chunk_in_flight = true;
        } else {
            nc_ready = true;      // nothing to fetch: enter_chunk will see the end of the share
        }
    }
    __device__ __forceinline__ void commit() {   // after the trip's wait: everything requested has landed
        if (next_in_flight) { next_off = in_next; next_in_flight = false; }
        if (chunk_in_flight) { nc_end = in_end; nc_off = in_off; nc_next = in_nx; chunk_in_flight = false; nc_ready = true; }
    }
    // enters the claimed chunk and claims the one after it; false when the share is used up
    __device__ __forceinline__ bool enter_chunk(const ChunkPlan &P, uint32_t *claim) {
        if (c_next >= P.n_chunks) return false;
        r_lo = P.first_read(c_next);
        r = P.last_read(c_next);
        off = nc_off;
        k = nc_end - nc_off;
        next_off = nc_next;
        c_next = kQueryBlock + atomicAdd(claim, 1u);    // chunks below kQueryBlock are the lanes' first ones
        fetch_pending = true;                      // the offsets are requested at the top of the next trip
        nc_ready = false;
        return true;
    }
};

}  // namespace

}  // namespace colbwt
