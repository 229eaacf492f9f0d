// query_kernels.hip -- the PML / col-ID query on gfx950 (MI355X), wave64.
//
// One lane per read; each lane walks its read right to left and carries the
// reference's whole per-base state (interval, offset, PML length) in
// registers.  Reference semantics, step for step:
//   col_pml::_query_pml     col_bwt.hpp:498-529   (extend / reset, col id)
//   col_pml::threshold_step col_bwt.hpp:531-574   (mismatch re-orientation)
//   LF_table::succ_char / pred_char  LF_table.hpp:271-298
//   LF_table::LF / LF_idx   LF_table.hpp:251-268  (LF step + fast-forward)
//
// Memory behaviour (HBM-bound integer pointer chasing, no MFMA):
//   * the only random access of a step is the 16-byte row the LF step lands
//     on (device_layout.h); that one load also carries char, col id, the next
//     interval/offset, idx (for pos) and the run length, so the following
//     step needs no further table access unless it fast-forwards / mismatches;
//   * HBM serves every random 16-byte row load as one 128-byte line fill
//     (calibrated with tools/gather_bench: two loads in the two 64-byte halves
//     of a line cost ONE TCC_EA0_RDREQ), so what matters is the number of
//     distinct lines a step touches: rows are 16-byte aligned (8 per line), the
//     threshold decision is pre-resolved into 2-bit hints inside the row, and
//     a mismatch runs only the scan whose result it will use;
//   * read bytes: each lane stages 64 bytes of its read in LDS per refill and picks one
//     byte per step from there; the window slides and is refilled for the whole wave at
//     once (lane_io.h SlidingWindow);
//   * PML (u16) and col id (u8) are collected for 16 bases in registers and
//     leave as 32-byte and 16-byte aligned stores (whole 32-byte sectors).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"
#include "lane_io.h"
#include "lf_device.h"
#include "query_kernels.h"

namespace colbwt {

// col_pml::threshold_step (col_bwt.hpp:531-574).  Reference order: find succ,
// take its threshold, then pred only when pos < thr (the !MULTI_THREAD arm;
// the threaded arm computes the same values).  The per-row hint tells in
// advance how `pos < thr` comes out, so only the scan that decides the result
// runs; kHintCompare falls back to the reference's exact sequence.
__device__ __forceinline__ void threshold_step(const DevTable &T, const uint8_t *s_cmap,
                                               uint32_t &i, uint64_t &o, uint4 &w, uint32_t c) {
    const uint32_t cidx = s_cmap[c];
    if (cidx == kAbsent) return;  // c occurs nowhere: (interval, offset) unchanged (:533-534)
    uint32_t hint = kHintCompare;
    const uint32_t slot = hint_slot(cidx, s_cmap[row_char(w)]);
    if (slot < kHintSlots) {
        hint = (row_hints(w) >> (2 * slot)) & 3u;   // rarer characters: compare
        const uint32_t dist = row_dist(w, slot);
        if (dist != kDistFar && hint != kHintCompare) {
            // decided and close: the target row's distance is in the row; one load, no scan
            if (hint == kHintPred) {                         // :565-569
                i -= dist;
                w = T.rows[i];
                o = row_len(T, i, w) - 1;                    // LF_table.hpp:282
            } else {                                         // :552-557
                i += dist;
                w = T.rows[i];
                o = 0;
            }
            return;
        }
    }
    uint4 t;
    if (hint == kHintPred) {
        // pos < thr for every offset of this row (or no successor, thr = n :535): pred wins if it exists
        const uint32_t q = pred_char(T, i, c, cidx, t);      // :562
        if (q != kNone) {                                     // :565-569
            i = q;
            o = row_len(T, q, t) - 1;                         // LF_table.hpp:282
            w = t;
            return;
        }
        const uint32_t s = succ_char(T, i, c, cidx, t);      // :548
        if (s != kNone) { i = s; o = 0; w = t; }              // :552-557
        return;
    }
    if (hint == kHintSucc) {
        // pos >= thr(succ) for every offset of this row: the successor exists and wins (:552-557)
        const uint32_t s = succ_char(T, i, c, cidx, t);
        if (s != kNone) { i = s; o = 0; w = t; }
        return;
    }
    const uint64_t pos = row_idx(T, i) + o;  // LF_table::to_idx (LF_table.hpp:214-217)
    uint64_t thr = T.n;                   // :535
    uint32_t ni = i;
    uint64_t no = o;
    uint4 nw = w;
    const uint32_t s = succ_char(T, i, c, cidx, t);  // :548
    if (s != kNone) {                                 // :552-557
        thr = T.thr[s];
        ni = s;
        no = 0;
        nw = t;
    }
    if (pos < thr) {                                  // :560
        const uint32_t q = pred_char(T, i, c, cidx, t);  // :562
        if (q != kNone) {                             // :565-569
            ni = q;
            no = row_len(T, q, t) - 1;                // LF_table.hpp:282
            nw = t;
        }
    }
    i = ni;                                           // :572-573
    o = no;
    w = nw;
}

// Register budget: <= 64 VGPRs and <= 80 SGPRs keep 8 waves per SIMD resident
// (MI355X_MICROARCH.md: residency of 256-thread blocks by .sgpr_count).
template <typename PmlT>
__global__ __launch_bounds__(kQueryBlock) __attribute__((amdgpu_num_sgpr(80), amdgpu_num_vgpr(64)))
void pml_query_kernel(DevTable T, const uint8_t *__restrict__ bases,
                                                                const uint64_t *__restrict__ read_off,
                                                                uint64_t n_reads, PmlT *__restrict__ pml,
                                                                uint8_t *__restrict__ cid,
                                                                const uint32_t *__restrict__ order) {
    constexpr bool kWide = sizeof(PmlT) == 4;
    __shared__ uint32_t s_rd[16][kQueryBlock];
    __shared__ uint8_t s_cmap[256];
    for (uint32_t t = threadIdx.x; t < 256; t += kQueryBlock) s_cmap[t] = T.cmap[t];
    __syncthreads();

    const uint64_t slot = (uint64_t)blockIdx.x * kQueryBlock + threadIdx.x;
    if (slot >= n_reads) return;
    // ragged batches: `order` lists the reads by decreasing length, so the 64 lanes of a
    // wave walk reads of similar length and the longest reads start first
    const uint64_t rd = order ? order[slot] : slot;
    const uint64_t off = read_off[rd];
    const uint64_t m = read_off[rd + 1] - off;
    if (m == 0) return;

    // col_bwt.hpp:503-508: pos = n-1, interval = r-1, offset = len(r-1)-1, length = 0
    uint32_t i = T.r - 1;
    uint4 w = T.rows[i];
    uint64_t o = row_len(T, i, w) - 1;
    uint32_t L = 0;

    SlidingWindow win;          // refilled for the whole wave when any lane runs out (lane_io.h)
    OutAcc<PmlT> acc;
    win.init(off + m - 1);

    for (uint64_t k = m; k-- > 0;) {                 // col_bwt.hpp:510, i = m-1-k
        const uint64_t g = off + k;
        if (__any(win.avail(g) < 1)) win.refill(s_rd, bases, g);
        const uint32_t c = win.get(s_rd, g);         // :512 pattern[m-i-1], raw byte
        const uint32_t col_id = row_cid(w);          // :513 before any re-orientation
        if (row_char(w) == c) {                      // :516
            ++L;                                     // :517
        } else {
            L = 0;                                   // :521
            threshold_step(T, s_cmap, i, o, w, c);   // :522
        }
        if (kWide) {                                 // :525
            pml[g] = (PmlT)L;
            cid[g] = (uint8_t)col_id;
        } else {
            acc.push(L, col_id);
            if ((g & (kFlush - 1)) == 0) acc.flush(pml, cid, g);
        }
        if (k == 0) break;  // the reference's last LF (col_bwt.hpp:527) has no observable effect

        // LF_table::LF (LF_table.hpp:251-262).  The row knows where its image leaves the landing
        // row and the row after it (cut_a, len_b), so the first steps of the fast-forward
        // (:256-259) are taken before the load; the loop only runs for what is left.
        uint32_t j = row_interval(w);                // :253
        uint64_t t = o;
        const uint32_t cut = row_cut_a(w);
        if (cut != kCutNone && t >= cut) {
            j += 1;
            t -= cut;
            const uint32_t lb = row_len_b(w);
            if (lb != kCutNone && t >= lb) {
                j += 1;
                t -= lb;
            }
        } else {
            t += row_offset(w);                      // :254
        }
        w = T.rows[j];
        for (;;) {
            const uint64_t len = row_len(T, j, w);
            if (t < len || j >= T.r - 1) break;      // :256 (bounded: a validated table never passes r-1)
            t -= len;                                // :258
            ++j;
            w = T.rows[j];
        }
        i = j;
        o = t;
    }
    if constexpr (!kWide) {
        if (acc.cnt) acc.flush(pml, cid, off);       // the group at the start of the read
    }
}

void launch_pml_query(const DevTable &T, const uint8_t *d_bases, const uint64_t *d_read_off,
                      uint64_t n_reads, void *d_pml, int pml_bytes, uint8_t *d_cid, const uint32_t *d_order,
                      hipStream_t stream) {
    if (n_reads == 0) return;
    const uint64_t blocks = (n_reads + kQueryBlock - 1) / kQueryBlock;
    dim3 grid((uint32_t)blocks), block(kQueryBlock);
    if (pml_bytes == 2)
        hipLaunchKernelGGL(pml_query_kernel<uint16_t>, grid, block, 0, stream, T, d_bases, d_read_off, n_reads,
                           (uint16_t *)d_pml, d_cid, d_order);
    else
        hipLaunchKernelGGL(pml_query_kernel<uint32_t>, grid, block, 0, stream, T, d_bases, d_read_off, n_reads,
                           (uint32_t *)d_pml, d_cid, d_order);
}

}  // namespace colbwt
