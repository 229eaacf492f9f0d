// s2_query.hip -- the PML / col-ID query over the two-step layout (s2_layout.h).
//
// Same per-base semantics as query_kernels.hip (col_bwt.hpp:498-574,
// LF_table.hpp:251-298), one lane per read.  The difference is the LF jump: a
// refined row knows the character/col id of the row all its positions map into,
// so when the NEXT read base matches that character the lane emits both bases
// and takes LF o LF with a single row load (one 128-byte line fill for two
// bases); otherwise it takes the ordinary LF and the mismatch is handled at
// the landing row in the next iteration, exactly as the reference would.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"
#include "lane_io.h"
#include "lf_device.h"
#include "query_kernels.h"
#include "s2_layout.h"

namespace colbwt {

// col_pml::threshold_step (col_bwt.hpp:531-574) over refined rows; see
// query_kernels.hip for the hint logic.
__device__ __forceinline__ void s2_threshold_step(const S2Table &T, const uint8_t *s_cmap, uint32_t &i, uint32_t &o,
                                                  S2Row &w, uint32_t c) {
    const uint32_t cidx = s_cmap[c];
    if (cidx == kAbsent) return;  // c occurs nowhere: (interval, offset) unchanged (:533-534)
    uint32_t hint = kHintCompare;
    const uint32_t slot = hint_slot(cidx, s_cmap[s2_char(w)]);
    if (slot < kHintSlots) {
        hint = (s2_hints(w) >> (2 * slot)) & 3u;
        const uint32_t dist = s2_dist(w, slot);
        if (dist != kDistFar && hint != kHintCompare) {   // decided and close: one load, no scan
            if (hint == kHintPred) {                      // :565-569
                i -= dist;
                w = s2_load(T, i);
                o = s2_len(w) - 1;                        // LF_table.hpp:282
            } else {                                      // :552-557
                i += dist;
                w = s2_load(T, i);
                o = 0;
            }
            return;
        }
    }
    S2Row t;
    if (hint == kHintPred) {
        const uint32_t q = s2_pred_char(T, i, c, cidx, t);      // :562
        if (q != kNone) { i = q; o = s2_len(t) - 1; w = t; return; }   // :565-569, LF_table.hpp:282
        const uint32_t s = s2_succ_char(T, i, c, cidx, t);      // :548
        if (s != kNone) { i = s; o = 0; w = t; }                 // :552-557
        return;
    }
    if (hint == kHintSucc) {
        const uint32_t s = s2_succ_char(T, i, c, cidx, t);
        if (s != kNone) { i = s; o = 0; w = t; }
        return;
    }
    const uint64_t pos = T.idx[i] + o;    // LF_table::to_idx (LF_table.hpp:214-217)
    uint64_t thr = T.n;                   // :535
    uint32_t ni = i, no = o;
    S2Row nw = w;
    const uint32_t s = s2_succ_char(T, i, c, cidx, t);  // :548
    if (s != kNone) { thr = T.thr[s]; ni = s; no = 0; nw = t; }   // :552-557
    if (pos < thr) {                                     // :560
        const uint32_t q = s2_pred_char(T, i, c, cidx, t);  // :562
        if (q != kNone) { ni = q; no = s2_len(t) - 1; nw = t; }   // :565-569
    }
    i = ni; o = no; w = nw;                              // :572-573
}

namespace {

template <typename PmlT>
__global__ __launch_bounds__(kQueryBlock) __attribute__((amdgpu_num_sgpr(80), amdgpu_num_vgpr(64)))
void s2_query_kernel(S2Table T, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ read_off,
                     uint64_t n_reads, PmlT *__restrict__ pml, uint8_t *__restrict__ cid,
                     const uint32_t *__restrict__ order) {
    constexpr bool kWide = sizeof(PmlT) == 4;
    __shared__ uint32_t s_rd[16][kQueryBlock];
    __shared__ uint8_t s_cmap[256];
    for (uint32_t t = threadIdx.x; t < 256; t += kQueryBlock) s_cmap[t] = T.cmap[t];
    __syncthreads();

    const uint64_t slot = (uint64_t)blockIdx.x * kQueryBlock + threadIdx.x;
    if (slot >= n_reads) return;
    const uint64_t rd = order ? order[slot] : slot;
    const uint64_t off = read_off[rd];
    const uint64_t m = read_off[rd + 1] - off;
    if (m == 0) return;

    // col_bwt.hpp:503-508: pos = n-1 = the last position of the last (refined) row
    uint32_t i = T.r2 - 1;
    S2Row w = s2_load(T, i);
    uint32_t o = s2_len(w) - 1;
    uint32_t L = 0;
    OutAcc<PmlT> acc;
    ReadWindow win;
    win.refill(s_rd, bases, off + m - 1);

    auto emit = [&](uint64_t g, uint32_t len, uint32_t col_id, bool last) {   // :525
        if constexpr (kWide) {
            pml[g] = (PmlT)len;
            cid[g] = (uint8_t)col_id;
        } else {
            acc.push(len, col_id);
            if ((g & (kFlush - 1)) == 0 || last) acc.flush(pml, cid, g);
        }
    };

    for (uint64_t k = m; k > 0;) {
        const uint64_t g = off + k - 1;
        const uint32_t c = win.get(s_rd, g);              // :512 pattern[m-i-1], raw byte
        const uint32_t col_id = s2_cid(w);               // :513 before any re-orientation
        if (s2_char(w) == c) {                           // :516
            ++L;
        } else {
            L = 0;                                       // :521
            s2_threshold_step(T, s_cmap, i, o, w, c);    // :522
        }
        --k;
        emit(g, L, col_id, k == 0);
        if (k == 0) break;                               // the last LF (:527) has no observable effect
        if ((g & 63) == 0) win.refill(s_rd, bases, g - 1);

        // One LF step (:527) lands every position of this refined row in the same original
        // row, whose character / col id are char2 / cid2.  If the next base matches it,
        // the next iteration would be ++length with that col id (:513-517) followed by
        // another LF: emit it here and jump LF o LF in one go.
        uint32_t j, t;
        const uint32_t c2 = win.get(s_rd, g - 1);
        if (c2 == s2_char2(w)) {
            ++L;
            --k;
            emit(g - 1, L, s2_cid2(w), k == 0);
            if (k == 0) break;
            if (((g - 1) & 63) == 0) win.refill(s_rd, bases, g - 2);
            j = s2_i2(w);
            t = s2_o2(w) + o;
        } else {
            j = s2_i1(w);
            t = s2_o1(w) + o;
        }
        // LF_table::LF fast-forward (LF_table.hpp:256-259) over refined rows, up to three rows
        // per memory round trip (a row carries the lengths of the two rows after it)
        w = s2_load(T, j);
        for (;;) {
            const uint32_t len = s2_len(w);
            if (t < len || j >= T.r2 - 1) break;
            t -= len;
            uint32_t hop = 1;
            const uint32_t l1 = s2_len8_next1(w);
            if (l1 != kLen8Long && t >= l1 && j + 1 < T.r2 - 1) {
                t -= l1;
                hop = 2;
                const uint32_t l2 = s2_len8_next2(w);
                if (l2 != kLen8Long && t >= l2 && j + 2 < T.r2 - 1) {
                    t -= l2;
                    hop = 3;
                }
            }
            j += hop;
            w = s2_load(T, j);
        }
        i = j;
        o = t;
    }
}

}  // namespace

void launch_s2_query(const S2Table &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads,
                     void *d_pml, int pml_bytes, uint8_t *d_cid, const uint32_t *d_order, hipStream_t stream) {
    if (n_reads == 0) return;
    const uint64_t blocks = (n_reads + kQueryBlock - 1) / kQueryBlock;
    dim3 grid((uint32_t)blocks), block(kQueryBlock);
    if (pml_bytes == 2)
        hipLaunchKernelGGL(s2_query_kernel<uint16_t>, grid, block, 0, stream, T, d_bases, d_read_off, n_reads,
                           (uint16_t *)d_pml, d_cid, d_order);
    else
        hipLaunchKernelGGL(s2_query_kernel<uint32_t>, grid, block, 0, stream, T, d_bases, d_read_off, n_reads,
                           (uint32_t *)d_pml, d_cid, d_order);
}

}  // namespace colbwt
