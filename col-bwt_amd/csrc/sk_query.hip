// sk_query.hip -- the PML / col-ID query over the K-step layouts (sk_layout.h).
//
// Same per-base semantics as query_kernels.hip (col_bwt.hpp:498-574,
// LF_table.hpp:251-298), one lane per read.  The difference is the LF jump: a
// level-K row knows the characters / col ids of the rows all its positions walk
// through during the next K-1 LF steps, so while the NEXT read bases keep
// matching them the lane emits those bases from registers and then takes
// LF^s (s <= K) with a single row load -- one 128-byte line fill and one
// dependent round trip for up to K bases.  The first non-matching look-ahead
// base is NOT consumed: the lane lands with the ordinary jump of the bases it
// did consume and the mismatch is handled there in the next iteration, exactly
// as the reference would.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"
#include "lane_io.h"
#include "lf_device.h"
#include "query_kernels.h"
#include "sk_layout.h"

namespace colbwt {

// col_pml::threshold_step (col_bwt.hpp:531-574) over level-K rows; see
// query_kernels.hip for the hint / distance logic.
template <int K>
__device__ __forceinline__ void sk_threshold_step(const SKTable &T, const uint8_t *s_cmap, uint32_t &i, uint32_t &o,
                                                  SKRow<K> &w, uint32_t c) {
    const uint32_t cidx = s_cmap[c];
    if (cidx == kAbsent) return;  // c occurs nowhere: (interval, offset) unchanged (:533-534)
    uint32_t hint = kHintCompare;
    const uint32_t slot = hint_slot(cidx, s_cmap[sk_char<K>(w)]);
    if (slot < kHintSlots) {
        hint = (sk_hints<K>(w) >> (2 * slot)) & 3u;
        const uint32_t dist = sk_dist<K>(w, slot);
        if (dist != kSKDistFar && hint != kHintCompare) { // decided and close: one load, no scan
            if (hint == kHintPred) {                      // :565-569
                i -= dist;
                w = sk_load<K>(T, i);
                o = sk_len<K>(w) - 1;                     // LF_table.hpp:282
            } else {                                      // :552-557
                i += dist;
                w = sk_load<K>(T, i);
                o = 0;
            }
            return;
        }
    }
    SKRow<K> t;
    if (hint == kHintPred) {
        const uint32_t q = sk_pred_char<K>(T, i, c, cidx, t);      // :562
        if (q != kNone) { i = q; o = sk_len<K>(t) - 1; w = t; return; }   // :565-569
        const uint32_t s = sk_succ_char<K>(T, i, c, cidx, t);      // :548
        if (s != kNone) { i = s; o = 0; w = t; }                    // :552-557
        return;
    }
    if (hint == kHintSucc) {
        const uint32_t s = sk_succ_char<K>(T, i, c, cidx, t);
        if (s != kNone) { i = s; o = 0; w = t; }
        return;
    }
    const uint64_t pos = T.idx[i] + o;    // LF_table::to_idx (LF_table.hpp:214-217)
    uint64_t thr = T.n;                   // :535
    uint32_t ni = i, no = o;
    SKRow<K> nw = w;
    const uint32_t s = sk_succ_char<K>(T, i, c, cidx, t);  // :548
    if (s != kNone) { thr = T.thr[s]; ni = s; no = 0; nw = t; }   // :552-557
    if (pos < thr) {                                        // :560
        const uint32_t q = sk_pred_char<K>(T, i, c, cidx, t);  // :562
        if (q != kNone) { ni = q; no = sk_len<K>(t) - 1; nw = t; }   // :565-569
    }
    i = ni; o = no; w = nw;                                 // :572-573
}

namespace {

template <int K, typename PmlT>
__global__ __launch_bounds__(kQueryBlock) __attribute__((amdgpu_num_sgpr(80), amdgpu_num_vgpr(64)))
void sk_query_kernel(SKTable T, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ read_off,
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

    // col_bwt.hpp:503-508: pos = n-1 = the last position of the last row
    uint32_t i = T.r - 1;
    SKRow<K> w = sk_load<K>(T, i);
    uint32_t o = sk_len<K>(w) - 1;
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
        uint64_t g = off + k - 1;
        const uint32_t c = win.get(s_rd, g);             // :512 pattern[m-i-1], raw byte
        const uint32_t col_id = sk_cid<K>(w);            // :513 before any re-orientation
        if (sk_char<K>(w) == c) {                        // :516
            ++L;
        } else {
            L = 0;                                       // :521
            sk_threshold_step<K>(T, s_cmap, i, o, w, c); // :522
        }
        --k;
        emit(g, L, col_id, k == 0);
        if (k == 0) break;                               // the last LF (:527) has no observable effect
        if ((g & 63) == 0) win.refill(s_rd, bases, g - 1);

        // After a-1 LF steps (:527) every position of this row is in one original row whose
        // character / col id are char_a / cid_a.  While the next base matches it, the next
        // iteration would be ++length with that col id (:513-517) followed by another LF:
        // emit it here and extend the jump by one LF step.
        uint32_t steps = 1;
        bool run = true;
        auto look = [&](uint32_t ch_a, uint32_t cid_a, uint32_t a) {
            if (!run) return;
            if (win.get(s_rd, g - 1) != ch_a) { run = false; return; }
            --g;
            ++L;
            --k;
            emit(g, L, cid_a, k == 0);
            steps = a;
            if (k == 0) { run = false; return; }
            if ((g & 63) == 0) win.refill(s_rd, bases, g - 1);
        };
        look(sk_char_at<K, 2>(w), sk_cid_at<K, 2>(w), 2);
        if constexpr (K >= 3) look(sk_char_at<K, 3>(w), sk_cid_at<K, 3>(w), 3);
        if (k == 0) break;

        // LF^steps, then the fast-forward of LF_table::LF (LF_table.hpp:256-259) over level-K
        // rows, two rows per memory round trip (a row carries the next row's length)
        uint32_t j = sk_I<K>(w, steps);
        uint32_t t = sk_O<K>(w, steps) + o;
        w = sk_load<K>(T, j);
        for (;;) {
            const uint32_t len = sk_len<K>(w);
            if (t < len || j >= T.r - 1) break;
            t -= len;
            uint32_t hop = 1;
            const uint32_t l1 = sk_len8_next1<K>(w);
            if (l1 != kLen8Long && t >= l1 && j + 1 < T.r - 1) {
                t -= l1;
                hop = 2;
            }
            j += hop;
            w = sk_load<K>(T, j);
        }
        i = j;
        o = t;
    }
}

template <int K>
void launch_k(const SKTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, void *d_pml,
              int pml_bytes, uint8_t *d_cid, const uint32_t *d_order, hipStream_t stream) {
    const uint64_t blocks = (n_reads + kQueryBlock - 1) / kQueryBlock;
    dim3 grid((uint32_t)blocks), block(kQueryBlock);
    if (pml_bytes == 2)
        hipLaunchKernelGGL((sk_query_kernel<K, uint16_t>), grid, block, 0, stream, T, d_bases, d_read_off, n_reads,
                           (uint16_t *)d_pml, d_cid, d_order);
    else
        hipLaunchKernelGGL((sk_query_kernel<K, uint32_t>), grid, block, 0, stream, T, d_bases, d_read_off, n_reads,
                           (uint32_t *)d_pml, d_cid, d_order);
}

}  // namespace

void launch_sk_query(const SKTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads,
                     void *d_pml, int pml_bytes, uint8_t *d_cid, const uint32_t *d_order, hipStream_t stream) {
    if (n_reads == 0) return;
    if (T.steps == 3) launch_k<3>(T, d_bases, d_read_off, n_reads, d_pml, pml_bytes, d_cid, d_order, stream);
    else launch_k<2>(T, d_bases, d_read_off, n_reads, d_pml, pml_bytes, d_cid, d_order, stream);
}

}  // namespace colbwt
