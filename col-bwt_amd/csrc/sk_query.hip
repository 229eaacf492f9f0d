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
#include <stdlib.h>

#include "device_layout.h"
#include "lane_io.h"
#include "lf_device.h"
#include "query_kernels.h"
#include "sk_layout.h"

namespace colbwt {

// The part of col_pml::threshold_step (col_bwt.hpp:531-574) the kernel cannot turn into
// a single row load: the target is far away (scan, then the jump tables) or the row's
// hint says the threshold falls inside the row (compare positions, :560).  cidx is the
// dense index of c (present in the BWT); see query_kernels.hip for the hint logic.
template <int K>
__device__ __forceinline__ void sk_threshold_scan(const SKTable &T, uint32_t &i, uint32_t &o, SKRow<K> &w, uint32_t c,
                                               uint32_t cidx, uint32_t hint) {
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

// What the row being loaded is for travels in the offset register: an LF landing still to be
// placed carries its offset (< 2^17), the other arrivals one of these codes.
constexpr uint32_t kOffLastPos = 0xFFFFFFFFu;   // LF-style arrival, offset clamped to the row's last position
constexpr uint32_t kOffPred = 0xFFFFFFFEu;      // threshold target reached from below: offset = len - 1
constexpr uint32_t kOffSucc = 0xFFFFFFFDu;      // threshold target reached from above: offset = 0

// One row load per loop trip.  A lane's work is a chain of dependent row loads
// (LF landing, fast-forward hop, threshold target); the wave pays one memory
// round trip per LOAD SITE it passes, so all three kinds share the single load
// at the top of the loop and each lane spends the trip on whatever its own
// chain needs next.  A wave then runs for max-over-lanes(loads of the lane)
// round trips instead of iterations x (LF + hops + threshold) round trips.
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

    OutAcc18 acc;
    SlidingWindow win;
    win.init(off + m - 1);

    auto emit = [&](uint64_t g, uint32_t len, uint32_t col_id) {   // :525
        if constexpr (kWide) {
            pml[g] = (PmlT)len;
            cid[g] = (uint8_t)col_id;
        } else {
            acc.push(len, col_id);
        }
    };

    // col_bwt.hpp:503-508: pos = n-1 = the last position of the last row.  Expressed as an LF
    // arrival at the last row with an offset beyond it, which the arrival clamps to len - 1.
    uint32_t j = T.r - 1;
    uint32_t o = kOffLastPos;
    uint32_t L = 0;
    uint64_t k = m;                                          // bases not yet reported
    for (;;) {
        uint64_t g = off + k - 1;                            // :512 pattern[m-i-1] is the next base
        // the trip's memory traffic besides the row, issued while the wave is converged
        if constexpr (!kWide) acc.flush_group((uint16_t *)pml, cid, g + 1);
        {
            const uint32_t want = k < (uint32_t)K ? (uint32_t)k : (uint32_t)K;   // bases a trip may consume
            if (__any(win.avail(g) < want)) win.refill(s_rd, bases, g);
        }
        SKRow<K> w = sk_load<K>(T, j);
        bool jump = true;                                    // false: j already names the next row to load
        if (o != kOffPred && o != kOffSucc) {
            const uint32_t len = sk_len<K>(w);
            if (o >= len && j < T.r - 1) {
                // fast-forward of LF_table::LF (LF_table.hpp:256-259) over level-K rows: one row
                // per trip (the first step of most of them was taken at the jump, see sk_cut)
                o -= len;
                j += 1;
                jump = false;
            } else {
                o = o < len ? o : len - 1;
                const uint32_t c = win.get(s_rd, g);         // raw byte
                const uint32_t col_id = sk_cid<K>(w);        // :513 before any re-orientation
                const bool match = sk_char<K>(w) == c;       // :516
                L = match ? L + 1 : 0;                       // :517 / :521
                --k;
                emit(g, L, col_id);                  // :525
                if (k == 0) break;                           // the last LF (:527) has no observable effect
                --g;
                if (!match) {                                // :522 threshold_step
                    const uint32_t cidx = s_cmap[c];
                    if (cidx != kAbsent) {                   // else (interval, offset) unchanged (:533-534)
                        uint32_t hint = kHintCompare, dist = kSKDistFar;
                        const uint32_t hs = hint_slot(cidx, s_cmap[sk_char<K>(w)]);
                        if (hs < kHintSlots) {
                            hint = (sk_hints<K>(w) >> (2 * hs)) & 3u;
                            dist = sk_dist<K>(w, hs);
                        }
                        if (dist != kSKDistFar && hint != kHintCompare) {   // decided and close: the
                            j = hint == kHintPred ? j - dist : j + dist;     // target is the next load
                            o = hint == kHintPred ? kOffPred : kOffSucc;
                            jump = false;
                        } else {
                            sk_threshold_scan<K>(T, j, o, w, c, cidx, hint);
                        }
                    }
                }
            }
        } else {
            o = o == kOffPred ? sk_len<K>(w) - 1 : 0;        // LF_table.hpp:282 / :296
        }
        if (!jump) continue;

        // After a-1 LF steps (:527) every position of this row is in one original row whose
        // character / col id are char_a / cid_a.  While the next base matches it, the next
        // iteration would be ++length with that col id (:513-517) followed by another LF:
        // emit it here and extend the jump by one LF step.
        uint32_t steps = 1;
        bool run = true;
        auto look = [&](uint32_t ch_a, uint32_t cid_a, uint32_t a) {
            if (!run) return;
            if (win.get(s_rd, g) != ch_a) { run = false; return; }
            ++L;
            --k;
            emit(g, L, cid_a);
            steps = a;
            if (k == 0) { run = false; return; }
            --g;
        };
        look(sk_char_at<K, 2>(w), sk_cid_at<K, 2>(w), 2);
        if constexpr (K >= 3) look(sk_char_at<K, 3>(w), sk_cid_at<K, 3>(w), 3);
        if (k == 0) break;
        j = sk_I<K>(w, steps);                               // LF^steps lands at (I_s, O_s + o) ...
        const uint32_t cut = sk_cut_a<K>(w);
        if (steps == (uint32_t)K && cut != kSKCutNone && o >= cut) {
            j += 1;                                          // ... which is already in the next row
            o -= cut;
            const uint32_t lb = sk_len_b<K>(w);
            if (lb != kSKCutNone && o >= lb) {               // ... or in the one after
                j += 1;
                o -= lb;
            }
        } else {
            o += sk_O<K>(w, steps);
        }
    }
    if constexpr (!kWide) {                                  // k == 0: what the last trip pushed
        acc.flush_group((uint16_t *)pml, cid, off);
        acc.flush_rest((uint16_t *)pml, cid, off);
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

void launch_sk_query(const SKTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, uint64_t n_bases,
                     void *d_pml, int pml_bytes, uint8_t *d_cid, const uint32_t *d_order, hipStream_t stream) {
    if (n_reads == 0) return;
    // three-step rows: the kernel with persistent lanes and pair-fetched rows (sk3_query.hip); the one
    // below remains for two-step rows (24-byte rows do not split into 16-byte pieces) and, with
    // COLBWT_SK3_ROUND1=1, for A/B runs
    static const bool round1 = getenv("COLBWT_SK3_ROUND1") != nullptr;
    if (T.steps == 3 && !round1) {
        launch_sk3_query(T, d_bases, d_read_off, n_reads, n_bases, d_pml, pml_bytes, d_cid, stream);
        return;
    }
    if (T.steps == 3) launch_k<3>(T, d_bases, d_read_off, n_reads, d_pml, pml_bytes, d_cid, d_order, stream);
    else launch_k<2>(T, d_bases, d_read_off, n_reads, d_pml, pml_bytes, d_cid, d_order, stream);
}

}  // namespace colbwt
