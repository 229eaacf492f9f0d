// sk_build.hip -- builds the K-step layouts (sk_layout.h) on the device.
// Level 2 is refined from the one-step tables (device_layout.h), level 3 from
// level 2.  Runs once per index, off the query path.
//
//   count   per source row: number of new rows = pieces of its LF image between
//           source-row boundaries (+ cuts at 65534 positions)
//   scan    exclusive prefix sum -> first new row of every source row
//   emit    new rows: idx, len, char, col id, look-ahead chars/ids, and the image
//           of their first position in SOURCE coordinates, parked in the I slots
//   link    parked image -> landings of LF .. LF^K in NEW coordinates, next-row lengths
//   finish  the same jump tables, threshold hints and mismatch distances as the
//           one-step layout
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "device_layout.h"
#include "jump_tables.h"
#include "lf_device.h"
#include "query_kernels.h"
#include "sk_layout.h"

namespace colbwt {

namespace {

// ---- source views: what a refinement pass needs from the level below ------
struct SrcL1 {  // the one-step table
    DevTable T;
    HintChars chars;
    static constexpr int kSteps = 1;
    // Positions strictly inside row i where a mismatch on one of the hinted characters
    // changes sides: the threshold of that character's next run (col_bwt.hpp:552-560).
    // The first refinement cuts there too, so that no refined row contains a threshold
    // and every hint is decided -- in a real index the threshold of a run lies between
    // the previous run of its character and its head, i.e. inside one of the rows in
    // between, and the query would otherwise fall back to scans + position compares.
    __device__ __forceinline__ uint32_t cuts(uint32_t i, uint64_t (&cut)[kHintSlots]) const {
        const uint4 w = T.rows[i];
        const uint32_t aidx = T.cmap[row_char(w)];
        const uint32_t top = T.sigma < kHintMaxSigma ? T.sigma : kHintMaxSigma;
        uint32_t nc = 0;
        for (uint32_t cidx = 0; cidx < top; ++cidx) {
            const uint32_t slot = hint_slot(cidx, aidx);
            if (cidx == aidx || slot >= kHintSlots) continue;
            if (((row_hints(w) >> (2 * slot)) & 3u) != kHintCompare) continue;
            uint4 t;
            const uint32_t s = succ_char(T, i, chars.c[cidx], cidx, t);
            if (s == kNone) continue;                  // thr = n: never inside a row
            const uint64_t thr = T.thr[s];
            uint32_t q = nc++;                         // insertion sort, ascending
            while (q > 0 && cut[q - 1] > thr) { cut[q] = cut[q - 1]; --q; }
            cut[q] = thr;
        }
        return nc;
    }
    __device__ __forceinline__ uint32_t rows() const { return T.r; }
    __device__ __forceinline__ uint64_t n() const { return T.n; }
    __device__ __forceinline__ uint64_t idx(uint32_t j) const { return T.idx[j]; }
    __device__ __forceinline__ uint64_t len(uint32_t j) const { return T.idx[(uint64_t)j + 1] - T.idx[j]; }
    __device__ __forceinline__ uint64_t thr(uint32_t j) const { return T.thr[j]; }
    // (row, offset) of LF^s(first position of row j); may still need the fast-forward
    __device__ __forceinline__ void lf(uint32_t j, int, uint32_t &dj, uint64_t &dt) const {
        const uint4 w = T.rows[j];
        dj = row_interval(w);
        dt = row_offset(w);
    }
    // character / col id met after a-1 LF steps from any position of row j (a = 1 only)
    __device__ __forceinline__ uint32_t ch_at(uint32_t j, int) const { return row_char(T.rows[j]); }
    __device__ __forceinline__ uint32_t cid_at(uint32_t j, int) const { return row_cid(T.rows[j]); }
};

template <int KS>
struct SrcSK {  // a K-step table as the source of the next level
    SKTable T;
    static constexpr int kSteps = KS;
    __device__ __forceinline__ uint32_t cuts(uint32_t, uint64_t (&)[kHintSlots]) const { return 0; }   // cut at level 2 already
    __device__ __forceinline__ uint32_t rows() const { return T.r; }
    __device__ __forceinline__ uint64_t n() const { return T.n; }
    __device__ __forceinline__ uint64_t idx(uint32_t j) const { return T.idx[j]; }
    __device__ __forceinline__ uint64_t len(uint32_t j) const { return T.idx[(uint64_t)j + 1] - T.idx[j]; }
    __device__ __forceinline__ uint64_t thr(uint32_t j) const { return T.thr[j]; }
    __device__ __forceinline__ void lf(uint32_t j, int s, uint32_t &dj, uint64_t &dt) const {
        const SKRow<KS> w = sk_load<KS>(T, j);
        dj = sk_I<KS>(w, (uint32_t)s);
        dt = sk_O<KS>(w, (uint32_t)s);
    }
    __device__ __forceinline__ uint32_t ch_at(uint32_t j, int a) const {
        const SKRow<KS> w = sk_load<KS>(T, j);
        if (a == 1) return sk_char<KS>(w);
        if constexpr (KS >= 3) { if (a == 3) return sk_char_at<KS, 3>(w); }
        return sk_char_at<KS, 2>(w);
    }
    __device__ __forceinline__ uint32_t cid_at(uint32_t j, int a) const {
        const SKRow<KS> w = sk_load<KS>(T, j);
        if (a == 1) return sk_cid<KS>(w);
        if constexpr (KS >= 3) { if (a == 3) return sk_cid_at<KS, 3>(w); }
        return sk_cid_at<KS, 2>(w);
    }
};

// Fast-forward (LF_table.hpp:256-259) of (j, t) over the source rows.
template <class Src>
__device__ __forceinline__ void src_ff(const Src &S, uint32_t &j, uint64_t &t) {
    uint64_t lenj = S.len(j);
    while (t >= lenj && j < S.rows() - 1) {
        t -= lenj;
        ++j;
        lenj = S.len(j);
    }
}

// Walks the LF image of source row i piece by piece: f(piece_start, piece_len, j, t)
// with (j, t) = source row / offset the piece's first position maps to.
template <class Src, typename F>
__device__ __forceinline__ void for_each_piece(const Src &S, uint32_t i, F f) {
    uint64_t rem = S.len(i);
    uint64_t b = S.idx(i);
    uint64_t cut[kHintSlots];
    const uint32_t nc = S.cuts(i, cut);
    uint32_t ci = 0;
    uint32_t j;
    uint64_t t;
    S.lf(i, 1, j, t);
    src_ff(S, j, t);
    uint64_t lenj = S.len(j);
    while (rem > 0) {
        const uint64_t avail = (j < S.rows() - 1 && t < lenj) ? lenj - t : rem;  // the last row absorbs everything
        uint64_t take = avail < rem ? avail : rem;
        while (ci < nc && cut[ci] <= b) ++ci;                                   // thresholds inside the row
        if (ci < nc && cut[ci] - b < take) take = cut[ci] - b;
        rem -= take;
        while (take > 0) {               // cut pieces longer than kSKMaxLen
            const uint64_t piece = take < kSKMaxLen ? take : kSKMaxLen;
            f(b, (uint32_t)piece, j, t);
            b += piece;
            t += piece;
            take -= piece;
        }
        if (rem > 0 && t >= lenj && j < S.rows() - 1) {
            ++j;
            t = 0;
            lenj = S.len(j);
        }
    }
}

template <class Src>
__global__ __launch_bounds__(256) void sk_count_kernel(Src S, uint32_t *__restrict__ count) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= S.rows()) return;
    uint32_t pieces = 0;
    for_each_piece(S, (uint32_t)i, [&](uint64_t, uint32_t, uint32_t, uint64_t) { ++pieces; });
    count[i] = pieces;
}

// Block-level exclusive scan of 1024 items per block; block totals go to `totals`.
__global__ __launch_bounds__(256) void scan_block_kernel(uint32_t *__restrict__ data, uint64_t n,
                                                         uint32_t *__restrict__ totals) {
    __shared__ uint32_t s_sum[256];
    const uint64_t base = (uint64_t)blockIdx.x * 1024 + (uint64_t)threadIdx.x * 4;
    uint32_t v[4], run = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[q] = base + q < n ? data[base + q] : 0;
        const uint32_t x = v[q];
        v[q] = run;
        run += x;
    }
    s_sum[threadIdx.x] = run;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {   // Hillis-Steele over the 256 per-thread sums
        const uint32_t add = threadIdx.x >= d ? s_sum[threadIdx.x - d] : 0;
        __syncthreads();
        s_sum[threadIdx.x] += add;
        __syncthreads();
    }
    const uint32_t before = threadIdx.x ? s_sum[threadIdx.x - 1] : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (base + q < n) data[base + q] = v[q] + before;
    if (threadIdx.x == 255) totals[blockIdx.x] = s_sum[255];
}

__global__ __launch_bounds__(256) void scan_add_kernel(uint32_t *__restrict__ data, uint64_t n,
                                                       const uint32_t *__restrict__ block_off) {
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint64_t k = i + (uint64_t)q * 256;
        if (k < n) data[k] += block_off[blockIdx.x];
    }
}

template <int K>
__device__ __forceinline__ uint32_t *sk_row_ptr(uint8_t *lines, uint32_t j) {
    return reinterpret_cast<uint32_t *>(lines + sk_row_off<K>(j));
}

template <class Src, int K>
__global__ __launch_bounds__(256) void sk_emit_kernel(Src S, const uint32_t *__restrict__ first, uint8_t *__restrict__ lines,
                                                      uint64_t *__restrict__ idx_new, uint64_t *__restrict__ thr_new,
                                                      uint32_t r_new) {
    static_assert(Src::kSteps == K - 1, "level K is refined from level K-1");
    constexpr uint32_t kL = SKGeom<K>::kLen;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= S.rows()) return;
    const uint32_t ch = S.ch_at((uint32_t)i, 1), cidv = S.cid_at((uint32_t)i, 1);
    // look-ahead constants the source row already knows (after 1 .. K-2 steps)
    uint32_t ch2 = 0, cid2 = 0;
    if constexpr (K >= 3) { ch2 = S.ch_at((uint32_t)i, 2); cid2 = S.cid_at((uint32_t)i, 2); }
    const uint64_t thr = S.thr((uint32_t)i);
    uint32_t out = first[i];
    for_each_piece(S, (uint32_t)i, [&](uint64_t b, uint32_t len, uint32_t j, uint64_t t) {
        uint32_t *p = sk_row_ptr<K>(lines, out);
        // the deepest look-ahead comes from the source row the piece maps into
        const uint32_t chK = S.ch_at(j, K - 1), cidK = S.cid_at(j, K - 1);
        p[0] = j;                       // parked: image of the first position, SOURCE coordinates
        p[1] = (uint32_t)t;
        p[2] = (uint32_t)(t >> 32);
        if constexpr (K == 2) {
            p[kL + 2] = (chK << 8) | (cidK << 16) | (kHintAllCompare << 24) | 0xFFu;
        } else {
            p[3] = 0;
            p[4] = (chK << 16) | (cidK << 24);
            p[kL + 2] = (ch2 << 8) | (cid2 << 16) | (kHintAllCompare << 24) | 0xFFu;
        }
        p[kL] = len | (ch << 16) | (cidv << 24);
        p[kL + 1] = 0xFFFFFFFFu;        // next-row lengths unknown, mismatch targets far (filled later)
        idx_new[out] = b;
        thr_new[out] = thr;
        ++out;
    });
    if (i + 1 == S.rows()) {            // sentinel row: idx = n
        uint32_t *p = sk_row_ptr<K>(lines, r_new);
        for (uint32_t q = 0; q < SKGeom<K>::kDwords; ++q) p[q] = 0;
        p[kL + 1] = 0xFFFFFFFFu;
        idx_new[r_new] = S.n();
    }
}

// New row holding BWT position `pos`, which lies in source row j.
__device__ __forceinline__ uint32_t sk_find(const uint64_t *idx_new, const uint32_t *first, uint32_t j, uint64_t pos) {
    uint32_t lo = first[j], hi = first[j + 1];   // rows lo .. hi-1 tile source row j
    while (hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (idx_new[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

template <class Src, int K>
__global__ __launch_bounds__(256) void sk_link_kernel(Src S, const uint32_t *__restrict__ first, uint8_t *lines,
                                                      const uint64_t *__restrict__ idx_new, uint32_t r_new) {
    constexpr uint32_t kL = SKGeom<K>::kLen, kO = SKGeom<K>::kO;
    const uint64_t i2 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i2 >= r_new) return;
    uint32_t *p = sk_row_ptr<K>(lines, (uint32_t)i2);
    const uint32_t d = p[0];                                     // LF(first position) = (d, t) in source coordinates
    const uint64_t t = (uint64_t)p[1] | ((uint64_t)p[2] << 32);
    uint32_t I[K], O[K];
    {
        const uint64_t pos = S.idx(d) + t;
        I[0] = sk_find(idx_new, first, d, pos);
        O[0] = (uint32_t)(pos - idx_new[I[0]]);
    }
#pragma unroll
    for (int s = 2; s <= K; ++s) {   // LF^s(first) = LF^(s-1) applied at offset t of source row d
        uint32_t e;
        uint64_t u;
        S.lf(d, s - 1, e, u);
        u += t;
        src_ff(S, e, u);
        const uint64_t pos = S.idx(e) + u;
        I[s - 1] = sk_find(idx_new, first, e, pos);
        O[s - 1] = (uint32_t)(pos - idx_new[I[s - 1]]);
    }
    p[0] = I[0];
    p[1] = I[1];
    if constexpr (K == 2) {
        p[kO] = (O[0] & 0xFFFFu) | (O[1] << 16);
    } else {
        p[2] = I[2];
        p[kO] = (O[0] & 0xFFFFu) | (O[1] << 16);
        p[kO + 1] = (p[kO + 1] & 0xFFFF0000u) | (O[2] & 0xFFFFu);
    }
    // where the image under the deepest jump, LF^K, leaves its landing row and the row after it
    // (sk_layout.h); distances stay "far" until the hint pass
    uint32_t cut = kSKCutNone, len_b = kSKCutNone;
    if ((uint64_t)I[K - 1] + 1 < r_new) {
        const uint64_t c = idx_new[(uint64_t)I[K - 1] + 1] - idx_new[I[K - 1]] - O[K - 1];
        if (c < kSKCutNone) {
            cut = (uint32_t)c;
            if ((uint64_t)I[K - 1] + 2 < r_new) {
                const uint64_t l = idx_new[(uint64_t)I[K - 1] + 2] - idx_new[(uint64_t)I[K - 1] + 1];
                if (l < kSKCutNone) len_b = (uint32_t)l;
            }
        }
    }
    cut |= len_b << 4;
    p[kL + 1] = cut | 0xFFFFFF00u;
    p[kL + 2] |= 0xFFu;
}

// One wave per jump block (see block_first_last_kernel in index_kernels.hip).
template <int K>
__global__ __launch_bounds__(256) void sk_block_first_last_kernel(SKTable T, uint32_t *__restrict__ first,
                                                                  uint32_t *__restrict__ last) {
    constexpr int kChunks = sk_block_rows<K>() / 64;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= T.nblk) return;
    const uint64_t base = (uint64_t)b * sk_block_rows<K>();
    uint32_t cx[kChunks];
#pragma unroll
    for (int s = 0; s < kChunks; ++s) {
        const uint64_t row = base + (uint64_t)s * 64 + lane;
        cx[s] = kNone;
        if (row < T.r) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(T.lines + sk_row_off<K>((uint32_t)row));
            cx[s] = T.cmap[(p[SKGeom<K>::kLen] >> 16) & 0xFFu];
        }
    }
    for (uint32_t c = 0; c < T.sigma; ++c) {
        uint32_t f = kNone, l = kNone;
#pragma unroll
        for (int s = 0; s < kChunks; ++s) {
            const unsigned long long m = __ballot(cx[s] == c);
            if (m) {
                const uint32_t lo = (uint32_t)base + s * 64 + (uint32_t)__builtin_ctzll(m);
                const uint32_t hi = (uint32_t)base + s * 64 + 63u - (uint32_t)__builtin_clzll(m);
                if (f == kNone) f = lo;
                l = hi;
            }
        }
        if (lane == 0) {
            first[(uint64_t)b * T.sigma + c] = f;
            last[(uint64_t)b * T.sigma + c] = l;
        }
    }
}

// Threshold hints and mismatch distances (see hint_kernel in index_kernels.hip).
template <int K>
__global__ __launch_bounds__(256) void sk_hint_kernel(SKTable T, uint8_t *lines_rw, HintChars chars) {
    constexpr uint32_t kL = SKGeom<K>::kLen;
    const uint64_t i64 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i64 >= T.r) return;
    const uint32_t i = (uint32_t)i64;
    const SKRow<K> w = sk_load<K>(T, i);
    const uint32_t aidx = T.cmap[sk_char<K>(w)];
    const uint64_t lo = T.idx[i];
    const uint64_t hi = lo + sk_len<K>(w) - 1;
    uint32_t hints = kHintAllCompare, dists = 0xFFFFFFFFu;   // 4 x 8 bits
    const uint32_t top = T.sigma < kHintMaxSigma ? T.sigma : kHintMaxSigma;
    for (uint32_t cidx = 0; cidx < top; ++cidx) {
        if (cidx == aidx || hint_slot(cidx, aidx) >= kHintSlots) continue;
        SKRow<K> t;
        const uint32_t s = sk_succ_char<K>(T, i, chars.c[cidx], cidx, t);
        const uint64_t thr = (s != kNone) ? T.thr[s] : T.n;
        const uint32_t code = hi < thr ? kHintPred : (lo >= thr ? kHintSucc : kHintCompare);
        const uint32_t slot = hint_slot(cidx, aidx);
        hints = (hints & ~(3u << (2 * slot))) | (code << (2 * slot));
        uint32_t dist = kSKDistFar;
        if (code == kHintSucc && s - i < kSKDistFar) dist = s - i;
        if (code == kHintPred) {
            const uint32_t q = sk_pred_char<K>(T, i, chars.c[cidx], cidx, t);
            if (q != kNone && i - q < kSKDistFar) dist = i - q;
        }
        dists = (dists & ~(0xFFu << (8 * slot))) | (dist << (8 * slot));
    }
    uint32_t *p = sk_row_ptr<K>(lines_rw, i);
    p[kL + 1] = (w.d[kL + 1] & 0x000000FFu) | (dists << 8);                       // slots 0..2
    p[kL + 2] = (w.d[kL + 2] & 0x00FFFF00u) | (hints << 24) | (dists >> 24);      // slot 3
}

#define SK_TRY(expr)                                                          \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) {                                               \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);          \
            return false;                                                     \
        }                                                                     \
    } while (0)

// One refinement pass.  `finish` = also build jump tables, hints and distances (a level
// that is only the source of the next one does not need them).
template <class Src, int K>
bool build_level(const Src &S, uint64_t src_rows, const uint8_t *d_cmap, uint32_t sigma, const HintChars &chars,
                 bool finish, SKTable &out, SKBuffers &buf, std::string &err) {
    const uint64_t r = src_rows;
    uint32_t *d_first = nullptr, *d_tot = nullptr;
    SK_TRY(hipMalloc((void **)&d_first, (r + 1) * sizeof(uint32_t)));
    const uint32_t rblocks = (uint32_t)((r + 255) / 256);
    hipLaunchKernelGGL(sk_count_kernel<Src>, dim3(rblocks), dim3(256), 0, 0, S, d_first);
    SK_TRY(hipGetLastError());
    SK_TRY(hipMemset(d_first + r, 0, sizeof(uint32_t)));   // the extra slot receives the total
    const uint64_t nscan = r + 1;
    const uint32_t sblocks = (uint32_t)((nscan + 1023) / 1024);
    SK_TRY(hipMalloc((void **)&d_tot, sblocks * sizeof(uint32_t)));
    hipLaunchKernelGGL(scan_block_kernel, dim3(sblocks), dim3(256), 0, 0, d_first, nscan, d_tot);
    SK_TRY(hipStreamSynchronize(0));
    std::vector<uint32_t> tot(sblocks);
    SK_TRY(hipMemcpy(tot.data(), d_tot, sblocks * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint64_t run = 0;
    for (uint32_t b = 0; b < sblocks; ++b) {
        const uint64_t x = tot[b];
        tot[b] = (uint32_t)run;
        run += x;
    }
    if (run > 0xFFFFFFFEull) {
        (void)hipFree(d_first);
        (void)hipFree(d_tot);
        err = std::to_string(K) + "-step layout needs " + std::to_string(run) + " rows (> 2^32-2)";
        return false;
    }
    SK_TRY(hipMemcpy(d_tot, tot.data(), sblocks * sizeof(uint32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(scan_add_kernel, dim3(sblocks), dim3(256), 0, 0, d_first, nscan, d_tot);
    SK_TRY(hipStreamSynchronize(0));
    (void)hipFree(d_tot);
    const uint32_t r_new = (uint32_t)run;

    constexpr uint32_t rpl = SKGeom<K>::kRowsPerLine;
    const uint64_t nlines = ((uint64_t)r_new + 1 + rpl - 1) / rpl + 1;
    SK_TRY(hipMalloc(&buf.lines, nlines * 128));
    SK_TRY(hipMemset(buf.lines, 0, nlines * 128));
    SK_TRY(hipMalloc(&buf.idx, ((uint64_t)r_new + 4) * sizeof(uint64_t)));
    SK_TRY(hipMemset(buf.idx, 0xFF, ((uint64_t)r_new + 4) * sizeof(uint64_t)));
    SK_TRY(hipMalloc(&buf.thr, (uint64_t)r_new * sizeof(uint64_t)));
    hipLaunchKernelGGL((sk_emit_kernel<Src, K>), dim3(rblocks), dim3(256), 0, 0, S, d_first, (uint8_t *)buf.lines,
                       (uint64_t *)buf.idx, (uint64_t *)buf.thr, r_new);
    SK_TRY(hipStreamSynchronize(0));
    const uint32_t nblocks = (uint32_t)(((uint64_t)r_new + 255) / 256);
    hipLaunchKernelGGL((sk_link_kernel<Src, K>), dim3(nblocks), dim3(256), 0, 0, S, d_first, (uint8_t *)buf.lines,
                       (const uint64_t *)buf.idx, r_new);
    SK_TRY(hipStreamSynchronize(0));
    (void)hipFree(d_first);

    out.lines = (const uint8_t *)buf.lines;
    out.idx = (const uint64_t *)buf.idx;
    out.thr = (const uint64_t *)buf.thr;
    out.cmap = d_cmap;
    out.n = S.T.n;
    out.r = r_new;
    out.sigma = sigma;
    out.nblk = (uint32_t)(((uint64_t)r_new + sk_block_rows<K>() - 1) / sk_block_rows<K>());
    out.steps = K;
    out.next_tbl = out.prev_tbl = nullptr;
    buf.bytes = nlines * 128 + (2 * (uint64_t)r_new + 4) * sizeof(uint64_t);
    if (!finish) return true;

    const uint64_t entries = (uint64_t)out.nblk * out.sigma;
    SK_TRY(hipMalloc(&buf.next, (entries ? entries : 1) * sizeof(uint32_t)));
    SK_TRY(hipMalloc(&buf.prev, (entries ? entries : 1) * sizeof(uint32_t)));
    out.next_tbl = (const uint32_t *)buf.next;
    out.prev_tbl = (const uint32_t *)buf.prev;
    hipLaunchKernelGGL(sk_block_first_last_kernel<K>, dim3((out.nblk + 3) / 4), dim3(256), 0, 0, out,
                       (uint32_t *)buf.next, (uint32_t *)buf.prev);
    SK_TRY(hipStreamSynchronize(0));
    {
        std::vector<uint32_t> first(entries), last(entries), next, prev;
        SK_TRY(hipMemcpy(first.data(), buf.next, entries * sizeof(uint32_t), hipMemcpyDeviceToHost));
        SK_TRY(hipMemcpy(last.data(), buf.prev, entries * sizeof(uint32_t), hipMemcpyDeviceToHost));
        finish_jump_tables(first, last, out.nblk, out.sigma, next, prev);
        SK_TRY(hipMemcpy(buf.next, next.data(), entries * sizeof(uint32_t), hipMemcpyHostToDevice));
        SK_TRY(hipMemcpy(buf.prev, prev.data(), entries * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(sk_hint_kernel<K>, dim3(nblocks), dim3(256), 0, 0, out, (uint8_t *)buf.lines, chars);
    SK_TRY(hipGetLastError());
    SK_TRY(hipStreamSynchronize(0));
    buf.bytes += 2 * (entries ? entries : 1) * sizeof(uint32_t);
    return true;
}

}  // namespace

void SKBuffers::release() {
    for (void **p : {&lines, &idx, &thr, &next, &prev}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    bytes = 0;
}

// Builds the `steps`-step layout (2 or 3) from the one-step tables of `T`.  Returns false
// with `err` set when it cannot (more than 2^32-2 rows, out of memory).
bool build_sk(const DevTable &T, const HintChars &chars, int steps, SKTable &out, SKBuffers &buf, std::string &err) {
    SrcL1 s1{T, chars};
    if (steps == 2) return build_level<SrcL1, 2>(s1, T.r, T.cmap, T.sigma, chars, true, out, buf, err);
    // level 3 is refined from a temporary level 2
    SKTable t2{};
    SKBuffers b2;
    bool ok = build_level<SrcL1, 2>(s1, T.r, T.cmap, T.sigma, chars, false, t2, b2, err);
    if (ok) {
        SrcSK<2> s2{t2};
        ok = build_level<SrcSK<2>, 3>(s2, t2.r, T.cmap, T.sigma, chars, true, out, buf, err);
    }
    b2.release();
    if (!ok) buf.release();
    return ok;
}

}  // namespace colbwt
