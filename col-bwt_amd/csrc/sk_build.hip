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

#include <functional>
#include <string>
#include <vector>

#include "../../include/colbwt.h"
#include "dev_mem.h"
#include "device_layout.h"
#include "jump_tables.h"
#include "lf_device.h"
#include "query_kernels.h"
#include "read_sampler.h"
#include "refine.h"
#include "sk_layout.h"

namespace colbwt {

namespace {

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

// One refinement pass.  `finish` = also build jump tables, hints and distances (a level
// that is only the source of the next one does not need them).  Returns COLBWT_OK,
// COLBWT_ERR_NOMEM (HBM or the 2^32-2 row limit: a shallower layout may still fit) or
// COLBWT_ERR_HIP (anything else: retrying would only hide it).
template <class Src, int K>
int build_level(const Src &S, uint64_t src_rows, const uint8_t *d_cmap, uint32_t sigma, const HintChars &chars,
                bool finish, SKTable &out, SKBuffers &buf, std::string &err) {
    const uint64_t r = src_rows;
    DevPtr first_buf;
    uint64_t run = 0;
    {
        const int rc = count_and_scan(S, r, first_buf, run, err);
        if (rc != COLBWT_OK) return rc;
    }
    if (run > 0xFFFFFFFEull) {
        err = std::to_string(K) + "-step layout needs " + std::to_string(run) + " rows (> 2^32-2)";
        return COLBWT_ERR_NOMEM;
    }
    uint32_t *d_first = first_buf.as<uint32_t>();
    const uint32_t rblocks = (uint32_t)((r + 255) / 256);
    const uint32_t r_new = (uint32_t)run;

    constexpr uint32_t rpl = SKGeom<K>::kRowsPerLine;
    const uint64_t nlines = ((uint64_t)r_new + 1 + rpl - 1) / rpl + 1;
    {
        PlainAllocScope whole(finish);     // the table the query fetches from: one hipMalloc block (dev_mem.h)
        SK_TRY(buf.lines.alloc(nlines * 128));
    }
    SK_TRY(hipMemset(buf.lines.get(), 0, nlines * 128));
    SK_TRY(buf.idx.alloc(((uint64_t)r_new + 4) * sizeof(uint64_t)));
    SK_TRY(hipMemset(buf.idx.get(), 0xFF, ((uint64_t)r_new + 4) * sizeof(uint64_t)));
    SK_TRY(buf.thr.alloc(((uint64_t)r_new + 1) * sizeof(uint64_t)));
    uint8_t *const d_lines = buf.lines.as<uint8_t>();
    uint64_t *const d_idx = buf.idx.as<uint64_t>(), *const d_thr = buf.thr.as<uint64_t>();
    hipLaunchKernelGGL((sk_emit_kernel<Src, K>), dim3(rblocks), dim3(256), 0, 0, S, d_first, d_lines, d_idx, d_thr, r_new);
    SK_TRY(hipStreamSynchronize(0));
    const uint32_t nblocks = (uint32_t)(((uint64_t)r_new + 255) / 256);
    hipLaunchKernelGGL((sk_link_kernel<Src, K>), dim3(nblocks), dim3(256), 0, 0, S, d_first, d_lines,
                       (const uint64_t *)d_idx, r_new);
    SK_TRY(hipStreamSynchronize(0));
    first_buf.reset();

    out.lines = d_lines;
    out.idx = d_idx;
    out.thr = d_thr;
    out.cmap = d_cmap;
    out.n = S.T.n;
    out.r = r_new;
    out.sigma = sigma;
    out.nblk = (uint32_t)(((uint64_t)r_new + sk_block_rows<K>() - 1) / sk_block_rows<K>());
    out.steps = K;
    out.next_tbl = out.prev_tbl = nullptr;
    if (!finish) return COLBWT_OK;

    const uint64_t entries = (uint64_t)out.nblk * out.sigma;
    SK_TRY(buf.next.alloc((entries ? entries : 1) * sizeof(uint32_t)));
    SK_TRY(buf.prev.alloc((entries ? entries : 1) * sizeof(uint32_t)));
    uint32_t *const d_next = buf.next.as<uint32_t>(), *const d_prev = buf.prev.as<uint32_t>();
    out.next_tbl = d_next;
    out.prev_tbl = d_prev;
    hipLaunchKernelGGL(sk_block_first_last_kernel<K>, dim3((out.nblk + 3) / 4), dim3(256), 0, 0, out, d_next, d_prev);
    SK_TRY(hipStreamSynchronize(0));
    {
        std::vector<uint32_t> first(entries), last(entries), next, prev;
        SK_TRY(hipMemcpy(first.data(), buf.next.get(), entries * sizeof(uint32_t), hipMemcpyDeviceToHost));
        SK_TRY(hipMemcpy(last.data(), buf.prev.get(), entries * sizeof(uint32_t), hipMemcpyDeviceToHost));
        finish_jump_tables(first, last, out.nblk, out.sigma, next, prev);
        SK_TRY(hipMemcpy(buf.next.get(), next.data(), entries * sizeof(uint32_t), hipMemcpyHostToDevice));
        SK_TRY(hipMemcpy(buf.prev.get(), prev.data(), entries * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(sk_hint_kernel<K>, dim3(nblocks), dim3(256), 0, 0, out, d_lines, chars);
    SK_TRY(hipGetLastError());
    SK_TRY(hipStreamSynchronize(0));
    return COLBWT_OK;
}

template <int K>
struct SKView {
    SKTable T;
    __device__ __forceinline__ uint64_t n() const { return T.n; }
    __device__ __forceinline__ uint32_t rows() const { return T.r; }
    __device__ __forceinline__ uint64_t idx(uint32_t j) const { return T.idx[j]; }
    __device__ __forceinline__ SKRow<K> load(uint32_t j) const { return sk_load<K>(T, j); }
    __device__ __forceinline__ uint32_t ch(const SKRow<K> &w) const { return sk_char<K>(w); }
    __device__ __forceinline__ uint32_t lf_row(const SKRow<K> &w) const { return sk_I<K>(w, 1); }
    __device__ __forceinline__ uint32_t lf_off(const SKRow<K> &w) const { return sk_O<K>(w, 1); }
    __device__ __forceinline__ uint64_t len(uint32_t, const SKRow<K> &w) const { return sk_len<K>(w); }
};

// The read sampler over a K-step table (read_sampler.h): the one-step tables are freed once the
// refined rows exist.
template <int K>
__global__ __launch_bounds__(256) void sk_synth_reads_kernel(SKView<K> V, uint64_t n_reads, uint32_t m, uint32_t sub_permille,
                                                             uint64_t seed, uint8_t *__restrict__ bases,
                                                             uint64_t *__restrict__ read_off) {
    const uint64_t rd = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (rd > n_reads) return;
    read_off[rd] = rd * m;
    if (rd == n_reads) return;
    sample_read(V, rd, m, sub_permille, seed, bases + rd * m);
}

}  // namespace

void launch_sk_synth_reads(const SKTable &T, uint64_t n_reads, uint32_t read_len, uint32_t sub_permille, uint64_t seed,
                           uint8_t *d_bases, uint64_t *d_read_off, hipStream_t stream) {
    const uint32_t blocks = (uint32_t)((n_reads + 1 + 255) / 256);
    if (T.steps == 3)
        hipLaunchKernelGGL(sk_synth_reads_kernel<3>, dim3(blocks), dim3(256), 0, stream, SKView<3>{T}, n_reads, read_len,
                           sub_permille, seed, d_bases, d_read_off);
    else
        hipLaunchKernelGGL(sk_synth_reads_kernel<2>, dim3(blocks), dim3(256), 0, stream, SKView<2>{T}, n_reads, read_len,
                           sub_permille, seed, d_bases, d_read_off);
}

void SKBuffers::release() {
    for (DevPtr *p : {&lines, &idx, &thr, &next, &prev}) p->reset();
}

uint64_t SKBuffers::bytes() const {
    return lines.bytes() + idx.bytes() + thr.bytes() + next.bytes() + prev.bytes();
}

// Builds the `steps`-step layout (2 or 3) from the one-step tables of `T`.  Returns COLBWT_OK or
// the code of what went wrong with `err` set; on failure nothing stays allocated.  Once the last
// pass that reads the one-step tables is done, `source_done` is called: the owner frees them
// before the next level is allocated (they are not needed by the K-step query).
int build_sk(const DevTable &T, const HintChars &chars, int steps, SKTable &out, SKBuffers &buf, std::string &err,
             const std::function<void()> &source_done) {
    SrcL1 s1{T, chars};
    const uint8_t *cmap = T.cmap;
    const uint32_t sigma = T.sigma;
    int rc;
    if (steps == 2) {
        rc = build_level<SrcL1, 2>(s1, T.r, cmap, sigma, chars, true, out, buf, err);
        if (rc == COLBWT_OK) source_done();
    } else {
        // level 3 is refined from a temporary level 2
        SKTable t2{};
        SKBuffers b2;
        rc = build_level<SrcL1, 2>(s1, T.r, cmap, sigma, chars, false, t2, b2, err);
        if (rc == COLBWT_OK) {
            source_done();
            SrcSK<2> s2{t2};
            rc = build_level<SrcSK<2>, 3>(s2, t2.r, cmap, sigma, chars, true, out, buf, err);
        }
        b2.release();
    }
    if (rc != COLBWT_OK) buf.release();
    return rc;
}

}  // namespace colbwt
