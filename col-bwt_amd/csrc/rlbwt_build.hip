// Front of the pipeline (SURVEY.md 8(f) "next" #4): FASTA documents -> run-length BWT
// (.bwt.heads / .bwt.len), thresholds (.thr_pos) and multi-MUMs (.col_mums), the files the
// reference takes from `mumemto mum -K -R -T` (scripts/col-bwt.py:121-145; byte formats:
// col_bwt.hpp:167-171 heads and 5-byte lengths, col_bwt.hpp:446-448 thresholds,
// col_split.cpp:90-106 multi-MUMs).  mumemto is an un-vendored dependency
// (thirdparty/CMakeLists.txt, pinned by branch name only) and prefix-free parsing is a host
// algorithm; here the whole construction runs in HBM:
//
//   suffix array   prefix doubling: rank pairs (rank[i], rank[i+h]) radix-sorted (rocPRIM), h = 8, 16, ..
//                  until every suffix is alone in its group; the rank array of every round is kept
//   LCP            binary lifting over those rank arrays (equal rank in round j <=> equal first
//                  8*2^j characters), a thread per suffix-array position -- no sequential Kasai pass
//   RLBWT          BWT[k] = T[SA[k]-1], run starts by stream compaction
//   thresholds     runs grouped by character (stable radix sort); a wavefront per pair of consecutive
//                  same-character runs takes the first position of the minimum LCP in
//                  (end of the previous run, head of this run]; 0 for a character's first run
//   multi-MUMs     a thread per window of num_docs consecutive suffixes: the LCP of the group
//                  (matches never run across a record separator) is >= min length and larger than
//                  both outer LCPs (exactly num_docs occurrences), the suffixes come from num_docs
//                  different documents, and the preceding characters differ (left-maximal)
//
// What of this is mumemto's own convention cannot be pinned here (its source and outputs are
// absent): "parity unpinned" -- tests compare with oracle/rlbwt_oracle.py (the same definitions,
// written independently) and with a brute-force enumeration of substrings on tiny texts.
//
// Limits: text shorter than 2^32 - 1 characters (32-bit suffix numbers), up to 4096 documents;
// HBM: 29 bytes per character + 4 per doubling round.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/colbwt.h"
#include "dev_mem.h"
#include "rlbwt_build.h"

namespace colbwt {
namespace {

constexpr int kTB = 256;
constexpr int kMaxLevels = 40;
constexpr uint32_t kMaxDocs = 4096;

#define RB_TRY(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
            return e_ == hipErrorOutOfMemory ? COLBWT_ERR_NOMEM : COLBWT_ERR_HIP;             \
        }                                                                                     \
    } while (0)

inline bool no(hipError_t e) { return e != hipSuccess; }
inline unsigned grid_for(uint64_t n) { return (unsigned)std::min<uint64_t>((n + kTB - 1) / kTB, 1u << 20); }

// First 8 characters of every suffix, big-endian, zero-padded past the end (the text's last
// character is its only 0, so padded keys of different suffixes differ).
__global__ void first_keys_kernel(const uint8_t *T, uint64_t n, uint64_t *key, uint32_t *sa) {
    for (uint64_t i = blockIdx.x * (uint64_t)kTB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kTB) {
        uint64_t k = 0;
        for (int b = 0; b < 8; ++b) k = (k << 8) | (i + b < n ? T[i + b] : 0);
        key[i] = k;
        sa[i] = (uint32_t)i;
    }
}

// headpos[k] = k where a new group of equal keys starts, else 0; *n_groups counts the groups.
__global__ void group_heads_kernel(const uint64_t *key, uint64_t n, uint32_t *headpos, unsigned long long *n_groups) {
    unsigned long long mine = 0;
    for (uint64_t k = blockIdx.x * (uint64_t)kTB + threadIdx.x; k < n; k += (uint64_t)gridDim.x * kTB) {
        const bool head = k == 0 || key[k] != key[k - 1];
        headpos[k] = head ? (uint32_t)k : 0u;
        mine += head;
    }
    for (int d = 32; d; d >>= 1) mine += __shfl_down(mine, d);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_groups, mine);
}

__global__ void scatter_rank_kernel(const uint32_t *sa, const uint32_t *group_of, uint64_t n, uint32_t *rank) {
    for (uint64_t k = blockIdx.x * (uint64_t)kTB + threadIdx.x; k < n; k += (uint64_t)gridDim.x * kTB) rank[sa[k]] = group_of[k];
}

// Key of the next round: (rank by the first h characters, rank of the suffix h further on; 0 = past the end).
__global__ void pair_keys_kernel(const uint32_t *sa, const uint32_t *rank, uint64_t n, uint64_t h, int bits, uint64_t *key) {
    for (uint64_t k = blockIdx.x * (uint64_t)kTB + threadIdx.x; k < n; k += (uint64_t)gridDim.x * kTB) {
        const uint64_t i = sa[k];
        const uint64_t second = i + h < n ? (uint64_t)rank[i + h] + 1 : 0;
        key[k] = ((uint64_t)rank[i] << bits) | second;
    }
}

struct Levels {
    const uint32_t *rank[kMaxLevels];   // rank[j]: equal <=> equal first 8 << j characters
    int count;                          // the last one is unique and never consulted
};

__global__ void lcp_kernel(const uint8_t *T, const uint32_t *sa, uint64_t n, Levels lv, uint32_t *lcp) {
    for (uint64_t k = blockIdx.x * (uint64_t)kTB + threadIdx.x; k < n; k += (uint64_t)gridDim.x * kTB) {
        if (k == 0) { lcp[0] = 0; continue; }
        const uint64_t a = sa[k - 1], b = sa[k];
        uint64_t l = 0;
        for (int j = lv.count - 2; j >= 0; --j) {
            const uint64_t h = 8ull << j;
            if (a + l < n && b + l < n && lv.rank[j][a + l] == lv.rank[j][b + l]) l += h;
        }
        while (a + l < n && b + l < n && T[a + l] == T[b + l]) ++l;   // fewer than 8 left
        lcp[k] = (uint32_t)l;
    }
}

__global__ void bwt_kernel(const uint8_t *T, const uint32_t *sa, uint64_t n, uint8_t *bwt) {
    for (uint64_t k = blockIdx.x * (uint64_t)kTB + threadIdx.x; k < n; k += (uint64_t)gridDim.x * kTB) {
        const uint64_t i = sa[k];
        bwt[k] = T[i ? i - 1 : n - 1];
    }
}

// A run is a maximal stretch of equal characters AS THE CONSUMERS SEE THEM: the reference folds
// every byte <= 1 to the terminator when it reads .bwt.heads (col_bwt.hpp:167-171,
// FL_table.hpp:102-104) and hands out one threshold per maximal group of equal folded characters
// (read_thresholds, col_bwt.hpp:446-451).  The final 0 and the separators 1 are therefore ONE
// class here: where the 0 touches a run of 1s (the suffix-array neighbours of suffix 0 are other
// record starts -- always so when documents share a prefix) they are one run with head 1, and the
// k-th threshold written belongs to the k-th group the builder forms.
__device__ __forceinline__ uint8_t fold_sep(uint8_t c) { return c <= 1 ? (uint8_t)1 : c; }

__global__ void run_flags_kernel(const uint8_t *bwt, uint64_t n, uint8_t *flag) {
    for (uint64_t k = blockIdx.x * (uint64_t)kTB + threadIdx.x; k < n; k += (uint64_t)gridDim.x * kTB)
        flag[k] = k == 0 || fold_sep(bwt[k]) != fold_sep(bwt[k - 1]);
}

__global__ void run_chars_kernel(const uint8_t *bwt, const uint32_t *start, uint64_t r, uint8_t *head, uint32_t *id) {
    for (uint64_t j = blockIdx.x * (uint64_t)kTB + threadIdx.x; j < r; j += (uint64_t)gridDim.x * kTB) {
        head[j] = fold_sep(bwt[start[j]]);
        id[j] = (uint32_t)j;
    }
}

// Minimum of (lcp << 32 | position) over every block of kThrBlock positions: the long segments of
// the threshold pass (a rare character -- separators, N -- has runs hundreds of millions of
// positions apart, and ONE wavefront scanning such a gap took seconds) step over whole blocks.
constexpr uint32_t kThrBlock = 2048;

__global__ void block_min_kernel(const uint32_t *lcp, uint64_t n, uint64_t n_blocks, uint64_t *block_min) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)kTB + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * kTB) >> 6;
    for (uint64_t b = wave; b < n_blocks; b += n_waves) {
        const uint64_t lo = b * kThrBlock, hi = min(lo + kThrBlock, n);
        uint64_t best = ~0ull;
        for (uint64_t k = lo + lane; k < hi; k += 64) best = min(best, ((uint64_t)lcp[k] << 32) | k);
        for (int d = 32; d; d >>= 1) best = min(best, (uint64_t)__shfl_xor((unsigned long long)best, d));
        if (lane == 0) block_min[b] = best;
    }
}

// by_char: run numbers sorted stably by character.  One wavefront per entry t: the run j = by_char[t]
// and the previous run of its character p = by_char[t-1]; first minimum of lcp over
// [start[p+1], start[j]].
__global__ void threshold_kernel(const uint32_t *by_char, const uint8_t *head, const uint32_t *start, const uint32_t *lcp,
                                 const uint64_t *block_min, uint64_t r, uint32_t *thr) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)kTB + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * kTB) >> 6;
    for (uint64_t t = wave; t < r; t += n_waves) {
        const uint32_t j = by_char[t];
        if (t == 0 || head[by_char[t - 1]] != head[j]) {
            if (lane == 0) thr[j] = 0;
            continue;
        }
        const uint64_t lo = start[by_char[t - 1] + 1], hi = (uint64_t)start[j] + 1;   // [lo, hi); the previous run is not the last one
        uint64_t best = ~0ull;                                                        // (lcp << 32 | position): first minimum
        uint64_t b_lo = (lo + kThrBlock - 1) / kThrBlock, b_hi = hi / kThrBlock;      // whole blocks [b_lo, b_hi)
        if (b_lo >= b_hi) b_lo = b_hi = hi / kThrBlock + 1;                           // none: one direct scan
        const uint64_t head_end = min(hi, b_lo * kThrBlock), tail_begin = max(lo, b_hi * kThrBlock);
        for (uint64_t k = lo + lane; k < head_end; k += 64) best = min(best, ((uint64_t)lcp[k] << 32) | k);
        for (uint64_t b = b_lo + lane; b < b_hi; b += 64) best = min(best, block_min[b]);
        if (head_end < hi)
            for (uint64_t k = max(tail_begin, head_end) + lane; k < hi; k += 64) best = min(best, ((uint64_t)lcp[k] << 32) | k);
        for (int d = 32; d; d >>= 1) best = min(best, (uint64_t)__shfl_xor((unsigned long long)best, d));
        if (lane == 0) thr[j] = (uint32_t)best;
    }
}

__device__ inline uint32_t lower_bound_u32(const uint32_t *v, uint32_t n, uint32_t x) {   // first index with v[] >= x
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (v[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ inline uint32_t upper_bound_u32(const uint32_t *v, uint32_t n, uint32_t x) {   // first index with v[] > x
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (v[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// A match never contains a separator (a character <= 1): the common prefix of two suffixes is cut at
// the first separator of either (the same offset in both).  seps: ascending separator positions; the
// text's last character is one.
__global__ void cap_lcp_kernel(const uint32_t *sa, const uint32_t *seps, uint32_t n_seps, uint64_t n, uint32_t *lcp) {
    for (uint64_t k = blockIdx.x * (uint64_t)kTB + threadIdx.x; k < n; k += (uint64_t)gridDim.x * kTB) {
        const uint32_t p = sa[k], l = lcp[k];
        if (!l) continue;
        lcp[k] = min(l, seps[lower_bound_u32(seps, n_seps, p)] - p);   // first separator at or after p
    }
}

// Minimum of lcp over every window of w consecutive entries in O(1) per window: the array is cut
// into blocks of w entries, pre[k] = minimum from the start of k's block to k, suf[k] = minimum from
// k to the end of its block; a window [a, a + w) spans at most two blocks, so its minimum is
// min(suf[a], pre[a + w - 1]).  One thread per block (w entries each).
__global__ void window_min_kernel(const uint32_t *lcp, uint64_t n, uint32_t w, uint32_t *pre, uint32_t *suf) {
    const uint64_t n_blocks = (n + w - 1) / w;
    for (uint64_t b = blockIdx.x * (uint64_t)kTB + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * kTB) {
        const uint64_t lo = b * w, hi = min(lo + w, n);
        uint32_t m = ~0u;
        for (uint64_t k = lo; k < hi; ++k) { m = min(m, lcp[k]); pre[k] = m; }
        m = ~0u;
        for (uint64_t k = hi; k > lo; --k) { m = min(m, lcp[k - 1]); suf[k - 1] = m; }
    }
}

// A multi-MUM starts at suffix-array rank i when the n_docs suffixes from there share at least
// min_len characters (the minimum of the n_docs - 1 capped LCPs between them: one window lookup),
// more than with the suffix on either side, are not all preceded by the same character, and come
// one from every document.  The tests run from the cheapest to the dearest: the per-document walk
// (a binary search per suffix, 512 bytes of flags per thread) only happens for the windows that
// passed everything else -- a handful per thousand positions even on hundreds of similar genomes.
__global__ void mum_kernel(const uint32_t *sa, const uint32_t *lcp, const uint32_t *pre, const uint32_t *suf, const uint8_t *bwt,
                           const uint32_t *doc_start, uint32_t n_docs, uint64_t n, uint32_t min_len, uint8_t *flag, uint32_t *mum_len) {
    for (uint64_t i = blockIdx.x * (uint64_t)kTB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kTB) {
        flag[i] = 0;
        if (i + n_docs > n) continue;
        const uint32_t m = min(suf[i + 1], pre[i + n_docs - 1]);          // lcp[i + 1 .. i + n_docs - 1]
        if (m < min_len || lcp[i] >= m || (i + n_docs < n && lcp[i + n_docs] >= m)) continue;
        const uint8_t c0 = bwt[i];
        bool differ = c0 <= 1;
        for (uint32_t t = 1; t < n_docs && !differ; ++t) differ = bwt[i + t] != c0;
        if (!differ) continue;
        uint32_t seen[kMaxDocs / 32];
        for (uint32_t w = 0; w < (n_docs + 31) / 32; ++w) seen[w] = 0;
        bool ok = true;
        for (uint32_t t = 0; t < n_docs; ++t) {
            const uint32_t d = upper_bound_u32(doc_start, n_docs, sa[i + t]) - 1;
            if (seen[d >> 5] >> (d & 31) & 1) { ok = false; break; }
            seen[d >> 5] |= 1u << (d & 31);
        }
        if (!ok) continue;
        flag[i] = 1;
        mum_len[i] = m;
    }
}

__global__ void gather_u32_kernel(const uint32_t *src, const uint32_t *at, uint64_t m, uint32_t *dst) {
    for (uint64_t j = blockIdx.x * (uint64_t)kTB + threadIdx.x; j < m; j += (uint64_t)gridDim.x * kTB) dst[j] = src[at[j]];
}

struct BudgetScope {
    DevBudget b;
    VmmScope keep_granules;   // freed arrays' memory stays with the process until the open returns (dev_vmm.h)
    DevBudget *prev;
    BudgetScope() : prev(current_budget()) { b.limit = env_budget_bytes(); current_budget() = &b; }
    ~BudgetScope() { current_budget() = prev; }
};

}  // namespace

int rlbwt_from_text(const uint8_t *text, uint64_t n, const uint64_t *doc_start, uint32_t n_docs, uint64_t min_mum,
                    int device, RlbwtResult &out, std::string &err) {
    if (!text || !doc_start || n < 2 || n_docs == 0) { err = "empty text or no documents"; return COLBWT_ERR_ARG; }
    if (n >= 0xffffffffull) { err = "text of " + std::to_string(n) + " characters: the suffix numbers are 32-bit"; return COLBWT_ERR_FORMAT; }
    if (n_docs > kMaxDocs) { err = "more than " + std::to_string(kMaxDocs) + " documents"; return COLBWT_ERR_FORMAT; }
    if (text[n - 1] != 0) { err = "the text must end with its only 0 byte"; return COLBWT_ERR_FORMAT; }
    LoadClock clock;
    std::vector<uint32_t> seps, docs(n_docs);
    for (uint64_t i = 0; i < n; ++i) {
        if (text[i] <= 1) seps.push_back((uint32_t)i);
        if (text[i] == 0 && i != n - 1) { err = "0 byte inside the text at " + std::to_string(i); return COLBWT_ERR_FORMAT; }
    }
    for (uint32_t d = 0; d < n_docs; ++d) {
        if (doc_start[d] >= n || (d ? doc_start[d] <= doc_start[d - 1] : doc_start[0] != 0)) {
            err = "document starts must ascend from 0";
            return COLBWT_ERR_ARG;
        }
        docs[d] = (uint32_t)doc_start[d];
    }
    if (min_mum == 0) min_mum = 1;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) { err = "no such HIP device"; return COLBWT_ERR_NO_DEVICE; }
    RB_TRY(hipSetDevice(device));
    BudgetScope budget;

    DevPtr d_text, d_key[2], d_sa[2], d_tmp, d_groups, d_scratch32;
    std::vector<DevPtr> levels;
    if (no(d_text.alloc(n)) || no(d_key[0].alloc(8 * n)) || no(d_key[1].alloc(8 * n)) || no(d_sa[0].alloc(4 * n)) || no(d_sa[1].alloc(4 * n)) ||
        no(d_scratch32.alloc(4 * n)) || no(d_groups.alloc(8))) {
        err = "out of device memory for the suffix sort of " + std::to_string(n) + " characters";
        return COLBWT_ERR_NOMEM;
    }
    clock.lap("rlbwt: text checked, buffers allocated");
    RB_TRY(hipMemcpy(d_text.get(), text, n, hipMemcpyHostToDevice));
    hipcub::DoubleBuffer<uint64_t> keys(d_key[0].as<uint64_t>(), d_key[1].as<uint64_t>());
    hipcub::DoubleBuffer<uint32_t> sa(d_sa[0].as<uint32_t>(), d_sa[1].as<uint32_t>());
    size_t tmp_bytes = 0, need = 0;
    RB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, sa, (size_t)n, 0, 64));
    RB_TRY(hipcub::DeviceScan::InclusiveScan(nullptr, need, d_scratch32.as<uint32_t>(), d_scratch32.as<uint32_t>(), hipcub::Max(), (size_t)n));
    tmp_bytes = std::max(tmp_bytes, need);
    RB_TRY(hipcub::DeviceSelect::Flagged(nullptr, need, hipcub::CountingInputIterator<uint32_t>(0), (uint8_t *)nullptr,
                                         (uint32_t *)nullptr, (unsigned long long *)nullptr, (size_t)n));
    tmp_bytes = std::max(tmp_bytes, need) + 256;
    if (no(d_tmp.alloc(tmp_bytes))) { err = "out of device memory (sort workspace)"; return COLBWT_ERR_NOMEM; }

    const unsigned grid = grid_for(n);
    const uint8_t *T = d_text.as<uint8_t>();     // raw pointers for the launches
    unsigned long long *d_count = d_groups.as<unsigned long long>();
    int bits = 1;
    while ((n >> bits) != 0) ++bits;          // values up to n fit
    {
        uint64_t *k0 = keys.Current();
        uint32_t *s0 = sa.Current();
        hipLaunchKernelGGL(first_keys_kernel, dim3(grid), dim3(kTB), 0, 0, T, n, k0, s0);
    }
    uint64_t h = 8;
    int end_bit = 64;
    out.rounds = 0;
    for (;;) {
        size_t tb = tmp_bytes;
        RB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp.get(), tb, keys, sa, (size_t)n, 0, end_bit));
        RB_TRY(hipMemsetAsync(d_groups.get(), 0, 8, 0));
        uint32_t *headpos = d_scratch32.as<uint32_t>();
        uint64_t *kc = keys.Current();
        const uint32_t *sc = sa.Current();
        hipLaunchKernelGGL(group_heads_kernel, dim3(grid), dim3(kTB), 0, 0, kc, n, headpos, d_count);
        tb = tmp_bytes;
        RB_TRY(hipcub::DeviceScan::InclusiveScan(d_tmp.get(), tb, headpos, headpos, hipcub::Max(), (size_t)n));
        DevPtr rank;
        if ((int)levels.size() >= kMaxLevels || no(rank.alloc(4 * n))) { err = "out of device memory for the rank array of round " + std::to_string(levels.size()); return COLBWT_ERR_NOMEM; }
        uint32_t *d_rank = rank.as<uint32_t>();
        hipLaunchKernelGGL(scatter_rank_kernel, dim3(grid), dim3(kTB), 0, 0, sc, headpos, n, d_rank);
        levels.push_back(std::move(rank));
        unsigned long long groups = 0;
        RB_TRY(hipMemcpy(&groups, d_groups.get(), 8, hipMemcpyDeviceToHost));
        ++out.rounds;
        if (groups == n) break;
        hipLaunchKernelGGL(pair_keys_kernel, dim3(grid), dim3(kTB), 0, 0, sc, d_rank, n, h, bits, kc);
        end_bit = 2 * bits;
        h *= 2;
    }
    clock.lap("rlbwt: suffix array");
    const uint32_t *d_sa_final = sa.Current();
    uint32_t *d_sa_spare = sa.Alternate();

    // LCP, BWT, runs
    uint32_t *d_lcp = d_scratch32.as<uint32_t>();
    Levels lv{};
    lv.count = (int)levels.size();
    for (int j = 0; j < lv.count; ++j) lv.rank[j] = levels[j].as<uint32_t>();
    hipLaunchKernelGGL(lcp_kernel, dim3(grid), dim3(kTB), 0, 0, T, d_sa_final, n, lv, d_lcp);
    // A pattern never holds a separator, so what lies behind one cannot decide between two runs:
    // thresholds and multi-MUMs both work on the LCPs cut at the first separator (all separators are
    // the same byte 1: the raw LCP of `T\1ACG..` and `T\1ACG..` runs on into the next records).
    DevPtr d_seps;
    if (no(d_seps.alloc(4 * seps.size()))) { err = "out of device memory"; return COLBWT_ERR_NOMEM; }
    RB_TRY(hipMemcpy(d_seps.get(), seps.data(), 4 * seps.size(), hipMemcpyHostToDevice));
    const uint32_t *p_seps = d_seps.as<uint32_t>();
    const uint32_t n_seps = (uint32_t)seps.size();
    hipLaunchKernelGGL(cap_lcp_kernel, dim3(grid), dim3(kTB), 0, 0, d_sa_final, p_seps, n_seps, n, d_lcp);
    RB_TRY(hipDeviceSynchronize());
    clock.lap("rlbwt: LCP");
    levels.clear();                                     // the rank arrays are no longer needed
    clock.lap("rlbwt:   rank arrays freed");
    uint8_t *d_bwt = reinterpret_cast<uint8_t *>(keys.Current());          // key buffers are free now: bwt | flags
    uint8_t *d_flag = d_bwt + n;
    uint32_t *d_start = reinterpret_cast<uint32_t *>(keys.Alternate());    // run starts, then run-sized arrays behind
    hipLaunchKernelGGL(bwt_kernel, dim3(grid), dim3(kTB), 0, 0, T, d_sa_final, n, d_bwt);
    hipLaunchKernelGGL(run_flags_kernel, dim3(grid), dim3(kTB), 0, 0, d_bwt, n, d_flag);
    size_t tb = tmp_bytes;
    RB_TRY(hipcub::DeviceSelect::Flagged(d_tmp.get(), tb, hipcub::CountingInputIterator<uint32_t>(0), d_flag, d_start,
                                         d_groups.as<unsigned long long>(), (size_t)n));
    unsigned long long r = 0;
    RB_TRY(hipMemcpy(&r, d_groups.get(), 8, hipMemcpyDeviceToHost));
    clock.lap("rlbwt:   BWT, run starts");

    // thresholds
    DevPtr d_head, d_head2, d_id, d_id2, d_thr;
    if (no(d_head.alloc(r)) || no(d_head2.alloc(r)) || no(d_id.alloc(4 * r)) || no(d_id2.alloc(4 * r)) || no(d_thr.alloc(4 * r))) {
        err = "out of device memory for " + std::to_string(r) + " runs";
        return COLBWT_ERR_NOMEM;
    }
    const unsigned rgrid = grid_for(r);
    uint8_t *p_head = d_head.as<uint8_t>();
    uint32_t *p_id = d_id.as<uint32_t>(), *p_id2 = d_id2.as<uint32_t>(), *p_thr = d_thr.as<uint32_t>();
    hipLaunchKernelGGL(run_chars_kernel, dim3(rgrid), dim3(kTB), 0, 0, d_bwt, d_start, (uint64_t)r, p_head, p_id);
    {
        size_t sb = 0;
        RB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sb, d_head.as<uint8_t>(), d_head2.as<uint8_t>(), d_id.as<uint32_t>(),
                                                  d_id2.as<uint32_t>(), (size_t)r, 0, 8));
        DevPtr extra;
        void *ws = d_tmp.get();
        if (sb > tmp_bytes) {
            if (no(extra.alloc(sb))) { err = "out of device memory (run sort workspace)"; return COLBWT_ERR_NOMEM; }
            ws = extra.get();
        }
        RB_TRY(hipcub::DeviceRadixSort::SortPairs(ws, sb, d_head.as<uint8_t>(), d_head2.as<uint8_t>(), d_id.as<uint32_t>(),
                                                  d_id2.as<uint32_t>(), (size_t)r, 0, 8));
        RB_TRY(hipDeviceSynchronize());
    }
    clock.lap("rlbwt:   runs sorted by character");
    const uint64_t n_thr_blocks = (n + kThrBlock - 1) / kThrBlock;
    DevPtr d_block_min;
    if (no(d_block_min.alloc(8 * n_thr_blocks))) { err = "out of device memory"; return COLBWT_ERR_NOMEM; }
    uint64_t *p_block_min = d_block_min.as<uint64_t>();
    hipLaunchKernelGGL(block_min_kernel, dim3(grid_for(n_thr_blocks * 64)), dim3(kTB), 0, 0, d_lcp, n, n_thr_blocks, p_block_min);
    hipLaunchKernelGGL(threshold_kernel, dim3(grid_for(r * 64)), dim3(kTB), 0, 0, p_id2, p_head, d_start, d_lcp, p_block_min, (uint64_t)r,
                       p_thr);
    RB_TRY(hipDeviceSynchronize());

    clock.lap("rlbwt: runs, thresholds");
    out.n = n;
    out.heads.resize(r);
    out.lens.resize(r);
    out.thr.resize(r);
    {
        std::vector<uint32_t> st(r), th(r);
        RB_TRY(hipMemcpy(out.heads.data(), d_head.get(), r, hipMemcpyDeviceToHost));
        RB_TRY(hipMemcpy(st.data(), d_start, 4 * r, hipMemcpyDeviceToHost));
        RB_TRY(hipMemcpy(th.data(), d_thr.get(), 4 * r, hipMemcpyDeviceToHost));
        for (uint64_t j = 0; j < r; ++j) {
            out.lens[j] = (j + 1 < r ? st[j + 1] : n) - st[j];
            out.thr[j] = th[j];
        }
    }

    clock.lap("rlbwt: runs copied back");
    // multi-MUMs
    out.mum_len.clear();
    out.mum_pos.clear();
    if (n_docs >= 2) {
        DevPtr d_docs, d_pre, d_suf;
        if (no(d_docs.alloc(4 * n_docs)) || no(d_pre.alloc(4 * n)) || no(d_suf.alloc(4 * n))) { err = "out of device memory"; return COLBWT_ERR_NOMEM; }
        RB_TRY(hipMemcpy(d_docs.get(), docs.data(), 4 * n_docs, hipMemcpyHostToDevice));
        const uint32_t *p_docs = d_docs.as<uint32_t>();
        const uint32_t min_len = (uint32_t)std::min<uint64_t>(min_mum, 0xffffffffu);
        uint32_t *p_pre = d_pre.as<uint32_t>(), *p_suf = d_suf.as<uint32_t>();
        const uint32_t window = n_docs - 1;
        hipLaunchKernelGGL(window_min_kernel, dim3(grid_for((n + window - 1) / window)), dim3(kTB), 0, 0, d_lcp, n, window, p_pre, p_suf);
        uint32_t *d_mlen = d_sa_spare;
        hipLaunchKernelGGL(mum_kernel, dim3(grid), dim3(kTB), 0, 0, d_sa_final, d_lcp, p_pre, p_suf, d_bwt, p_docs, n_docs, n, min_len, d_flag,
                           d_mlen);
        tb = tmp_bytes;
        uint32_t *d_pos = d_start;                     // run starts are on the host already
        RB_TRY(hipcub::DeviceSelect::Flagged(d_tmp.get(), tb, hipcub::CountingInputIterator<uint32_t>(0), d_flag, d_pos,
                                             d_groups.as<unsigned long long>(), (size_t)n));
        unsigned long long m = 0;
        RB_TRY(hipMemcpy(&m, d_groups.get(), 8, hipMemcpyDeviceToHost));
        if (m) {
            DevPtr d_len;
            if (no(d_len.alloc(4 * m))) { err = "out of device memory"; return COLBWT_ERR_NOMEM; }
            uint32_t *p_len = d_len.as<uint32_t>();
            hipLaunchKernelGGL(gather_u32_kernel, dim3(grid_for(m)), dim3(kTB), 0, 0, d_mlen, d_pos, (uint64_t)m, p_len);
            std::vector<uint32_t> p(m), l(m);
            RB_TRY(hipMemcpy(p.data(), d_pos, 4 * m, hipMemcpyDeviceToHost));
            RB_TRY(hipMemcpy(l.data(), d_len.get(), 4 * m, hipMemcpyDeviceToHost));
            out.mum_pos.assign(p.begin(), p.end());
            out.mum_len.assign(l.begin(), l.end());
        }
    }
    RB_TRY(hipDeviceSynchronize());
    RB_TRY(hipGetLastError());
    clock.lap("rlbwt: multi-MUMs");
    return COLBWT_OK;
}

}  // namespace colbwt

// ---- C-ABI (include/colbwt.h) ------------------------------------------------
struct colbwt_rlbwt {
    colbwt::RlbwtResult res;
    uint32_t n_docs = 0;
};

namespace {
thread_local std::string g_rlbwt_err;
}

extern "C" const char *colbwt_rlbwt_error(void) { return g_rlbwt_err.c_str(); }

extern "C" int colbwt_rlbwt_build_text(const uint8_t *text, uint64_t n, const uint64_t *doc_start, uint32_t n_docs,
                                       uint64_t min_mum, int device, colbwt_rlbwt **out) {
    if (!out) { g_rlbwt_err = "null argument"; return COLBWT_ERR_ARG; }
    *out = nullptr;
    colbwt_rlbwt *h = new colbwt_rlbwt;
    h->n_docs = n_docs;
    const int rc = colbwt::rlbwt_from_text(text, n, doc_start, n_docs, min_mum, device, h->res, g_rlbwt_err);
    if (rc != COLBWT_OK) { delete h; return rc; }
    *out = h;
    return COLBWT_OK;
}

extern "C" int colbwt_rlbwt_build_files(const char *const *fastas, uint32_t n_files, int revcomp, uint64_t min_mum, int device,
                                        const char *out_prefix, colbwt_rlbwt **out) {
    if (out) *out = nullptr;
    if (!fastas || !n_files || (!out_prefix && !out)) { g_rlbwt_err = "null argument"; return COLBWT_ERR_ARG; }
    std::vector<std::string> paths;
    for (uint32_t i = 0; i < n_files; ++i) {
        if (!fastas[i]) { g_rlbwt_err = "null path"; return COLBWT_ERR_ARG; }
        paths.push_back(fastas[i]);
    }
    std::vector<uint8_t> text;
    std::vector<uint64_t> doc_start;
    colbwt::LoadClock clock;
    if (!colbwt::text_from_fastas(paths, revcomp != 0, text, doc_start, g_rlbwt_err)) return COLBWT_ERR_IO;
    clock.lap("rlbwt: FASTA files read");
    colbwt_rlbwt *h = new colbwt_rlbwt;
    h->n_docs = n_files;
    int rc = colbwt::rlbwt_from_text(text.data(), text.size(), doc_start.data(), n_files, min_mum, device, h->res, g_rlbwt_err);
    clock.lap("rlbwt: construction");
    if (rc == COLBWT_OK && out_prefix && !colbwt::write_rlbwt_files(out_prefix, h->res, n_files, g_rlbwt_err)) rc = COLBWT_ERR_IO;
    clock.lap("rlbwt: files written");
    if (rc != COLBWT_OK || !out) delete h; else *out = h;
    return rc;
}

extern "C" void colbwt_rlbwt_get(const colbwt_rlbwt *h, colbwt_rlbwt_view *v) {
    if (!h || !v) return;
    v->n = h->res.n;
    v->n_runs = h->res.heads.size();
    v->n_mums = h->res.mum_len.size();
    v->n_docs = h->n_docs;
    v->rounds = h->res.rounds;
    v->heads = h->res.heads.data();
    v->lens = h->res.lens.data();
    v->thr_pos = h->res.thr.data();
    v->mum_len = h->res.mum_len.data();
    v->mum_pos = h->res.mum_pos.data();
}

extern "C" void colbwt_rlbwt_free(colbwt_rlbwt *h) { delete h; }
