// col_split.hip -- the sub-run splitter (SURVEY.md 8(f) "next" #2): multi-MUMs -> `.col_runs`
// (where sub-runs start) + `.col_ids` (their chain statistic), the inputs of the index builder.
//
// Reference: build_FL (src/build_FL.cpp:27-74 -> FL_table, include/ds/FL_table.hpp:82-130,
// 343-391) followed by col_split (src/col_split.cpp:62-140 -> include/col_split.hpp:54-136 split,
// :226-247 FL_range, :258-372 find_col_runs, :138-157 + :374-390 writers).  Three parts:
//
//   host    FL table from .bwt.heads / .bwt.len (the reference's `.FL_table` file embeds an sdsl
//           sd_vector, so the table is rebuilt rather than read);
//   device  FL-stepping of every multi-MUM -- the reference steps each of them twice, one after
//           the other (col_split.hpp:111-134).  A MUM of length m on N documents is N * m
//           dependent random reads of a 14-byte row: the same move-table walk as the query, one
//           lane per MUM in `tunnels` mode (the N rows move as one range until they diverge),
//           one workgroup per MUM with one lane per document in `all` mode (lanes track where
//           neighbouring rows fall into different F runs: those are the pieces of FL_range).
//           Every mark (position, id, height) is an atomicMax on a dense per-position array whose
//           key order is the reference's overwrite rule: the last MUM wins (tunnels, :126-128),
//           the tallest piece and then the first MUM wins (all, :118-124).  Marks of one MUM never
//           meet (its rows walk distinct text positions), so the MUM number orders the writers;
//   host    the marked positions, compacted in order on the device, go through the overlap sweep
//           of find_col_runs (inherently sequential: a min-heap of open intervals).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/colbwt.h"
#include "dev_mem.h"
#include "index.h"

namespace colbwt {

namespace {

struct FLDev {
    const uint64_t *idx;       // r + 1 : first F position of every F run (idx[r] = n)
    const uint32_t *interval;  // r     : F run holding FL(first position of the run)
    const uint16_t *offset;    // r     : its offset there (the 16 bits the reference's bit-field keeps)
    uint64_t n;
    uint32_t r;
};

__device__ __forceinline__ uint64_t fl_len(const FLDev &T, uint32_t i) { return T.idx[(uint64_t)i + 1] - T.idx[i]; }

// FL_table::FL (FL_table.hpp:227-238)
__device__ __forceinline__ void fl_step(const FLDev &T, uint32_t &run, uint64_t &off) {
    uint32_t ni = T.interval[run];
    uint64_t no = (uint64_t)T.offset[run] + off;
    uint64_t len = fl_len(T, ni);
    while (no >= len && ni < T.r - 1) {
        no -= len;
        ++ni;
        len = fl_len(T, ni);
    }
    run = ni;
    off = no;
}

// F run holding F position p
__device__ __forceinline__ uint32_t fl_run_of(const FLDev &T, uint64_t p) {
    uint64_t lo = 0, hi = T.r;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (T.idx[mid] <= p) lo = mid; else hi = mid;
    }
    return (uint32_t)lo;
}

// `tunnels` (Options::Mode::Tunneled): one lane per multi-MUM.  The N rows stay one range
// while offset + N <= len(run) (FL_range returns one piece, col_split.hpp:230-243); the walk ends
// the first time it would not (skip_non_tunnel, :81 / :99).  best[pos] = 1 + the last MUM marking pos.
__global__ __launch_bounds__(256) void tunnel_kernel(FLDev T, const uint64_t *__restrict__ col_len, const uint64_t *__restrict__ col_pos,
                                                     uint64_t n_cols, uint32_t N, uint32_t rate, uint32_t *__restrict__ best) {
    const uint64_t m = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= n_cols) return;
    const uint64_t pos = col_pos[m];
    uint32_t run = fl_run_of(T, pos);
    uint64_t off = pos - T.idx[run];
    if (off + N > fl_len(T, run)) return;                      // :79-81 more than one piece at once
    fl_step(T, run, off);
    const uint64_t len = col_len[m];
    for (uint64_t j = 0; j < len; ++j) {
        if (j % rate == 0) atomicMax(&best[T.idx[run] + off], (uint32_t)m + 1u);   // :87-93, :126-128
        if (off + N > fl_len(T, run)) break;                   // :95-99 the next FL_range diverges
        fl_step(T, run, off);
    }
}

// `all` (Options::Mode::All): one workgroup per multi-MUM, lane d = document row d.  A lane
// walks its own row and the row above it; `cut` says the two have been in different F runs
// before some step so far -- i.e. row d starts a piece of the FL_range recursion (:83-98).  At a
// marking step the piece starts mark their position with the piece's height (rows to the next
// start).  best[pos] = height << 32 | ~MUM: the tallest piece wins, then the first MUM (:118-124).
__global__ __launch_bounds__(1024) void all_kernel(FLDev T, const uint64_t *__restrict__ col_len, const uint64_t *__restrict__ col_pos,
                                                   uint64_t n_cols, uint32_t N, uint32_t rate, unsigned long long *__restrict__ best) {
    __shared__ uint8_t s_start[1024];
    const uint32_t d = threadIdx.x;
    for (uint64_t m = blockIdx.x; m < n_cols; m += gridDim.x) {
        const bool row = d < N;
        uint32_t run = 0, prun = 0;
        uint64_t off = 0, poff = 0;
        bool cut = d == 0;
        if (row) {
            const uint64_t p = col_pos[m] + d;
            run = fl_run_of(T, p < T.n ? p : T.n - 1);
            off = p - T.idx[run];
            if (d > 0) {
                prun = fl_run_of(T, p - 1);
                poff = p - 1 - T.idx[prun];
            }
        }
        const uint64_t len = col_len[m];
        auto advance = [&] {                                   // one FL_range application to every piece
            if (row) {
                if (d > 0) {
                    cut = cut || run != prun;                  // different F runs: the range splits here (:232-243)
                    fl_step(T, prun, poff);
                }
                fl_step(T, run, off);
            }
        };
        advance();                                             // :79
        for (uint64_t j = 0; j < len; ++j) {
            if (j % rate == 0) {                               // :87-93
                __syncthreads();
                s_start[d] = row && cut;
                __syncthreads();
                if (row && cut) {
                    uint32_t h = 1;
                    while (d + h < N && !s_start[d + h]) ++h;
                    const unsigned long long key = ((unsigned long long)h << 32) | (0xFFFFFFFFull - (unsigned long long)m);
                    atomicMax(&best[T.idx[run] + off], key);
                }
            }
            advance();                                         // :95-96
        }
        __syncthreads();
    }
}

// Marked positions in ascending order: 1024 positions per workgroup, count / scan / scatter.
template <typename KeyT>
__global__ __launch_bounds__(256) void count_marks_kernel(const KeyT *__restrict__ best, uint64_t n, uint32_t *__restrict__ tile_count) {
    __shared__ uint32_t s_sum[256];
    const uint64_t base = (uint64_t)blockIdx.x * 1024 + (uint64_t)threadIdx.x * 4;
    uint32_t c = 0;
    for (int q = 0; q < 4; ++q) c += base + q < n && best[base + q] != 0;
    s_sum[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) s_sum[threadIdx.x] += s_sum[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_count[blockIdx.x] = s_sum[0];
}

template <typename KeyT>
__global__ __launch_bounds__(256) void emit_marks_kernel(const KeyT *__restrict__ best, uint64_t n, const uint64_t *__restrict__ tile_first,
                                                         uint32_t N, uint64_t *__restrict__ out_pos, uint32_t *__restrict__ out_mum,
                                                         uint16_t *__restrict__ out_height) {
    __shared__ uint32_t s_sum[256];
    const uint64_t base = (uint64_t)blockIdx.x * 1024 + (uint64_t)threadIdx.x * 4;
    uint32_t c = 0;
    for (int q = 0; q < 4; ++q) c += base + q < n && best[base + q] != 0;
    s_sum[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t dd = 1; dd < 256; dd <<= 1) {                // inclusive scan of the per-thread counts
        const uint32_t add = threadIdx.x >= dd ? s_sum[threadIdx.x - dd] : 0;
        __syncthreads();
        s_sum[threadIdx.x] += add;
        __syncthreads();
    }
    uint64_t at = tile_first[blockIdx.x] + s_sum[threadIdx.x] - c;
    for (int q = 0; q < 4; ++q) {
        if (base + q >= n) break;
        const KeyT k = best[base + q];
        if (k == 0) continue;
        out_pos[at] = base + q;
        if constexpr (sizeof(KeyT) == 4) {
            out_mum[at] = (uint32_t)k - 1u;
            out_height[at] = (uint16_t)N;
        } else {
            out_mum[at] = (uint32_t)(0xFFFFFFFFull - ((unsigned long long)k & 0xFFFFFFFFull));
            out_height[at] = (uint16_t)((unsigned long long)k >> 32);
        }
        ++at;
    }
}

#define CS_TRY(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                   \
            (void)hipGetLastError();                                                   \
            return e_ == hipErrorOutOfMemory ? COLBWT_ERR_NOMEM : COLBWT_ERR_HIP;      \
        }                                                                              \
    } while (0)

struct FLHost {
    uint64_t n = 0;
    std::vector<uint64_t> idx, L_head;       // r + 1 each
    std::vector<uint32_t> interval;
    std::vector<uint16_t> offset;
};

// FL_table(heads, lengths): FL_table.hpp:82-130, compute_table :343-376, compute_L_heads :378-391.
bool build_fl(const uint8_t *heads, uint64_t n_heads, const uint64_t *lens, FLHost &t) {
    uint64_t r = 0;
    while (r < n_heads && heads[r] != 0xFF) ++r;               // `char c = heads.get()` compared with EOF (:99)
    if (r == 0 || r > 0xFFFFFFFEull) return false;
    std::vector<uint8_t> ch(r);
    uint64_t count[257] = {0};
    t.L_head.resize(r + 1);
    uint64_t n = 0;
    for (uint64_t i = 0; i < r; ++i) {
        const int c = (int)(signed char)heads[i];
        ch[i] = (uint8_t)(c <= 1 ? 1 : c);                     // :102 (signed: bytes >= 0x80 fold too)
        t.L_head[i] = n;
        n += lens[i];
        ++count[ch[i] + 1];
    }
    t.L_head[r] = n;
    t.n = n;
    for (int c = 0; c < 256; ++c) count[c + 1] += count[c];    // first F row of every character
    // F order = characters ascending, runs of a character in L order (:345-357)
    std::vector<uint64_t> cursor(count, count + 256);
    std::vector<uint32_t> L_of(r);
    std::vector<uint64_t> len_F(r);
    for (uint64_t i = 0; i < r; ++i) {
        const uint64_t k = cursor[ch[i]]++;
        L_of[k] = (uint32_t)i;
        len_F[k] = lens[i];
    }
    t.idx.resize(r + 1);
    uint64_t at = 0;
    for (uint64_t k = 0; k < r; ++k) {
        t.idx[k] = at;
        at += len_F[k];
    }
    t.idx[r] = n;
    t.interval.resize(r);
    t.offset.resize(r);
    for (int c = 0; c < 256; ++c) {                            // :359-375 L and F scanned in step per character
        uint64_t F_curr = 0;
        for (uint64_t k = count[c]; k < count[c + 1]; ++k) {
            const uint64_t L_seen = t.L_head[L_of[k]];
            while (t.idx[F_curr + 1] <= L_seen) ++F_curr;
            t.interval[k] = (uint32_t)F_curr;
            t.offset[k] = (uint16_t)(L_seen - t.idx[F_curr]);  // `ulint offset : LEN_BITS`
        }
    }
    return true;
}

uint8_t bin_id(uint64_t id) { return (uint8_t)(id >= 256 ? id % 255 + 1 : id); }   // col_split.hpp:222-224

// What col_split::find_col_runs (col_split.hpp:258-342) computes, derived here as two passes
// over sorted arrays instead of the reference's heap of open intervals.
//
// A marked position k opens the interval [pos[k], pos[k] + heights[k]) carrying ids[k].  The
// reference's rule for which id is "in force" along the BWT only ever looks at moments where at
// most one interval is open, so all it needs of the open set is its SIZE and, when that is one,
// WHICH interval it is -- a counter and the XOR of the open intervals' numbers give both:
//
//   pass 1 (col events, ascending):  intervals are closed in order of (end, start), always before
//     the next one opens; a close at x with exactly one interval left that reaches beyond x hands
//     x to that interval's id, a close that leaves none open hands x to id 0 unless an interval
//     opens at x itself (or x is the end of the BWT); an interval that opens alone takes its start
//     for its id when the id is non-zero.
//   pass 2 (merge with the BWT run heads):  every run head becomes a sub-run start carrying the id
//     in force there; a head that coincides with a col event is represented by the event.
void find_col_runs(const FLHost &t, const std::vector<uint64_t> &pos, const std::vector<uint8_t> &ids,
                   const std::vector<uint16_t> &heights, std::vector<uint64_t> &split_pos, std::vector<uint8_t> &split_ids) {
    split_pos.clear();
    split_ids.clear();
    const size_t m = pos.size();
    if (m == 0) return;                                        // nothing marked: no files' worth of output (:259-261)
    const uint64_t n = t.n, r = t.idx.size() - 1;

    // ---- pass 1
    std::vector<uint64_t> ends(m);
    for (size_t k = 0; k < m; ++k) ends[k] = pos[k] + heights[k];
    std::vector<uint64_t> by_end(m);                           // interval numbers by (end, start); starts ascend with k
    for (size_t k = 0; k < m; ++k) by_end[k] = k;
    std::stable_sort(by_end.begin(), by_end.end(), [&](uint64_t a, uint64_t b) { return ends[a] < ends[b]; });
    std::vector<std::pair<uint64_t, uint8_t>> events;          // (position, id taking over there)
    events.reserve(2 * m);
    size_t n_open = 0, closed = 0;
    uint64_t open_xor = 0;                                     // XOR of the open intervals' numbers
    auto close_up_to = [&](uint64_t x, size_t opened) {        // every OPENED interval ending at or before x
        while (closed < m && ends[by_end[closed]] <= x && by_end[closed] < opened) {
            const uint64_t e = by_end[closed++];
            --n_open;
            open_xor ^= e;
            if (n_open == 1 && ends[open_xor] > ends[e]) events.emplace_back(ends[e], ids[open_xor]);
            else if (n_open == 0 && ends[e] < x) events.emplace_back(ends[e], (uint8_t)0);
        }
    };
    for (size_t k = 0; k < m; ++k) {
        close_up_to(pos[k], k);
        ++n_open;
        open_xor ^= k;
        if (n_open == 1 && ids[k] > 0) events.emplace_back(pos[k], ids[k]);
    }
    close_up_to(n, m);

    // ---- pass 2
    split_pos.reserve(events.size() + r);
    split_ids.reserve(events.size() + r);
    uint64_t head = 0;                                         // next BWT run not yet written
    uint8_t in_force = 0;
    auto heads_before = [&](uint64_t x) {
        for (; head < r && t.L_head[head] < x; ++head) {
            split_pos.push_back(t.L_head[head]);
            split_ids.push_back(in_force);
        }
    };
    for (const auto &ev : events) {
        heads_before(ev.first);
        if (head < r && t.L_head[head] == ev.first) ++head;
        in_force = ev.second;
        split_pos.push_back(ev.first);
        split_ids.push_back(ev.second);
    }
    heads_before(n);
}

int col_split_run(const uint8_t *heads, uint64_t n_heads, const uint64_t *lens, const uint64_t *mum_len, const uint64_t *mum_pos,
                  uint64_t n_mums, uint32_t num_docs, int mode, int split_rate, int device, std::vector<uint64_t> &split_pos,
                  std::vector<uint8_t> &split_ids, uint64_t &n_out, std::string &err) {
    if (mode != COLBWT_SPLIT_TUNNELS && mode != COLBWT_SPLIT_ALL) { err = "mode must be COLBWT_SPLIT_TUNNELS or COLBWT_SPLIT_ALL"; return COLBWT_ERR_ARG; }
    if (split_rate < 1) { err = "split_rate must be >= 1"; return COLBWT_ERR_ARG; }
    const uint32_t N = num_docs & 0xFFFFu;                     // split(..., len_t N, ...): 16 bits (col_split.hpp:56)
    if (mode == COLBWT_SPLIT_ALL && N > 1024) { err = "`all` mode handles up to 1024 documents on the device"; return COLBWT_ERR_ARG; }
    FLHost t;
    if (!build_fl(heads, n_heads, lens, t)) { err = "empty or oversized .bwt.heads"; return COLBWT_ERR_FORMAT; }
    n_out = t.n;
    const uint64_t n = t.n, r = t.idx.size() - 1;
    // FL_loop (:69-109) consumes the MUMs in list order while their positions fall into the run
    // being visited: a position that goes backwards (or beyond n) stalls it for good.
    uint64_t used = 0;
    {
        uint64_t run_prev = 0;
        for (; used < n_mums; ++used) {
            if (mum_pos[used] >= n) break;
            const uint64_t run = (uint64_t)(std::upper_bound(t.idx.begin(), t.idx.begin() + r + 1, mum_pos[used]) - t.idx.begin()) - 1;
            if (run < run_prev) break;
            run_prev = run;
        }
    }
    std::vector<uint64_t> pos;
    std::vector<uint8_t> ids;
    std::vector<uint16_t> heights;
    if (used > 0 && N > 0) {
        int rc = select_device(device, err);
        if (rc != COLBWT_OK) return rc;
        if (used > 0xFFFFFFFEull) { err = "more than 2^32-2 multi-MUMs"; return COLBWT_ERR_ARG; }
        DevPtr d_idx, d_itv, d_off, d_len, d_pos, d_best, d_tiles, d_first, d_opos, d_omum, d_oh;
        CS_TRY(d_idx.alloc((r + 1) * 8));
        CS_TRY(d_itv.alloc(r * 4));
        CS_TRY(d_off.alloc(r * 2 + 2));
        CS_TRY(d_len.alloc(used * 8));
        CS_TRY(d_pos.alloc(used * 8));
        CS_TRY(hipMemcpy(d_idx.get(), t.idx.data(), (r + 1) * 8, hipMemcpyHostToDevice));
        CS_TRY(hipMemcpy(d_itv.get(), t.interval.data(), r * 4, hipMemcpyHostToDevice));
        CS_TRY(hipMemcpy(d_off.get(), t.offset.data(), r * 2, hipMemcpyHostToDevice));
        CS_TRY(hipMemcpy(d_len.get(), mum_len, used * 8, hipMemcpyHostToDevice));
        CS_TRY(hipMemcpy(d_pos.get(), mum_pos, used * 8, hipMemcpyHostToDevice));
        const FLDev T{d_idx.as<uint64_t>(), d_itv.as<uint32_t>(), d_off.as<uint16_t>(), n, (uint32_t)r};
        const uint64_t key_bytes = mode == COLBWT_SPLIT_ALL ? 8 : 4;
        CS_TRY(d_best.alloc(n * key_bytes));                   // dense: one key per BWT position
        CS_TRY(hipMemset(d_best.get(), 0, n * key_bytes));
        const uint64_t *dl = d_len.as<uint64_t>(), *dp = d_pos.as<uint64_t>();
        if (mode == COLBWT_SPLIT_TUNNELS) {
            uint32_t *best = d_best.as<uint32_t>();
            hipLaunchKernelGGL(tunnel_kernel, dim3((uint32_t)((used + 255) / 256)), dim3(256), 0, 0, T, dl, dp, used, N,
                               (uint32_t)split_rate, best);
        } else {
            unsigned long long *best = d_best.as<unsigned long long>();
            const uint32_t threads = std::max(64u, (N + 63u) & ~63u);
            hipLaunchKernelGGL(all_kernel, dim3((uint32_t)std::min<uint64_t>(used, 4096)), dim3(threads), 0, 0, T, dl, dp, used, N,
                               (uint32_t)split_rate, best);
        }
        CS_TRY(hipGetLastError());
        CS_TRY(hipStreamSynchronize(0));
        const uint64_t tiles = (n + 1023) / 1024;
        CS_TRY(d_tiles.alloc(tiles * 4));
        uint32_t *tile_count = d_tiles.as<uint32_t>();
        const uint32_t *const best32 = d_best.as<uint32_t>();
        const unsigned long long *const best64 = d_best.as<unsigned long long>();
        if (mode == COLBWT_SPLIT_TUNNELS)
            hipLaunchKernelGGL(count_marks_kernel<uint32_t>, dim3((uint32_t)tiles), dim3(256), 0, 0, best32, n, tile_count);
        else
            hipLaunchKernelGGL(count_marks_kernel<unsigned long long>, dim3((uint32_t)tiles), dim3(256), 0, 0, best64, n, tile_count);
        CS_TRY(hipGetLastError());
        std::vector<uint32_t> h_count(tiles);
        CS_TRY(hipMemcpy(h_count.data(), tile_count, tiles * 4, hipMemcpyDeviceToHost));
        std::vector<uint64_t> h_first(tiles + 1, 0);
        for (uint64_t b = 0; b < tiles; ++b) h_first[b + 1] = h_first[b] + h_count[b];
        const uint64_t marks = h_first[tiles];
        if (marks) {
            CS_TRY(d_first.alloc(tiles * 8));
            CS_TRY(hipMemcpy(d_first.get(), h_first.data(), tiles * 8, hipMemcpyHostToDevice));
            CS_TRY(d_opos.alloc(marks * 8));
            CS_TRY(d_omum.alloc(marks * 4));
            CS_TRY(d_oh.alloc(marks * 2));
            const uint64_t *tf = d_first.as<uint64_t>();
            uint64_t *op = d_opos.as<uint64_t>();
            uint32_t *om = d_omum.as<uint32_t>();
            uint16_t *oh = d_oh.as<uint16_t>();
            if (mode == COLBWT_SPLIT_TUNNELS)
                hipLaunchKernelGGL(emit_marks_kernel<uint32_t>, dim3((uint32_t)tiles), dim3(256), 0, 0, best32, n, tf, N, op, om, oh);
            else
                hipLaunchKernelGGL(emit_marks_kernel<unsigned long long>, dim3((uint32_t)tiles), dim3(256), 0, 0, best64, n, tf, N, op, om,
                                   oh);
            CS_TRY(hipGetLastError());
            pos.resize(marks);
            heights.resize(marks);
            std::vector<uint32_t> mum(marks);
            CS_TRY(hipMemcpy(pos.data(), op, marks * 8, hipMemcpyDeviceToHost));
            CS_TRY(hipMemcpy(mum.data(), om, marks * 4, hipMemcpyDeviceToHost));
            CS_TRY(hipMemcpy(heights.data(), oh, marks * 2, hipMemcpyDeviceToHost));
            ids.resize(marks);
            for (uint64_t k = 0; k < marks; ++k) ids[k] = bin_id((uint64_t)mum[k] + 1);   // c_id starts at 1 (:71)
        }
    }
    find_col_runs(t, pos, ids, heights, split_pos, split_ids);
    return COLBWT_OK;
}

thread_local std::string g_split_err;

bool read_all(const std::string &path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(sz > 0 ? (size_t)sz : 0);
    const bool ok = sz <= 0 || fread(out.data(), 1, (size_t)sz, f) == (size_t)sz;
    fclose(f);
    return ok;
}

uint64_t le5(const uint8_t *p) {
    uint64_t v = 0;
    memcpy(&v, p, 5);
    return v;
}

}  // namespace

}  // namespace colbwt

using namespace colbwt;

extern "C" const char *colbwt_col_split_error(void) { return g_split_err.c_str(); }

extern "C" int colbwt_col_split_arrays(const uint8_t *heads, uint64_t n_heads, const uint64_t *lens, const uint64_t *mum_len,
                                       const uint64_t *mum_pos, uint64_t n_mums, uint32_t num_docs, int mode, int split_rate,
                                       int device, uint64_t *split_pos, uint64_t cap, uint64_t *n_split, uint8_t *col_ids,
                                       uint64_t *bwt_len) {
    if (!heads || !lens || (n_mums && (!mum_len || !mum_pos)) || !n_split) {
        g_split_err = "null argument";
        return COLBWT_ERR_ARG;
    }
    std::vector<uint64_t> sp;
    std::vector<uint8_t> si;
    uint64_t n = 0;
    const int rc = col_split_run(heads, n_heads, lens, mum_len, mum_pos, n_mums, num_docs, mode, split_rate, device, sp, si, n, g_split_err);
    if (rc != COLBWT_OK) return rc;
    *n_split = sp.size();
    if (bwt_len) *bwt_len = n;
    if (sp.size() > cap || (sp.size() && (!split_pos || !col_ids))) {
        g_split_err = "output arrays too small: " + std::to_string(sp.size()) + " sub-run starts";
        return COLBWT_ERR_ARG;
    }
    if (!sp.empty()) {
        memcpy(split_pos, sp.data(), sp.size() * 8);
        memcpy(col_ids, si.data(), si.size());
    }
    return COLBWT_OK;
}

// col_split <prefix> -m <mode> -s <rate> (src/col_split.cpp:62-140) with build_FL folded in:
// reads <prefix>.bwt.heads, <prefix>.bwt.len, <prefix>.col_mums (5-byte num_docs, then 5-byte
// (length, position) pairs, :90-106); writes <prefix>.col_runs (bit_vector::serialize, col_split.hpp:
// 384-386: u64 length in bits + words -- sdsl's layout as SURVEY.md Appendix A records it,
// unverified against sdsl) and <prefix>.col_ids (one byte per sub-run start, :146-155).
extern "C" int colbwt_col_split(const char *prefix, int mode, int split_rate, int device) {
    if (!prefix) {
        g_split_err = "null prefix";
        return COLBWT_ERR_ARG;
    }
    const std::string p = prefix;
    std::vector<uint8_t> heads, len_raw, mums;
    if (!read_all(p + ".bwt.heads", heads) || !read_all(p + ".bwt.len", len_raw) || !read_all(p + ".col_mums", mums)) {
        g_split_err = "cannot read " + p + ".bwt.heads / .bwt.len / .col_mums";
        return COLBWT_ERR_IO;
    }
    std::vector<uint64_t> lens(heads.size(), 0);
    for (size_t i = 0; i < heads.size() && 5 * (i + 1) <= len_raw.size(); ++i) lens[i] = le5(len_raw.data() + 5 * i);
    const uint64_t file_values = mums.size() / 5;                          // :91-92
    if (file_values < 1) {
        g_split_err = p + ".col_mums holds no header";
        return COLBWT_ERR_FORMAT;
    }
    const uint64_t num_mums = (file_values - 1) / 2, num_docs = le5(mums.data());
    std::vector<uint64_t> mlen(num_mums), mpos(num_mums);
    for (uint64_t i = 0; i < num_mums; ++i) {                              // :101-104
        mlen[i] = le5(mums.data() + 5 * (1 + 2 * i));
        mpos[i] = le5(mums.data() + 5 * (2 + 2 * i));
    }
    std::vector<uint64_t> sp;
    std::vector<uint8_t> si;
    uint64_t n = 0;
    const int rc = col_split_run(heads.data(), heads.size(), lens.data(), mlen.data(), mpos.data(), num_mums, (uint32_t)num_docs, mode,
                                 split_rate, device, sp, si, n, g_split_err);
    if (rc != COLBWT_OK) return rc;
    std::vector<uint64_t> words((n + 63) / 64, 0);
    for (uint64_t q : sp) words[q >> 6] |= 1ull << (q & 63);
    FILE *f = fopen((p + ".col_runs").c_str(), "wb");
    bool ok = f != nullptr;
    if (ok) {
        ok = fwrite(&n, 8, 1, f) == 1 && (words.empty() || fwrite(words.data(), 8, words.size(), f) == words.size());
        ok = fclose(f) == 0 && ok;
    }
    f = ok ? fopen((p + ".col_ids").c_str(), "wb") : nullptr;
    ok = ok && f != nullptr;
    if (ok) {
        ok = si.empty() || fwrite(si.data(), 1, si.size(), f) == si.size();   // ids are binned already (:149-152 is idempotent)
        ok = fclose(f) == 0 && ok;
    }
    if (!ok) {
        g_split_err = "cannot write " + p + ".col_runs / .col_ids";
        return COLBWT_ERR_IO;
    }
    return COLBWT_OK;
}
