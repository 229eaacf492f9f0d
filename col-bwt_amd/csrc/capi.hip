// capi.hip -- the C-ABI of include/colbwt.h over the HIP engine.
// Host-side mirror of col_pml (col_bwt.hpp:386-575) and of pml_query's main
// (pml_query.cpp:92-143).  No CPU fallback anywhere: every entry point that
// computes PML/col-ids needs a HIP device and says so when there is none.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/colbwt.h"
#include "bin_writer.h"
#include "fasta_parallel.h"
#include "fastx_reader.h"
#include "index.h"
#include "query_kernels.h"
#include "text_writer.h"

using namespace colbwt;

#define COLBWT_LAYOUT_DEFAULT_CHOICE COLBWT_LAYOUT_MISMATCH_LINES
constexpr int kDefaultLineSteps = 8;

// Device buffers, stream and events of one host-entry query, kept with the handle between calls
// (a file query makes one call per 64 Mbase batch: four hipMalloc / hipFree pairs, a stream and
// four events per call otherwise).  One caller at a time holds it; concurrent callers on the
// same replica work with a set of their own that lives for the call.
struct BatchScratch {
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    void *buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // bases, offsets, pml, cid, order
    size_t cap[5] = {0, 0, 0, 0, 0};
    ~BatchScratch() { release(); }
    void release() {
        for (int k = 0; k < 5; ++k) {
            if (buf[k]) (void)hipFree(buf[k]);
            buf[k] = nullptr;
            cap[k] = 0;
        }
        for (auto &e : ev) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
    hipError_t ready() {   // stream + events exist
        hipError_t e = hipSuccess;
        if (!stream) e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
        for (auto &x : ev)
            if (e == hipSuccess && !x) e = hipEventCreate(&x);
        return e;
    }
    hipError_t need(int k, size_t bytes) {   // buf[k] holds at least `bytes` (grown with headroom)
        if (bytes <= cap[k]) return hipSuccess;
        if (buf[k]) (void)hipFree(buf[k]);
        buf[k] = nullptr;
        cap[k] = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        const hipError_t e = hipMalloc(&buf[k], want);
        if (e == hipSuccess) cap[k] = want;
        return e;
    }
};

// Page-locked host array for the results of a batch: the device-to-host copy of 3 bytes per
// base is the largest part of the GPU stage, and runs ~4x faster into pinned memory.  Pinning
// costs about as much as a copy, so the file entry point keeps its buffers with the handle.
class PinnedBuf {
public:
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { if (p_) (void)hipHostFree(p_); }
    bool ensure(size_t bytes) {
        if (bytes <= cap_) return true;
        if (p_) (void)hipHostFree(p_);
        p_ = nullptr;
        cap_ = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        if (hipHostMalloc(&p_, want, 0) != hipSuccess) { p_ = nullptr; (void)hipGetLastError(); return false; }
        cap_ = want;
        return true;
    }
    template <typename T>
    T *as() const { return static_cast<T *>(p_); }

private:
    void *p_ = nullptr;
    size_t cap_ = 0;
};

constexpr int kFileBatches = 3;   // batches of reads in flight in colbwt_query_file

struct colbwt_index {
    Index ix;
    // result arrays of colbwt_query_file's batches (one file query at a time uses them)
    std::mutex file_mu;
    PinnedBuf file_pml[kFileBatches], file_cid[kFileBatches];
    // Further replicas of the table, one per extra device (colbwt_index_open_devices): each is a
    // handle of its own kind, owned by this one.  The host entry points shard a batch over
    // [this] + more; the device entry points address the replica on the buffers' device.
    std::vector<colbwt_index *> more;
    std::mutex scratch_mu;
    BatchScratch scratch;
    // Two pinned staging buffers for results that go to pageable host memory (kept for the
    // life of the handle: pinning costs as much as a copy).  One caller at a time uses them;
    // concurrent callers fall back to the runtime's own pageable copy.
    std::mutex stage_mu;
    void *stage[2] = {nullptr, nullptr};
    size_t stage_bytes = 0;
    ~colbwt_index() {
        for (colbwt_index *r : more) delete r;
        if (ix.device() >= 0) (void)hipSetDevice(ix.device());
        scratch.release();
        for (void *p : stage)
            if (p) (void)hipHostFree(p);
    }
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define API_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            rc = fail(e_ == hipErrorOutOfMemory ? COLBWT_ERR_NOMEM : COLBWT_ERR_HIP,          \
                      std::string(#expr) + ": " + hipGetErrorString(e_));                     \
            goto done;                                                                        \
        }                                                                                     \
    } while (0)

bool widths_ok(const colbwt_widths *w) {
    return !w || (w->bwt_bytes == 5 && w->run_bytes == 4 && w->len_bytes == 2 && w->id_bits == 8);
}

struct MappedFile {
    const uint8_t *data = nullptr;
    uint64_t len = 0;
    int fd = -1;
    ~MappedFile() {
        if (data && len) munmap((void *)data, len);
        if (fd >= 0) close(fd);
    }
    bool open(const std::string &path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return false;
        len = (uint64_t)st.st_size;
        if (len == 0) return true;
        void *p = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) return false;
        data = (const uint8_t *)p;
        return true;
    }
};

constexpr size_t kStageBytes = 128u << 20;     // per staging buffer
constexpr size_t kStageMinTotal = 256u << 20;  // smaller results: the plain copy is as good
constexpr unsigned kStageThreads = 8;

bool is_pageable(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) == hipSuccess)                 // ROCm >= 6 knows plain host memory
        return a.type == hipMemoryTypeUnregistered;                   // else pinned / registered / device
    (void)hipGetLastError();                                          // older answer: "invalid value"
    return true;
}

void parallel_memcpy(uint8_t *dst, const uint8_t *src, size_t n) {
    if (n < (8u << 20)) {
        memcpy(dst, src, n);
        return;
    }
    std::thread ts[kStageThreads - 1];
    const size_t per = (n / kStageThreads + 4095) & ~(size_t)4095;
    for (unsigned t = 1; t < kStageThreads; ++t) {
        const size_t a = std::min(n, per * t), b = std::min(n, per * (t + 1));
        ts[t - 1] = std::thread([=] { if (b > a) memcpy(dst + a, src + a, b - a); });
    }
    memcpy(dst, src, std::min(n, per));
    for (auto &th : ts) th.join();
}

struct D2HSegment {
    uint8_t *dst;
    const uint8_t *src;
    size_t bytes;
};

// Both pinned staging buffers of the handle, or neither: a failed allocation is "not staged"
// (the caller takes the plain copy), never a half-initialised pair.
bool ensure_stage(colbwt_index *idx) {
    if (idx->stage[0] && idx->stage[1]) return true;
    for (int b = 0; b < 2; ++b)
        if (hipHostMalloc(&idx->stage[b], kStageBytes, 0) != hipSuccess) {
            (void)hipGetLastError();
            idx->stage[b] = nullptr;
            if (idx->stage[0]) (void)hipHostFree(idx->stage[0]);
            idx->stage[0] = nullptr;
            return false;
        }
    idx->stage_bytes = kStageBytes;
    return true;
}

// Device -> pageable host memory through the handle's two pinned buffers: the DMA of chunk
// k runs while host threads copy chunk k-1 out of its buffer.  Returns hipSuccess or the
// first HIP error; the stream is idle afterwards.
hipError_t staged_d2h(colbwt_index *idx, const D2HSegment *seg, int n_seg, hipStream_t stream) {
    hipError_t e = hipSuccess;
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (auto &x : ev)
        if ((e = hipEventCreateWithFlags(&x, hipEventDisableTiming)) != hipSuccess) {
            for (auto &y : ev)
                if (y) (void)hipEventDestroy(y);
            return e;
        }
    struct Piece {
        uint8_t *dst;
        size_t n;
    } prev{nullptr, 0};
    int k = 0;
    for (int s = 0; s < n_seg && e == hipSuccess; ++s) {
        for (size_t o = 0; o < seg[s].bytes && e == hipSuccess; o += kStageBytes, ++k) {
            const size_t n = std::min(kStageBytes, seg[s].bytes - o);
            e = hipMemcpyAsync(idx->stage[k & 1], seg[s].src + o, n, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipEventRecord(ev[k & 1], stream);
            if (prev.dst && e == hipSuccess) {
                e = hipEventSynchronize(ev[(k - 1) & 1]);
                if (e == hipSuccess) parallel_memcpy(prev.dst, (const uint8_t *)idx->stage[(k - 1) & 1], prev.n);
            }
            prev = Piece{seg[s].dst + o, n};
        }
    }
    if (prev.dst && e == hipSuccess) {
        e = hipEventSynchronize(ev[(k - 1) & 1]);
        if (e == hipSuccess) parallel_memcpy(prev.dst, (const uint8_t *)idx->stage[(k - 1) & 1], prev.n);
    }
    const hipError_t e2 = hipStreamSynchronize(stream);
    for (auto &x : ev) (void)hipEventDestroy(x);
    return e != hipSuccess ? e : e2;
}

// One replica's part of a host-entry query: reads [0, n_reads) of `read_off`, whose offsets are
// relative to `bases` after subtracting `off0` (a shard of a larger batch keeps the caller's
// offsets).  Results go to pml / cid indexed like `bases`.
template <typename PmlT>
int query_batch_host(colbwt_index *idx, const uint8_t *bases, const uint64_t *read_off, uint64_t off0, uint64_t n_reads,
                     PmlT *pml, uint8_t *cid, uint64_t max_len, uint64_t min_len, colbwt_stats *stats, std::string &errmsg) {
    auto bad = [&](int code, const std::string &m) { errmsg = m; return code; };
    const uint64_t n_bases = read_off[n_reads] - off0;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n_bases == 0) {
        if (stats) stats->n_reads = n_reads;
        return COLBWT_OK;
    }
    int rc = select_device(idx->ix.device(), errmsg);
    if (rc != COLBWT_OK) return rc;

    // the handle's scratch when nobody else holds it, else one for this call
    std::unique_lock<std::mutex> lease(idx->scratch_mu, std::defer_lock);
    BatchScratch own;
    BatchScratch &S = lease.try_lock() ? idx->scratch : own;

    // Ragged batch: assign lanes by decreasing read length (counting sort over 256 length
    // classes is enough: waves only need reads of SIMILAR length side by side).  The line-row
    // kernel balances by itself (persistent lanes claim chunks of reads).
    std::vector<uint32_t> order;
    if (!idx->ix.line_rows() && n_reads <= 0xFFFFFFFFull && n_reads > 64 &&
        max_len > min_len + (min_len >> 2) + 16) {
        const uint64_t span = max_len - min_len + 1;
        uint32_t shift = 0;
        while ((span >> shift) > 4096) ++shift;
        std::vector<uint64_t> start((span >> shift) + 2, 0);
        for (uint64_t k = 0; k < n_reads; ++k) ++start[((max_len - (read_off[k + 1] - read_off[k])) >> shift) + 1];
        for (size_t b = 1; b < start.size(); ++b) start[b] += start[b - 1];
        order.resize(n_reads);
        for (uint64_t k = 0; k < n_reads; ++k)
            order[start[(max_len - (read_off[k + 1] - read_off[k])) >> shift]++] = (uint32_t)k;
    }
    std::vector<uint64_t> rebased;                            // offsets from 0 for this shard
    const uint64_t *off_src = read_off;
    if (off0 != 0) {
        rebased.resize(n_reads + 1);
        for (uint64_t k = 0; k <= n_reads; ++k) rebased[k] = read_off[k] - off0;
        off_src = rebased.data();
    }
    const uint64_t bases_alloc = (n_bases + 64 + 63) & ~63ull;  // the kernel reads whole 64-byte blocks
    float ms_h2d = 0, ms_k = 0, ms_d2h = 0;
    uint8_t *d_bases = nullptr, *d_cid = nullptr;
    uint64_t *d_off = nullptr;
    uint32_t *d_order = nullptr;
    PmlT *d_pml = nullptr;
    hipStream_t stream = nullptr;
#define HOST_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (void)hipGetLastError();                                                            \
            /* nothing of this call may still be running when the caller gets its arrays (and the \
               next caller the staging buffers) back: kernel and copies are drained first */      \
            if (stream) (void)hipStreamSynchronize(stream);                                     \
            return bad(e_ == hipErrorOutOfMemory ? COLBWT_ERR_NOMEM : COLBWT_ERR_HIP,           \
                       std::string(#expr) + ": " + hipGetErrorString(e_));                      \
        }                                                                                       \
    } while (0)
    HOST_HIP(S.ready());
    HOST_HIP(S.need(0, bases_alloc));
    HOST_HIP(S.need(1, (n_reads + 1) * sizeof(uint64_t)));
    HOST_HIP(S.need(2, ((n_bases + 15) & ~15ull) * sizeof(PmlT)));
    HOST_HIP(S.need(3, (n_bases + 15) & ~15ull));
    if (!order.empty()) HOST_HIP(S.need(4, n_reads * sizeof(uint32_t)));
    stream = S.stream;
    d_bases = (uint8_t *)S.buf[0];
    d_off = (uint64_t *)S.buf[1];
    d_pml = (PmlT *)S.buf[2];
    d_cid = (uint8_t *)S.buf[3];
    d_order = order.empty() ? nullptr : (uint32_t *)S.buf[4];

    HOST_HIP(hipEventRecord(S.ev[0], stream));
    HOST_HIP(hipMemsetAsync(d_bases + (bases_alloc - 128), 0, 128, stream));
    HOST_HIP(hipMemcpyAsync(d_bases, bases, n_bases, hipMemcpyHostToDevice, stream));
    HOST_HIP(hipMemcpyAsync(d_off, off_src, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    if (d_order) HOST_HIP(hipMemcpyAsync(d_order, order.data(), n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HOST_HIP(hipEventRecord(S.ev[1], stream));
    if (idx->ix.line_rows())
        launch_fat_query(idx->ix.table_fat(), d_bases, d_off, n_reads, n_bases, d_pml, (int)sizeof(PmlT), d_cid, d_order, stream);
    else if (idx->ix.layout() >= 2)
        launch_sk_query(idx->ix.table_k(), d_bases, d_off, n_reads, n_bases, d_pml, (int)sizeof(PmlT), d_cid, d_order, stream);
    else
        launch_pml_query(idx->ix.table(), d_bases, d_off, n_reads, d_pml, (int)sizeof(PmlT), d_cid, d_order, stream);
    HOST_HIP(hipGetLastError());
    HOST_HIP(hipEventRecord(S.ev[2], stream));
    {
        // results into pageable memory go through pinned staging buffers with several copier
        // threads (the runtime's own pageable path manages ~17 GB/s); pinned destinations and
        // small batches are copied directly
        std::unique_lock<std::mutex> stage_lock(idx->stage_mu, std::defer_lock);
        const bool staged = n_bases * (sizeof(PmlT) + 1) >= kStageMinTotal && is_pageable(pml) && is_pageable(cid) &&
                            stage_lock.try_lock() && ensure_stage(idx);
        if (staged) {
            const D2HSegment seg[2] = {{(uint8_t *)pml, (const uint8_t *)d_pml, n_bases * sizeof(PmlT)},
                                       {cid, d_cid, n_bases}};
            HOST_HIP(staged_d2h(idx, seg, 2, stream));
        } else {
            HOST_HIP(hipMemcpyAsync(pml, d_pml, n_bases * sizeof(PmlT), hipMemcpyDeviceToHost, stream));
            HOST_HIP(hipMemcpyAsync(cid, d_cid, n_bases, hipMemcpyDeviceToHost, stream));
        }
    }
    HOST_HIP(hipEventRecord(S.ev[3], stream));
    HOST_HIP(hipStreamSynchronize(stream));
    HOST_HIP(hipEventElapsedTime(&ms_h2d, S.ev[0], S.ev[1]));
    HOST_HIP(hipEventElapsedTime(&ms_k, S.ev[1], S.ev[2]));
    HOST_HIP(hipEventElapsedTime(&ms_d2h, S.ev[2], S.ev[3]));
#undef HOST_HIP
    if (stats) {
        stats->n_reads = n_reads;
        stats->n_bases = n_bases;
        stats->h2d_ms = ms_h2d;
        stats->kernel_ms = ms_k;
        stats->d2h_ms = ms_d2h;
        stats->algorithmic_bytes = n_bases * (uint64_t)kAlgBytesPerBase;
    }
    return COLBWT_OK;
}

// col_pml::query_pml for a batch in host memory, over every replica of the handle: the reads are
// independent (the reference walks them one after the other, pml_query.cpp:74-86), so the batch
// is cut into contiguous shards of equal base count, one per device, each queried by a host
// thread of its own on its device's stream and copied straight into its slice of the caller's
// arrays -- no exchange between devices.
template <typename PmlT>
int query_batch_all(colbwt_index *idx, const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, PmlT *pml,
                    uint8_t *cid, colbwt_stats *stats) {
    if (!idx) return fail(COLBWT_ERR_ARG, "null index");
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n_reads == 0) return COLBWT_OK;
    if (!read_off) return fail(COLBWT_ERR_ARG, "null read_off");
    if (read_off[0] != 0) return fail(COLBWT_ERR_ARG, "read_off[0] must be 0");
    uint64_t max_len = 0, min_len = ~0ull;
    for (uint64_t k = 0; k < n_reads; ++k) {
        if (read_off[k + 1] < read_off[k]) return fail(COLBWT_ERR_ARG, "read_off not non-decreasing");
        max_len = std::max(max_len, read_off[k + 1] - read_off[k]);
        min_len = std::min(min_len, read_off[k + 1] - read_off[k]);
    }
    const uint64_t n_bases = read_off[n_reads];
    if (sizeof(PmlT) == 2 && max_len > 65535)
        return fail(COLBWT_ERR_ARG, "read longer than 65535 bases: use colbwt_query_batch_u32");
    if (max_len > 0xFFFFFFFFull) return fail(COLBWT_ERR_ARG, "read longer than 2^32-1 bases");
    if (n_bases == 0) {
        if (stats) stats->n_reads = n_reads;
        return COLBWT_OK;
    }
    if (!bases || !pml || !cid) return fail(COLBWT_ERR_ARG, "null bases/pml/cid");

    std::vector<colbwt_index *> reps{idx};
    reps.insert(reps.end(), idx->more.begin(), idx->more.end());
    const size_t R = reps.size();
    if (R == 1) {
        std::string msg;
        const int rc = query_batch_host<PmlT>(idx, bases, read_off, 0, n_reads, pml, cid, max_len, min_len, stats, msg);
        return rc == COLBWT_OK ? rc : fail(rc, msg);
    }
    // shard boundaries by base count (same rule as multi_gpu.shard_reads)
    std::vector<uint64_t> cut(R + 1, 0);
    cut[R] = n_reads;
    for (size_t r = 1; r < R; ++r) {
        const uint64_t target = n_bases / R * r + n_bases % R * r / R;
        const uint64_t k = (uint64_t)(std::lower_bound(read_off, read_off + n_reads + 1, target) - read_off);
        cut[r] = std::min(std::max(k, cut[r - 1]), n_reads);
    }
    std::vector<int> rcs(R, COLBWT_OK);
    std::vector<std::string> msgs(R);
    std::vector<colbwt_stats> sts(R);
    auto work = [&](size_t r) {
        const uint64_t lo = cut[r], hi = cut[r + 1];
        if (hi == lo) return;
        const uint64_t off0 = read_off[lo];
        rcs[r] = query_batch_host<PmlT>(reps[r], bases + off0, read_off + lo, off0, hi - lo, pml + off0, cid + off0, max_len,
                                        min_len, &sts[r], msgs[r]);
    };
    std::vector<std::thread> threads;
    for (size_t r = 1; r < R; ++r) threads.emplace_back(work, r);
    work(0);
    for (auto &t : threads) t.join();
    for (size_t r = 0; r < R; ++r)
        if (rcs[r] != COLBWT_OK) return fail(rcs[r], "device " + std::to_string(reps[r]->ix.device()) + ": " + msgs[r]);
    if (stats) {
        for (size_t r = 0; r < R; ++r) {
            stats->n_reads += sts[r].n_reads;
            stats->n_bases += sts[r].n_bases;
            stats->algorithmic_bytes += sts[r].algorithmic_bytes;
            stats->h2d_ms = std::max(stats->h2d_ms, sts[r].h2d_ms);          // the devices work side by side
            stats->kernel_ms = std::max(stats->kernel_ms, sts[r].kernel_ms);
            stats->d2h_ms = std::max(stats->d2h_ms, sts[r].d2h_ms);
        }
    }
    return COLBWT_OK;
}

}  // namespace

namespace {

// One batch of reads on its way through the three stages of colbwt_query_file.
struct FileBatch {
    std::vector<uint8_t> bases;
    std::vector<uint64_t> off;
    std::vector<std::string> names;
    PinnedBuf own_pml, own_cid;      // u16 or u32 values / u8 col ids, one per base
    PinnedBuf *pml = &own_pml, *cid = &own_cid;   // ... or the handle's, kept across calls
    bool wide = false;
};

// Hand-over point between two stages (a few batches deep; nullptr = end of stream).
class BatchQueue {
public:
    void push(FileBatch *b) {
        std::lock_guard<std::mutex> g(mu_);
        q_.push_back(b);
        cv_.notify_one();
    }
    FileBatch *pop() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return !q_.empty(); });
        FileBatch *b = q_.front();
        q_.pop_front();
        return b;
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<FileBatch *> q_;
};

}  // namespace

// The replica of a multi-device handle that lives on the device holding `d_ptr` (the first one
// when the pointer's device cannot be told or holds no replica).
static colbwt_index *replica_for(colbwt_index *idx, const void *d_ptr) {
    if (idx->more.empty()) return idx;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, d_ptr) != hipSuccess) {
        (void)hipGetLastError();
        return idx;
    }
    if (a.device == idx->ix.device()) return idx;
    for (colbwt_index *r : idx->more)
        if (r->ix.device() == a.device) return r;
    return idx;
}

extern "C" {

const char *colbwt_version(void) { return "colbwt-mi355x 0.2.0 (gfx950)"; }

const char *colbwt_last_error(void) { return g_err.c_str(); }

static int default_layout() {
    const char *e = getenv("COLBWT_LAYOUT");   // override: 1 = one-step, 2 / 3 = K-step rows, 4 = line rows
    if (e && e[0] >= '1' && e[0] <= '6' && e[1] == 0) return e[0] - '0';
    return COLBWT_LAYOUT_DEFAULT_CHOICE;
}

// Own steps K of a line-row open: from the layout argument, else COLBWT_LINE_ROWS_STEPS
// (experiments), else the default.
static int line_rows_steps(int layout_arg) {
    int steps = (layout_arg >> 8) & 0xFF;
    if (steps == 0) {
        steps = kDefaultLineSteps;
        const char *e = getenv("COLBWT_LINE_ROWS_STEPS");
        if (e && e[0] >= '2' && e[0] <= '8' && e[1] == 0) steps = e[0] - '0';
    }
    return steps;
}

int colbwt_index_open_memory(const void *bytes, uint64_t len, const colbwt_widths *widths, int device,
                             colbwt_index **out) {
    return colbwt_index_open_memory_layout(bytes, len, widths, device, COLBWT_LAYOUT_AUTO, out);
}

int colbwt_index_open_memory_layout(const void *bytes, uint64_t len, const colbwt_widths *widths, int device,
                                    int layout, colbwt_index **out) {
    if (!out) return fail(COLBWT_ERR_ARG, "null out");
    const bool automatic = layout == COLBWT_LAYOUT_AUTO;
    if (automatic) layout = default_layout();
    const int steps = line_rows_steps(layout);
    const bool steps_given = ((layout >> 8) & 0xFF) != 0 || getenv("COLBWT_LINE_ROWS_STEPS") != nullptr;
    layout &= 0xFF;
    if (layout < COLBWT_LAYOUT_ONE_STEP || layout > COLBWT_LAYOUT_MISMATCH_LINES_DEEP) return fail(COLBWT_ERR_ARG, "bad layout");
    if (layout >= COLBWT_LAYOUT_LINE_ROWS && !fat_steps_supported(steps))
        return fail(COLBWT_ERR_ARG, "line rows: own steps must be 4..8");
    *out = nullptr;
    if (!widths_ok(widths))
        return fail(COLBWT_ERR_ARG, "only the shipped widths BWT_BYTES=5 RUN_BYTES=4 LEN_BYTES=2 ID_BITS=8 are supported");
    colbwt_index *idx = new (std::nothrow) colbwt_index();
    if (!idx) return fail(COLBWT_ERR_NOMEM, "out of host memory");
    std::string err;
    // AUTO's first choice: mismatch lines, with deep entries when the table leaves room for batches
    const bool deep_if_room = automatic && layout == COLBWT_LAYOUT_MISMATCH_LINES && getenv("COLBWT_LAYOUT") == nullptr;
    int rc = idx->ix.load((const uint8_t *)bytes, len, device, deep_if_room ? kLayoutMismatchLinesAuto : layout, err, steps);
    if (rc == COLBWT_ERR_NOMEM && automatic) {
        // The table does not fit that way (HBM, or more than 2^32-2 refined rows): the ladder of
        // smaller layouts -- line rows at K = 8, 6, 4, three-, two-, one-step rows (the first attempt
        // already went from deep to plain mismatch entries by itself when the deep ones had no room).
        // A line-row build says at which refinement level it gave up (after that level's counting
        // pass, before anything of it was allocated): candidates that have to pass the same level
        // are not tried at all, so an index far too large for line rows costs one counting pass.
        struct Candidate { int layout, steps; };
        static const Candidate ladder[] = {{COLBWT_LAYOUT_MISMATCH_LINES_DEEP, 8}, {COLBWT_LAYOUT_MISMATCH_LINES, 8},   // (deep: only when asked for)
                                           {COLBWT_LAYOUT_LINE_ROWS, 8}, {COLBWT_LAYOUT_LINE_ROWS, 6},
                                           {COLBWT_LAYOUT_LINE_ROWS, 4},      {COLBWT_LAYOUT_THREE_STEP, 0}, {COLBWT_LAYOUT_TWO_STEP, 0},
                                           {COLBWT_LAYOUT_ONE_STEP, 0}};
        int hopeless_from = layout >= COLBWT_LAYOUT_LINE_ROWS ? idx->ix.fat_failed_level() : 0;   // steps >= this cannot be built
        for (const Candidate &c : ladder) {
            if (c.layout > layout || (c.layout == layout && (c.steps >= steps || steps_given))) continue;   // at or above the start
            if (c.layout >= COLBWT_LAYOUT_LINE_ROWS && (steps_given || (hopeless_from && c.steps >= hopeless_from))) continue;
            rc = idx->ix.load((const uint8_t *)bytes, len, device, c.layout, err, c.steps);
            if (rc != COLBWT_ERR_NOMEM) break;
            if (c.layout >= COLBWT_LAYOUT_LINE_ROWS) {
                const int f = idx->ix.fat_failed_level();
                if (f && (!hopeless_from || f < hopeless_from)) hopeless_from = f;
            }
        }
    }
    if (rc != COLBWT_OK) {
        delete idx;
        return fail(rc, err);
    }
    *out = idx;
    return COLBWT_OK;
}

int colbwt_index_open(const char *prefix_or_file, const colbwt_widths *widths, int device, colbwt_index **out) {
    return colbwt_index_open_layout(prefix_or_file, widths, device, COLBWT_LAYOUT_AUTO, out);
}

int colbwt_index_open_layout(const char *prefix_or_file, const colbwt_widths *widths, int device, int layout,
                             colbwt_index **out) {
    if (!prefix_or_file || !out) return fail(COLBWT_ERR_ARG, "null argument");
    *out = nullptr;
    // pml_query.cpp:110-111: filename = prefix + ".col_pml" (col_bwt.hpp:434-437)
    MappedFile mf;
    std::string path = std::string(prefix_or_file) + ".col_pml";
    if (!mf.open(path)) {
        MappedFile direct;
        path = prefix_or_file;
        if (!direct.open(path))
            return fail(COLBWT_ERR_IO, std::string("cannot open ") + prefix_or_file + ".col_pml (or " + prefix_or_file + ")");
        return colbwt_index_open_memory_layout(direct.data, direct.len, widths, device, layout, out);
    }
    return colbwt_index_open_memory_layout(mf.data, mf.len, widths, device, layout, out);
}

// The same table on several devices (the replacement's open takes the devices to
// use): replica k is loaded on devices[k]; every replica ends up with the layout the FIRST one
// got (AUTO falls back on the first device only, so that all replicas answer from the same kind
// of table), a device may be listed more than once (two replicas in one HBM).
int colbwt_index_open_memory_devices(const void *bytes, uint64_t len, const colbwt_widths *widths, const int *devices,
                                     int n_devices, int layout, colbwt_index **out) {
    if (!out) return fail(COLBWT_ERR_ARG, "null out");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return fail(COLBWT_ERR_ARG, "device list must hold 1 .. 64 devices");
    colbwt_index *first = nullptr;
    int rc = colbwt_index_open_memory_layout(bytes, len, widths, devices[0], layout, &first);
    if (rc != COLBWT_OK) return rc;
    colbwt_info info;
    (void)colbwt_index_info(first, &info);
    const int same = (int)info.layout | (info.layout >= COLBWT_LAYOUT_LINE_ROWS ? (int)(info.layout_shape >> 8) << 8 : 0);
    for (int k = 1; k < n_devices; ++k) {
        colbwt_index *rep = nullptr;
        rc = colbwt_index_open_memory_layout(bytes, len, widths, devices[k], same, &rep);
        if (rc != COLBWT_OK) {
            const std::string msg = "replica on device " + std::to_string(devices[k]) + ": " + g_err;
            delete first;
            return fail(rc, msg);
        }
        first->more.push_back(rep);
    }
    *out = first;
    return COLBWT_OK;
}

int colbwt_index_open_devices(const char *prefix_or_file, const colbwt_widths *widths, const int *devices, int n_devices,
                              int layout, colbwt_index **out) {
    if (!prefix_or_file || !out) return fail(COLBWT_ERR_ARG, "null argument");
    *out = nullptr;
    MappedFile mf;                                            // pml_query.cpp:110-111, as colbwt_index_open_layout
    if (!mf.open(std::string(prefix_or_file) + ".col_pml")) {
        MappedFile direct;
        if (!direct.open(prefix_or_file))
            return fail(COLBWT_ERR_IO, std::string("cannot open ") + prefix_or_file + ".col_pml (or " + prefix_or_file + ")");
        return colbwt_index_open_memory_devices(direct.data, direct.len, widths, devices, n_devices, layout, out);
    }
    return colbwt_index_open_memory_devices(mf.data, mf.len, widths, devices, n_devices, layout, out);
}

void colbwt_index_close(colbwt_index *idx) { delete idx; }

int colbwt_index_info(const colbwt_index *idx, colbwt_info *out) {
    if (!idx || !out) return fail(COLBWT_ERR_ARG, "null argument");
    out->bwt_r = idx->ix.bwt_r();
    out->n = idx->ix.n();
    out->r = idx->ix.r();
    out->sigma = idx->ix.sigma();
    out->device = (uint32_t)idx->ix.device();
    out->device_bytes = idx->ix.device_bytes();
    out->layout = (uint32_t)idx->ix.layout();
    out->layout_shape = idx->ix.line_rows() ? (idx->ix.table_fat().steps << 8) | kFatSlotSteps : 0;
    out->table_rows = idx->ix.table_rows();
    out->n_devices = 1 + (uint32_t)idx->more.size();
    out->reserved_ = 0;
    return COLBWT_OK;
}

int colbwt_query_batch(colbwt_index *idx, const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads,
                       uint16_t *pml, uint8_t *cid, colbwt_stats *stats) {
    return query_batch_all<uint16_t>(idx, bases, read_off, n_reads, pml, cid, stats);
}

int colbwt_query_batch_u32(colbwt_index *idx, const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads,
                           uint32_t *pml, uint8_t *cid, colbwt_stats *stats) {
    return query_batch_all<uint32_t>(idx, bases, read_off, n_reads, pml, cid, stats);
}

int colbwt_query_device(colbwt_index *idx, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads,
                        uint64_t n_bases, void *d_pml, int pml_bytes, uint8_t *d_cid, void *hip_stream,
                        colbwt_stats *stats) {
    return colbwt_query_device_ordered(idx, d_bases, d_read_off, n_reads, n_bases, d_pml, pml_bytes, d_cid, nullptr,
                                       hip_stream, stats);
}

int colbwt_query_device_ordered(colbwt_index *idx, const uint8_t *d_bases, const uint64_t *d_read_off,
                                uint64_t n_reads, uint64_t n_bases, void *d_pml, int pml_bytes, uint8_t *d_cid,
                                const uint32_t *d_order, void *hip_stream, colbwt_stats *stats) {
    if (!idx) return fail(COLBWT_ERR_ARG, "null index");
    if (stats) memset(stats, 0, sizeof(*stats));
    if (pml_bytes != 2 && pml_bytes != 4) return fail(COLBWT_ERR_ARG, "pml_bytes must be 2 or 4");
    if (n_reads == 0) return COLBWT_OK;
    if (!d_bases || !d_read_off || !d_pml || !d_cid) return fail(COLBWT_ERR_ARG, "null device pointer");
    if (((uintptr_t)d_bases & 15) || ((uintptr_t)d_pml & 31) || ((uintptr_t)d_cid & 15))
        return fail(COLBWT_ERR_ARG, "d_bases/d_cid must be 16-byte aligned and d_pml 32-byte aligned");
    idx = replica_for(idx, d_bases);
    int rc = select_device(idx->ix.device(), g_err);
    if (rc != COLBWT_OK) return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    if (stats) {
        API_HIP(hipEventCreate(&e0));
        API_HIP(hipEventCreate(&e1));
        API_HIP(hipEventRecord(e0, stream));
    }
    if (idx->ix.line_rows())
        launch_fat_query(idx->ix.table_fat(), d_bases, d_read_off, n_reads, n_bases, d_pml, pml_bytes, d_cid, d_order, stream);
    else if (idx->ix.layout() >= 2)
        launch_sk_query(idx->ix.table_k(), d_bases, d_read_off, n_reads, n_bases, d_pml, pml_bytes, d_cid, d_order, stream);
    else
        launch_pml_query(idx->ix.table(), d_bases, d_read_off, n_reads, d_pml, pml_bytes, d_cid, d_order, stream);
    API_HIP(hipGetLastError());
    if (stats) {
        API_HIP(hipEventRecord(e1, stream));
        API_HIP(hipEventSynchronize(e1));
        API_HIP(hipEventElapsedTime(&ms, e0, e1));
        stats->n_reads = n_reads;
        stats->n_bases = n_bases;
        stats->kernel_ms = ms;
        stats->algorithmic_bytes = n_bases * (uint64_t)kAlgBytesPerBase;
    }
    rc = COLBWT_OK;
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

// pml_query vec mode (pml_query.cpp:92-143) as a three-stage pipeline over batches of reads:
// a reader thread parses the FASTA/FASTQ (kseq semantics; a plain FASTA by several threads),
// the calling thread runs the GPU query (on every replica of the handle), a writer thread lays
// out and writes the two result files -- so the wall time is the slowest stage's, not the sum.
// Output bytes and order are those of the sequential loop.  `binary`: the container of
// bin_writer.h instead of the reference's text.
static int query_file_impl(colbwt_index *idx, const char *pattern_path, const std::string &pml_name,
                           const std::string &cid_name, uint64_t batch_bases, colbwt_stats *stats, bool binary) {
    if (stats) memset(stats, 0, sizeof(*stats));
    const size_t replicas = 1 + idx->more.size();
    if (batch_bases == 0) batch_bases = (64ull << 20) * replicas;
    ParallelFasta fasta;
    FastxReader reader;
    bool parallel = fasta.open(pattern_path);
    if (!parallel && !reader.open(pattern_path)) return fail(COLBWT_ERR_IO, std::string("cannot open pattern file ") + pattern_path);
    TextWriter wp, wc;
    BinWriter bp, bc;
    if (!(binary ? bp.open(pml_name) : wp.open(pml_name))) return fail(COLBWT_ERR_IO, "cannot create " + pml_name);
    if (!(binary ? bc.open(cid_name) : wc.open(cid_name))) return fail(COLBWT_ERR_IO, "cannot create " + cid_name);

    FileBatch pool[kFileBatches];
    std::unique_lock<std::mutex> pinned(idx->file_mu, std::defer_lock);
    if (pinned.try_lock())               // the handle's pinned arrays, unless another file query holds them
        for (int k = 0; k < kFileBatches; ++k) {
            pool[k].pml = &idx->file_pml[k];
            pool[k].cid = &idx->file_cid[k];
        }
    BatchQueue free_q, parsed_q, done_q;
    for (auto &b : pool) free_q.push(&b);
    std::atomic<bool> stop{false};   // a later stage failed: the reader stops feeding
    std::atomic<bool> reader_failed{false};
    const bool trace = getenv("COLBWT_TRACE") != nullptr;
    double t_parse = 0, t_gpu = 0, t_format = 0;   // busy seconds of the three stages
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    const unsigned host_threads = std::min(16u, std::max(2u, std::thread::hardware_concurrency()));

    std::thread reader_thread([&] {
        bool more = true;
        while (more && !stop.load()) {
            FileBatch *b = free_q.pop();
            const double t0 = now();
            b->bases.clear();
            b->names.clear();
            b->off.assign(1, 0);
            uint64_t max_len = 0;
            if (parallel) {
                const ParallelFasta::Result r = fasta.next_batch(batch_bases, host_threads / 2, b->names, b->bases, b->off, max_len);
                if (r == ParallelFasta::kEnd) more = false;
                if (r == ParallelFasta::kNotPlainFasta) {       // FASTQ-style lines ahead: one record at a time from here
                    parallel = false;
                    if (!reader.open_at(pattern_path, fasta.position())) {
                        reader_failed.store(true);
                        more = false;
                    }
                }
            }
            if (!parallel && more) {
                std::string name;
                while (b->bases.size() < batch_bases) {  // pml_query.cpp:74 while (patterns.read())
                    if (!reader.next(name, b->bases)) {
                        more = false;
                        break;
                    }
                    b->names.push_back(name);
                    max_len = std::max<uint64_t>(max_len, b->bases.size() - b->off.back());
                    b->off.push_back(b->bases.size());
                }
            }
            b->wide = max_len > 65535;
            t_parse += now() - t0;
            if (b->names.empty()) {
                free_q.push(b);
                if (more) continue;
                break;
            }
            parsed_q.push(b);
        }
        parsed_q.push(nullptr);
    });

    std::thread writer_thread([&] {
        for (;;) {
            FileBatch *b = done_q.pop();
            if (!b) break;
            const uint64_t n_reads = b->names.size();
            const double t0 = now();
            // pml_query.cpp:78-85; the two files are laid out side by side, each by several
            // host threads (same bytes, same order as the sequential loop)
            std::thread cid_thread([&] {
                if (binary) bc.batch<uint8_t>(b->names, b->off.data(), b->cid->as<uint8_t>(), n_reads, host_threads / 2);
                else wc.batch(b->names, b->off.data(), b->cid->as<uint8_t>(), n_reads, host_threads / 2);
            });
            if (binary) {
                if (b->wide) bp.batch<uint16_t>(b->names, b->off.data(), b->pml->as<uint32_t>(), n_reads, host_threads / 2);
                else bp.batch<uint16_t>(b->names, b->off.data(), b->pml->as<uint16_t>(), n_reads, host_threads / 2);
            } else {
                if (b->wide) wp.batch(b->names, b->off.data(), b->pml->as<uint32_t>(), n_reads, host_threads / 2);
                else wp.batch(b->names, b->off.data(), b->pml->as<uint16_t>(), n_reads, host_threads / 2);
            }
            cid_thread.join();
            t_format += now() - t0;
            free_q.push(b);
        }
    });

    int rc = COLBWT_OK;
    for (;;) {
        FileBatch *b = parsed_q.pop();
        if (!b) break;
        if (rc != COLBWT_OK) {           // keep draining so the reader can finish
            free_q.push(b);
            continue;
        }
        const uint64_t n_reads = b->names.size();
        const uint64_t nb = b->bases.size();
        const double t0 = now();
        colbwt_stats st{};
        rc = select_device(idx->ix.device(), g_err);
        if (rc == COLBWT_OK && (!b->cid->ensure(nb) || !b->pml->ensure(nb * (b->wide ? 4 : 2))))
            rc = fail(COLBWT_ERR_NOMEM, "cannot pin host memory for a batch of results");
        if (rc != COLBWT_OK) {
        } else if (b->wide) {
            rc = colbwt_query_batch_u32(idx, b->bases.data(), b->off.data(), n_reads, b->pml->as<uint32_t>(),
                                        b->cid->as<uint8_t>(), &st);
        } else {
            rc = colbwt_query_batch(idx, b->bases.data(), b->off.data(), n_reads, b->pml->as<uint16_t>(),
                                    b->cid->as<uint8_t>(), &st);
        }
        t_gpu += now() - t0;
        if (rc != COLBWT_OK) {
            stop.store(true);
            free_q.push(b);
            continue;
        }
        if (stats) {
            stats->n_reads += st.n_reads;
            stats->n_bases += st.n_bases;
            stats->h2d_ms += st.h2d_ms;
            stats->kernel_ms += st.kernel_ms;
            stats->d2h_ms += st.d2h_ms;
            stats->algorithmic_bytes += st.algorithmic_bytes;
        }
        done_q.push(b);
    }
    done_q.push(nullptr);
    reader_thread.join();
    writer_thread.join();
    const bool okp = binary ? bp.close() : wp.close(), okc = binary ? bc.close() : wc.close();
    if (trace)
        fprintf(stderr, "colbwt_query_file: wall %.3f s; busy: parse %.3f, gpu %.3f, layout+write %.3f\n",
                now() - t_begin, t_parse, t_gpu, t_format);
    if (rc != COLBWT_OK) return rc;     // message set by the failing call on this thread
    if (reader_failed.load()) return fail(COLBWT_ERR_IO, std::string("cannot re-open pattern file ") + pattern_path);
    if (!okp || !okc) return fail(COLBWT_ERR_IO, "short write on " + pml_name + " / " + cid_name);
    return COLBWT_OK;
}

int colbwt_query_file(colbwt_index *idx, const char *pattern_path, const char *pml_path, const char *cid_path,
                      uint64_t batch_bases, colbwt_stats *stats) {
    if (!idx || !pattern_path) return fail(COLBWT_ERR_ARG, "null argument");
    // pml_query.cpp:124-125
    return query_file_impl(idx, pattern_path, pml_path ? pml_path : std::string(pattern_path) + ".pml",
                           cid_path ? cid_path : std::string(pattern_path) + ".cid", batch_bases, stats, false);
}

int colbwt_query_file_binary(colbwt_index *idx, const char *pattern_path, const char *pml_bin_path, const char *cid_bin_path,
                             uint64_t batch_bases, colbwt_stats *stats) {
    if (!idx || !pattern_path) return fail(COLBWT_ERR_ARG, "null argument");
    return query_file_impl(idx, pattern_path, pml_bin_path ? pml_bin_path : std::string(pattern_path) + ".pml.bin",
                           cid_bin_path ? cid_bin_path : std::string(pattern_path) + ".cid.bin", batch_bases, stats, true);
}

int colbwt_binary_to_text(const char *bin_path, int value_bytes, const char *text_path) {
    if (!bin_path || !text_path) return fail(COLBWT_ERR_ARG, "null argument");
    std::string err;
    if (!binary_to_text(bin_path, value_bytes, text_path, err)) return fail(value_bytes == 1 || value_bytes == 2 ? COLBWT_ERR_IO : COLBWT_ERR_ARG, err);
    return COLBWT_OK;
}

int colbwt_pml_pack_device(const uint16_t *d_pml, uint64_t n_bases, uint32_t *d_mask, void *hip_stream) {
    if (!d_pml || !d_mask) return fail(COLBWT_ERR_ARG, "null argument");
    if ((uintptr_t)d_pml % 32 || (uintptr_t)d_mask % 4) return fail(COLBWT_ERR_ARG, "d_pml must be 32-byte aligned");
    launch_pml_pack(d_pml, n_bases, d_mask, (hipStream_t)hip_stream);
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(COLBWT_ERR_HIP, std::string("colbwt_pml_pack_device: ") + hipGetErrorString(e));
    }
    return COLBWT_OK;
}

int colbwt_read_end_mask_device(const uint64_t *d_read_off, uint64_t n_reads, uint32_t *d_mask, void *hip_stream) {
    if (!d_read_off || !d_mask) return fail(COLBWT_ERR_ARG, "null argument");
    launch_read_end_mask(d_read_off, n_reads, d_mask, (hipStream_t)hip_stream);
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(COLBWT_ERR_HIP, std::string("colbwt_read_end_mask_device: ") + hipGetErrorString(e));
    }
    return COLBWT_OK;
}

int colbwt_pml_unpack_device(const uint32_t *d_zero_mask, const uint32_t *d_end_mask, uint64_t first_word,
                             uint64_t n_words, uint64_t total_words, uint16_t *d_pml, void *hip_stream) {
    if (!d_zero_mask || !d_end_mask || !d_pml) return fail(COLBWT_ERR_ARG, "null argument");
    if (first_word + n_words > total_words) return fail(COLBWT_ERR_ARG, "word range beyond the masks");
    if ((uintptr_t)d_pml % 64) return fail(COLBWT_ERR_ARG, "d_pml must be 64-byte aligned");
    launch_pml_unpack(d_zero_mask, d_end_mask, first_word, n_words, total_words, d_pml, (hipStream_t)hip_stream);
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(COLBWT_ERR_HIP, std::string("colbwt_pml_unpack_device: ") + hipGetErrorString(e));
    }
    return COLBWT_OK;
}

int colbwt_index_cid_dictionary(const colbwt_index *idx, uint8_t *ids, uint32_t *n_ids) {
    if (!idx || !ids || !n_ids) return fail(COLBWT_ERR_ARG, "null argument");
    uint32_t n = 0;
    for (uint32_t c = 0; c < 256; ++c)
        if ((idx->ix.cid_set()[c >> 5] >> (c & 31)) & 1u) ids[n++] = (uint8_t)c;
    *n_ids = n;
    return COLBWT_OK;
}

static uint32_t cid_code_bits(uint32_t n_ids) {
    uint32_t bits = 1;
    while ((1u << bits) < n_ids) ++bits;
    return bits;
}

uint32_t colbwt_cid_code_bits(uint32_t n_ids) { return n_ids >= 1 && n_ids <= 256 ? cid_code_bits(n_ids) : 0; }

int colbwt_cid_pack_device(const uint8_t *d_cid, uint64_t n_bases, const uint8_t *ids, uint32_t n_ids, uint32_t *d_planes,
                           void *hip_stream) {
    if (!d_cid || !ids || !d_planes) return fail(COLBWT_ERR_ARG, "null argument");
    if (n_ids < 1 || n_ids > 256) return fail(COLBWT_ERR_ARG, "a dictionary holds 1 .. 256 col ids");
    if ((uintptr_t)d_cid % 16) return fail(COLBWT_ERR_ARG, "d_cid must be 16-byte aligned");
    CidLut code_of;
    memset(code_of.v, 0, sizeof(code_of.v));
    for (uint32_t k = 0; k < n_ids; ++k) code_of.v[ids[k]] = (uint8_t)k;
    launch_cid_pack(d_cid, n_bases, code_of, cid_code_bits(n_ids), d_planes, (hipStream_t)hip_stream);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(COLBWT_ERR_HIP, std::string("colbwt_cid_pack_device: ") + hipGetErrorString(e));
    return COLBWT_OK;
}

int colbwt_cid_unpack_device(const uint32_t *d_planes, uint64_t first_word, uint64_t n_words, const uint8_t *ids, uint32_t n_ids,
                             uint8_t *d_cid, void *hip_stream) {
    if (!d_planes || !ids || !d_cid) return fail(COLBWT_ERR_ARG, "null argument");
    if (n_ids < 1 || n_ids > 256) return fail(COLBWT_ERR_ARG, "a dictionary holds 1 .. 256 col ids");
    if ((uintptr_t)d_cid % 32) return fail(COLBWT_ERR_ARG, "d_cid must be 32-byte aligned");
    CidLut id_of;
    memset(id_of.v, 0, sizeof(id_of.v));
    for (uint32_t k = 0; k < n_ids; ++k) id_of.v[k] = ids[k];
    launch_cid_unpack(d_planes, first_word, n_words, id_of, cid_code_bits(n_ids), d_cid, (hipStream_t)hip_stream);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(COLBWT_ERR_HIP, std::string("colbwt_cid_unpack_device: ") + hipGetErrorString(e));
    return COLBWT_OK;
}

int colbwt_synth_reads_device(colbwt_index *idx, uint64_t n_reads, uint32_t read_len, uint32_t sub_permille,
                              uint64_t seed, uint8_t *d_bases, uint64_t *d_read_off, void *hip_stream) {
    if (!idx || !d_bases || !d_read_off || read_len == 0) return fail(COLBWT_ERR_ARG, "bad argument");
    idx = replica_for(idx, d_bases);
    int rc = select_device(idx->ix.device(), g_err);
    if (rc != COLBWT_OK) return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    API_HIP(hipMemsetAsync(d_bases + n_reads * (uint64_t)read_len, 0, 64, stream));
    if (idx->ix.line_rows())
        launch_fat_synth_reads(idx->ix.table_fat(), n_reads, read_len, sub_permille, seed, d_bases, d_read_off, stream);
    else if (idx->ix.layout() >= 2)   // the one-step tables are gone once the K-step rows exist
        launch_sk_synth_reads(idx->ix.table_k(), n_reads, read_len, sub_permille, seed, d_bases, d_read_off, stream);
    else
        launch_synth_reads(idx->ix.table(), n_reads, read_len, sub_permille, seed, d_bases, d_read_off, stream);
    API_HIP(hipGetLastError());
    rc = COLBWT_OK;
done:
    return rc;
}

}  // extern "C"
