// dev_mem.h -- device allocations of the index builder: every table and every temporary of
// Index::load / build_sk goes through dev_alloc / dev_free (COLBWT_ALLOC_LOG=1 lists the large ones
// on stderr), so that
//   * a failed build frees what it had allocated (DevPtr is RAII; the failure paths of the
//     K-step refinement used to leak their temporaries -- exactly where the AUTO layout
//     fallback needs the HBM back), and
//   * an index can be held to an HBM budget: COLBWT_HBM_BUDGET_MB (environment) bounds the bytes
//     ONE open may hold at any moment, temporaries included; an allocation beyond it fails like
//     a real hipErrorOutOfMemory.  Used to keep room for read batches next to a large index, and
//     by the tests to force the fallback without filling 288 GB.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include <algorithm>

#include "dev_vmm.h"

namespace colbwt {

struct DevBudget {
    uint64_t limit = ~0ull;   // bytes
    uint64_t used = 0;
    uint64_t peak = 0;
};

// the budget of the open that is running on this thread (nullptr: unlimited, untracked)
inline DevBudget *&current_budget() {
    static thread_local DevBudget *b = nullptr;
    return b;
}

inline uint64_t env_budget_bytes() {
    const char *e = getenv("COLBWT_HBM_BUDGET_MB");
    if (!e || !*e) return ~0ull;
    char *end = nullptr;
    const double mb = strtod(e, &end);   // fractions allowed ("0.5")
    if (end == e || !(mb > 0)) return ~0ull;
    return (uint64_t)(mb * 1048576.0);
}

// The table the query kernel fetches from stays a hipMalloc block: granules are created from whatever
// memory is free at that moment, and on a box whose previous process has just exited that is the
// scraps between regions still being wiped -- consecutive bench.py processes on one box measured
// 10.7-11.1 ms per launch on a fresh box but 11.7-12.6 ms later, against 10.9-11.5 ms throughout for a
// hipMalloc table, which waits for the wipe and gets the memory whole
// (profiles/r03w_bench_sequence.jsonl).  The builder's temporaries do not care.
inline bool &plain_alloc() {
    static thread_local bool plain = false;
    return plain;
}
struct PlainAllocScope {
    bool prev;
    explicit PlainAllocScope(bool on = true) : prev(plain_alloc()) { plain_alloc() = prev || on; }
    ~PlainAllocScope() { plain_alloc() = prev; }
    PlainAllocScope(const PlainAllocScope &) = delete;
    PlainAllocScope &operator=(const PlainAllocScope &) = delete;
};

inline hipError_t dev_alloc(void **p, uint64_t bytes) {
    *p = nullptr;
    DevBudget *b = current_budget();
    if (b && (bytes > b->limit || b->used > b->limit - bytes)) return hipErrorOutOfMemory;
    // large arrays: address ranges over the device's granule store (dev_vmm.h), so that an open does not
    // hand memory back to the driver and wait for it to be wiped; the rest: hipMalloc
    const bool ranged = vmm_takes(bytes) && !plain_alloc();
    hipError_t e = ranged ? vmm_alloc(p, bytes) : hipMalloc(p, bytes ? bytes : 1);
    if (e != hipSuccess && vmm_idle_bytes() != 0) {
        // the room may be in granules the store keeps for its next array
        int dev = 0;
        (void)hipGetLastError();
        if (hipGetDevice(&dev) == hipSuccess) vmm_trim(dev);
        e = ranged ? vmm_alloc(p, bytes) : hipMalloc(p, bytes ? bytes : 1);
    }
    if (e != hipSuccess && ranged) {   // the store is out of addresses (or the runtime lacks the API): plain memory
        (void)hipGetLastError();
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) {
        if (getenv("COLBWT_ALLOC_LOG")) {
            size_t free_b = 0, total_b = 0;
            (void)hipMemGetInfo(&free_b, &total_b);
            fprintf(stderr, "[colbwt alloc] ! %8.2f GB failed (%s, %s): device reports %.2f GB free\n", bytes * 1e-9, hipGetErrorString(e),
                    ranged ? "granule store" : "hipMalloc", free_b * 1e-9);
        }
        *p = nullptr;
        (void)hipGetLastError();   // the caller reports `e`; do not leave it for a later hipGetLastError
        return e;
    }
    if (b) {
        b->used += bytes;
        if (b->used > b->peak) b->peak = b->used;
    }
    if (getenv("COLBWT_ALLOC_LOG") && bytes >= (64u << 20)) {
        fprintf(stderr, "[colbwt alloc] + %8.2f GB  %p  used %8.2f", bytes * 1e-9, *p, b ? b->used * 1e-9 : 0.0);
        const VmmLast &l = vmm_last();
        if (ranged) fprintf(stderr, "   granules: %llu reused, %llu created in %.3f s, mapped in %.3f s, access set in %.3f s",
                            (unsigned long long)l.reused, (unsigned long long)l.created, l.create_s, l.map_s, l.access_s);
        fprintf(stderr, "\n");
    }
    return hipSuccess;
}

// Bytes an open on this thread could still allocate: what the device reports free, within what is
// left of the budget.  The builders ask before a refinement level is materialised, so a layout that
// cannot fit gives up after a counting pass instead of after tens of GB of allocations.
inline uint64_t dev_available_bytes() {
    size_t free_b = 0, total_b = 0;
    uint64_t avail = ~0ull;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) avail = free_b + vmm_idle_bytes();   // idle granules are the store's to reuse or return
    else (void)hipGetLastError();
    const DevBudget *b = current_budget();
    if (b && b->limit != ~0ull) avail = std::min<uint64_t>(avail, b->limit > b->used ? b->limit - b->used : 0);
    return avail;
}

inline void dev_free(void *p, uint64_t bytes) {
    if (!p) return;
    if (getenv("COLBWT_ALLOC_LOG") && bytes >= (64u << 20)) fprintf(stderr, "[colbwt alloc] - %8.2f GB  %p\n", bytes * 1e-9, p);
    if (!vmm_free(p)) (void)hipFree(p);
    DevBudget *b = current_budget();
    if (b) b->used = b->used > bytes ? b->used - bytes : 0;
}

// COLBWT_LOAD_TIMING=1: the stages of an index open with their wall times on stderr (the device is
// synchronised at every lap, so the times are the stages' own).
struct LoadClock {
    bool on;
    double t_prev;
    static double now() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec + ts.tv_nsec * 1e-9;
    }
    LoadClock() : on(getenv("COLBWT_LOAD_TIMING") != nullptr), t_prev(now()) {}
    void lap(const char *what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const double t = now();
        fprintf(stderr, "[colbwt load] %-36s %8.3f s\n", what, t - t_prev);
        t_prev = t;
    }
};

// Owner of one device allocation.
class DevPtr {
public:
    DevPtr() = default;
    DevPtr(const DevPtr &) = delete;
    DevPtr &operator=(const DevPtr &) = delete;
    DevPtr(DevPtr &&o) noexcept : p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; o.bytes_ = 0; }
    DevPtr &operator=(DevPtr &&o) noexcept {
        if (this != &o) {
            reset();
            p_ = o.p_;
            bytes_ = o.bytes_;
            o.p_ = nullptr;
            o.bytes_ = 0;
        }
        return *this;
    }
    ~DevPtr() { reset(); }
    hipError_t alloc(uint64_t bytes) {
        reset();
        const hipError_t e = dev_alloc(&p_, bytes);
        if (e == hipSuccess) bytes_ = bytes;
        return e;
    }
    void reset() {
        dev_free(p_, bytes_);
        p_ = nullptr;
        bytes_ = 0;
    }
    // hands the allocation over to a longer-lived owner (which frees it with dev_free)
    void *release(uint64_t *bytes = nullptr) {
        void *p = p_;
        if (bytes) *bytes = bytes_;
        p_ = nullptr;
        bytes_ = 0;
        return p;
    }
    template <typename T>
    T *as() const { return static_cast<T *>(p_); }
    void *get() const { return p_; }
    uint64_t bytes() const { return bytes_; }

private:
    void *p_ = nullptr;
    uint64_t bytes_ = 0;
};

}  // namespace colbwt
