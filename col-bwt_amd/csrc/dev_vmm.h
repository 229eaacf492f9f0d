// dev_vmm.h -- large device arrays as virtual ranges over a store of physical granules.
//
// Why: on this driver memory that has been FREED is wiped before it is handed out again, at ~33 GB/s,
// and an allocation that needs any of it waits for the whole wipe (tools/alloc_order_bench.hip:
// hipMalloc(170 GiB) returns in < 1 ms on untouched memory and in 5.1 s right after a hipFree of the
// same size; profiles/r03w_alloc_order.jsonl).  An index open allocates and frees ~150 GB of
// refinement-level arrays on its way to a 173 GB table (profiles/r03w_open_alloc_log.txt): the
// 4.3-5.4 s "lines allocated" stage of the open, and the 1-1.5 s stalls of its levels on a box that
// had work before, were waits for those wipes, not the cost of the allocation.
//
// So arrays of 512 MiB and more are ranges of one reservation of addresses (hipMemAddressReserve, once
// per device) mapped onto granules of physical memory (hipMemCreate, 256 MiB each).  Freeing an array unmaps it and puts its
// granules on the device's idle list; the next array takes granules from there first, whatever its
// size -- level L + 1's larger arrays run on level L - 1's memory plus a few new granules, and the
// final table on everything the levels held.  Nothing goes back to the driver while an open is
// running on the device (VmmScope), so nothing is wiped and nothing waits; the idle granules are
// released when the last open on the device returns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

#include <mutex>
#include <unordered_map>
#include <vector>

namespace colbwt {

constexpr int kVmmMaxDevices = 16;

struct VmmStore {
    struct Block {
        uint64_t bytes = 0;   // mapped size: whole granules
        int device = 0;
        std::vector<hipMemGenericAllocationHandle_t> granules;
    };
    // One reservation of addresses per device, made at the first use and kept; an array is the next
    // stretch of it, and NO ADDRESS IS USED TWICE: on this runtime a range that is unmapped and mapped
    // again -- onto other granules -- keeps translating to the old ones (tools/alloc_order_bench.hip
    // `w` / `c`: an array in a reused range reads back another array's pattern, whether the range
    // comes from hipMemAddressFree + hipMemAddressReserve or from inside one reservation;
    // profiles/r03w_granule_integrity*.jsonl).  Granules moving to NEW addresses are fine.  Addresses
    // are plentiful (a C2 open walks through ~0.4 TB of a 32 TB reservation); when they run out, arrays
    // come from hipMalloc as before.
    struct Arena {
        char *base = nullptr;
        uint64_t granules = 0, next = 0;
        bool failed = false;
    };
    std::mutex mu;
    std::unordered_map<void *, Block> live;                             // by base address
    std::vector<hipMemGenericAllocationHandle_t> idle[kVmmMaxDevices];  // created, mapped nowhere
    Arena arena[kVmmMaxDevices];
    int scopes[kVmmMaxDevices] = {};                                    // opens running on the device
    uint64_t granule = 256ull << 20;
    uint64_t min_bytes = 512ull << 20;
    bool on = true;

    VmmStore() {
        // COLBWT_VMM=0: every array through hipMalloc, as before (A/B runs).  COLBWT_VMM_GRANULE_MB /
        // COLBWT_VMM_MIN_MB: the tests shrink both so that small tables go through the store.
        if (const char *e = getenv("COLBWT_VMM")) on = atoi(e) != 0;
        if (const char *e = getenv("COLBWT_VMM_GRANULE_MB")) { const long v = atol(e); if (v >= 2) granule = ((uint64_t)v & ~1ull) << 20; }
        if (const char *e = getenv("COLBWT_VMM_MIN_MB")) { const long v = atol(e); if (v >= 0) min_bytes = (uint64_t)v << 20; }
    }
};

inline VmmStore &vmm_store() {
    static VmmStore s;
    return s;
}

inline bool vmm_takes(uint64_t bytes) {
    const VmmStore &s = vmm_store();
    return s.on && bytes >= s.min_bytes && bytes > 0;
}

// bytes of idle granules on the current device: an allocation may count on them
inline uint64_t vmm_idle_bytes() {
    VmmStore &s = vmm_store();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kVmmMaxDevices) return 0;
    std::lock_guard<std::mutex> g(s.mu);
    return (uint64_t)s.idle[dev].size() * s.granule;
}

namespace vmm_detail {
inline void release_all(std::vector<hipMemGenericAllocationHandle_t> &v) {
    for (auto h : v) (void)hipMemRelease(h);
    v.clear();
}

// n granules of addresses on `dev`, never handed out before; nullptr: no reservation, or used up.  s.mu held.
inline char *take_range(VmmStore &s, int dev, uint64_t n) {
    VmmStore::Arena &a = s.arena[dev];
    if (!a.base && !a.failed) {
        for (uint64_t tb : {32ull, 8ull, 2ull}) {
            void *va = nullptr;
            if (hipMemAddressReserve(&va, tb << 40, s.granule, nullptr, 0) == hipSuccess && va) {
                a.base = (char *)va;
                a.granules = (tb << 40) / s.granule;
                break;
            }
            (void)hipGetLastError();
        }
        a.failed = a.base == nullptr;
    }
    if (!a.base || a.granules - a.next < n) return nullptr;
    char *p = a.base + a.next * s.granule;
    a.next += n;
    return p;
}
}  // namespace vmm_detail

// what the last vmm_alloc on this thread did (COLBWT_ALLOC_LOG prints it)
struct VmmLast {
    uint64_t reused = 0, created = 0;
    double create_s = 0, map_s = 0, access_s = 0;
};
inline VmmLast &vmm_last() {
    static thread_local VmmLast l;
    return l;
}
inline double vmm_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

// hipMalloc's contract: *p set on success, hipErrorOutOfMemory when the device has no room.
inline hipError_t vmm_alloc(void **p, uint64_t bytes) {
    *p = nullptr;
    VmmStore &s = vmm_store();
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kVmmMaxDevices) return hipErrorInvalidDevice;
    const uint64_t n = (bytes + s.granule - 1) / s.granule, size = n * s.granule;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;

    void *va = nullptr;
    {
        std::lock_guard<std::mutex> g(s.mu);
        va = vmm_detail::take_range(s, dev, n);
    }
    if (!va) return hipErrorOutOfMemory;
    VmmStore::Block blk;
    blk.bytes = size;
    blk.device = dev;
    blk.granules.reserve(n);
    uint64_t mapped = 0;
    VmmLast &last = vmm_last();
    last = VmmLast();
    for (uint64_t i = 0; i < n && e == hipSuccess; ++i) {
        hipMemGenericAllocationHandle_t h;
        bool have = false;
        {
            std::lock_guard<std::mutex> g(s.mu);
            if (!s.idle[dev].empty()) { h = s.idle[dev].back(); s.idle[dev].pop_back(); have = true; }
        }
        double t = vmm_now();
        if (!have) e = hipMemCreate(&h, s.granule, &prop, 0);
        if (e != hipSuccess) break;
        (have ? last.reused : last.created) += 1;
        last.create_s += vmm_now() - t;
        blk.granules.push_back(h);
        t = vmm_now();
        e = hipMemMap((char *)va + i * s.granule, s.granule, 0, h, 0);
        last.map_s += vmm_now() - t;
        if (e == hipSuccess) mapped = i + 1;
    }
    if (e == hipSuccess) {
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        const double t = vmm_now();
        e = hipMemSetAccess(va, size, &acc, 1);
        last.access_s = vmm_now() - t;
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        for (uint64_t i = 0; i < mapped; ++i) (void)hipMemUnmap((char *)va + i * s.granule, s.granule);
        std::lock_guard<std::mutex> g(s.mu);
        if (s.scopes[dev] > 0) s.idle[dev].insert(s.idle[dev].end(), blk.granules.begin(), blk.granules.end());
        else vmm_detail::release_all(blk.granules);
        return hipErrorOutOfMemory;
    }
    {
        std::lock_guard<std::mutex> g(s.mu);
        s.live.emplace(va, std::move(blk));
    }
    *p = va;
    return hipSuccess;
}

// true when p was one of the store's arrays (and is gone now); false: not ours, the caller hipFrees.
// Like hipFree, waits for the device first: kernels still queued may be using the array.
inline bool vmm_free(void *p) {
    VmmStore &s = vmm_store();
    VmmStore::Block blk;
    {
        std::lock_guard<std::mutex> g(s.mu);
        auto it = s.live.find(p);
        if (it == s.live.end()) return false;
        blk = std::move(it->second);
        s.live.erase(it);
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != blk.device) (void)hipSetDevice(blk.device);
    (void)hipDeviceSynchronize();
    for (uint64_t i = 0; i < blk.granules.size(); ++i) (void)hipMemUnmap((char *)p + i * s.granule, s.granule);
    {
        std::lock_guard<std::mutex> g(s.mu);
        if (s.scopes[blk.device] > 0) s.idle[blk.device].insert(s.idle[blk.device].end(), blk.granules.begin(), blk.granules.end());
        else vmm_detail::release_all(blk.granules);
    }
    if (cur != blk.device) (void)hipSetDevice(cur);
    (void)hipGetLastError();
    return true;
}

// Idle granules go back to the driver (an allocation outside the store needs the room, or the last
// open on the device has returned).
inline void vmm_trim(int dev) {
    if (dev < 0 || dev >= kVmmMaxDevices) return;
    VmmStore &s = vmm_store();
    std::vector<hipMemGenericAllocationHandle_t> v;
    {
        std::lock_guard<std::mutex> g(s.mu);
        v.swap(s.idle[dev]);
    }
    vmm_detail::release_all(v);
}

// While one is alive on a device, freed granules are kept for the next array.
struct VmmScope {
    int dev = -1;
    VmmScope() {
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kVmmMaxDevices) { dev = -1; return; }
        VmmStore &s = vmm_store();
        std::lock_guard<std::mutex> g(s.mu);
        ++s.scopes[dev];
    }
    ~VmmScope() {
        if (dev < 0) return;
        VmmStore &s = vmm_store();
        bool last;
        {
            std::lock_guard<std::mutex> g(s.mu);
            last = --s.scopes[dev] == 0;
        }
        if (last) vmm_trim(dev);
    }
    VmmScope(const VmmScope &) = delete;
    VmmScope &operator=(const VmmScope &) = delete;
};

}  // namespace colbwt
