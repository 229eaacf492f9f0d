// tests/emu/hip/hip_runtime.h -- a tiny SIMT *emulator* of the HIP vocabulary
// the product sources use, so that the very same .hip files can be compiled
// with g++ and exercised on a CPU-only box (with ASan/UBSan) by the
// `-m "not gpu"` test tier.
//
// TEST INFRASTRUCTURE ONLY.  It is never built by __graft_entry__.build(),
// never linked into libcolbwt.so and never shipped: the product has no host
// path.  One OS thread per GPU thread of a block, blocks run one after the
// other; __syncthreads / __ballot are real rendezvous among those threads.
#pragma once
#include <stdint.h>
#include <sys/mman.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 {
    uint32_t x, y, z;
    dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorInvalidDevice = 101 };
typedef struct emu_stream *hipStream_t;
typedef struct emu_event { double t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipStreamNonBlocking = 1 };

namespace emu {
struct Barrier {
    std::mutex mu;
    std::condition_variable cv;
    uint32_t expected = 0, arrived = 0, generation = 0;
    uint64_t ballot_acc = 0, ballot_out = 0;
    uint64_t xchg[64] = {};   // __shfl_* hand-over of the wave
    void reset(uint32_t n) { expected = n; arrived = 0; generation = 0; ballot_acc = 0; }
    // returns the OR of `bits` over all participants of this generation
    uint64_t arrive(uint64_t bits) {
        std::unique_lock<std::mutex> lk(mu);
        ballot_acc |= bits;
        if (++arrived >= expected) {
            ballot_out = ballot_acc; ballot_acc = 0; arrived = 0; ++generation;
            cv.notify_all();
            return ballot_out;
        }
        const uint32_t gen = generation;
        cv.wait(lk, [&] { return generation != gen; });
        return ballot_out;
    }
    void leave() {  // a thread returned from the kernel: stop waiting for it
        std::unique_lock<std::mutex> lk(mu);
        if (expected) --expected;
        if (expected && arrived >= expected) {
            ballot_out = ballot_acc; ballot_acc = 0; arrived = 0; ++generation;
            cv.notify_all();
        }
    }
};
struct Ctx { dim3 tid, bid, bdim, gdim; Barrier *block; Barrier *wave; };
extern thread_local Ctx ctx;
extern std::mutex atomic_mu;
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
double now_ms();
}  // namespace emu

#define threadIdx (emu::ctx.tid)
#define blockIdx (emu::ctx.bid)
#define blockDim (emu::ctx.bdim)
#define gridDim (emu::ctx.gdim)

static inline void __syncthreads() { emu::ctx.block->arrive(0); }
static inline unsigned long long __ballot(int pred) {
    return emu::ctx.wave->arrive(pred ? (1ull << (emu::ctx.tid.x & 63)) : 0ull);
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
// Wave shuffles (all 64 lanes take part): values pass through the wave's exchange slots.
template <typename T>
static inline T emu_shfl(T v, int src_lane) {
    emu::Barrier *w = emu::ctx.wave;
    w->xchg[emu::ctx.tid.x & 63] = (uint64_t)v;
    w->arrive(0);
    const T r = (T)w->xchg[src_lane & 63];
    w->arrive(0);
    return r;
}
template <typename T>
static inline T __shfl_xor(T v, int mask) { return emu_shfl(v, (int)(emu::ctx.tid.x & 63) ^ mask); }
template <typename T>
static inline T __shfl_down(T v, int delta) {
    const int lane = (int)(emu::ctx.tid.x & 63);
    return emu_shfl(v, lane + delta < 64 ? lane + delta : lane);
}
// Cross-lane hand-over through LDS inside one wave: the 64 emulated lanes are OS threads, so the
// compiler-only barrier of the GPU build is a real one here.
#define __builtin_amdgcn_wave_barrier() ((void)emu::ctx.wave->arrive(0))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_waitcnt(imm) ((void)0)
// v_perm_b32: result byte i = byte (sel >> 8i) & 0xFF of {s0 (4..7), s1 (0..3)}
static inline uint32_t emu_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    const uint64_t both = ((uint64_t)s0 << 32) | s1;
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) out |= (uint32_t)((both >> (8 * ((sel >> (8 * i)) & 7u))) & 0xFFu) << (8 * i);
    return out;
}
#define __builtin_amdgcn_perm(s0, s1, sel) emu_perm((s0), (s1), (sel))
// LDS-DMA: `size` bytes per lane from the lane's global address to (wave-uniform LDS base) + lane * size
#define __builtin_amdgcn_global_load_lds(gptr, ldsptr, size, offset, aux) \
    memcpy(reinterpret_cast<char *>(ldsptr) + (size_t)(emu::ctx.tid.x & 63) * (size), reinterpret_cast<const char *>(gptr) + (offset), (size))
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) {
    std::lock_guard<std::mutex> g(emu::atomic_mu); uint32_t o = *p; *p = o | v; return o;
}
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) {
    std::lock_guard<std::mutex> g(emu::atomic_mu); uint32_t o = *p; *p = o + v; return o;
}
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) {
    std::lock_guard<std::mutex> g(emu::atomic_mu); unsigned long long o = *p; *p = o + v; return o;
}
static inline uint32_t atomicMax(uint32_t *p, uint32_t v) {
    std::lock_guard<std::mutex> g(emu::atomic_mu); uint32_t o = *p; if (v > o) *p = v; return o;
}
static inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) {
    std::lock_guard<std::mutex> g(emu::atomic_mu); unsigned long long o = *p; if (v > o) *p = v; return o;
}
static inline uint32_t atomicMin(uint32_t *p, uint32_t v) {
    std::lock_guard<std::mutex> g(emu::atomic_mu); uint32_t o = *p; if (v < o) *p = v; return o;
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })

static inline uint32_t __builtin_amdgcn_alignbyte(uint32_t hi, uint32_t lo, uint32_t r) {   // v_alignbyte_b32
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (r & 3u)));
}
static inline const char *hipGetErrorString(hipError_t) { return "emu error"; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
using std::max;
using std::min;
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
enum { hipDeviceAttributeMultiprocessorCount = 1 };
static inline hipError_t hipDeviceGetAttribute(int *v, int, int) { *v = 2; return hipSuccess; }   // a two-CU chip,
template <typename F> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, F, int, size_t) { *n = 1; return hipSuccess; }   // one block each
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) {
    // exact-size allocations so ASan sees every out-of-bounds device access
    *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = *total_b = (size_t)1 << 40; return hipSuccess; }
// virtual memory management (csrc/dev_vmm.h): a reserved range is anonymous host memory, granules are
// heap objects so that ASan sees a granule released twice or never
typedef struct emu_granule { size_t size; } *hipMemGenericAllocationHandle_t;
enum hipMemAllocationType { hipMemAllocationTypePinned = 1 };
enum hipMemLocationType { hipMemLocationTypeDevice = 1 };
enum hipMemAccessFlags { hipMemAccessFlagsProtReadWrite = 3 };
struct hipMemLocation { hipMemLocationType type; int id; };
struct hipMemAllocationProp { hipMemAllocationType type; hipMemLocation location; };
struct hipMemAccessDesc { hipMemLocation location; hipMemAccessFlags flags; };
namespace emu { inline long &granules_alive() { static long n = 0; return n; } }
static inline hipError_t hipMemAddressReserve(void **p, size_t n, size_t, void *, unsigned long long) {
    void *m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);   // terabytes of addresses, pages on touch
    *p = m == MAP_FAILED ? nullptr : m;
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipMemAddressFree(void *p, size_t n) { munmap(p, n); return hipSuccess; }
static inline hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *h, size_t n, const hipMemAllocationProp *, unsigned long long) {
    std::lock_guard<std::mutex> g(emu::atomic_mu); *h = new emu_granule{n}; ++emu::granules_alive(); return hipSuccess;
}
static inline hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h) {
    std::lock_guard<std::mutex> g(emu::atomic_mu); delete h; --emu::granules_alive(); return hipSuccess;
}
static inline hipError_t hipMemMap(void *, size_t n, size_t, hipMemGenericAllocationHandle_t h, unsigned long long) {
    std::lock_guard<std::mutex> g(emu::atomic_mu); if (n != h->size) abort(); return hipSuccess;
}
static inline hipError_t hipMemSetAccess(void *, size_t, const hipMemAccessDesc *, size_t) { return hipSuccess; }
static inline hipError_t hipMemUnmap(void *, size_t) { return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
enum { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1 };
struct hipPointerAttribute_t { int type; int device; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { a->type = hipMemoryTypeUnregistered; a->device = 0; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emu_event{0}; return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event{0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = emu::now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
