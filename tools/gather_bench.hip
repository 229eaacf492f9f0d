// gather_bench.hip -- calibration micro-benchmark for the query kernel's
// access pattern: dependent random 16-byte loads from a table far larger than
// the Infinity Cache.  Gives (a) the chip's achievable random-row rate, the
// practical ceiling the LF step is measured against, (b) a known load count to
// calibrate rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ for this pattern
// (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access
// pattern"), and (c) the HBM fetch granularity per cache policy.
//   usage: gather_bench <table_MiB> <lanes> <steps> <mode> [reps] [alloc]
//   mode 0: one dependent 16 B load per step (pure pointer chase)
//   mode 1: 16 B at the random row + 16 B at row+1 (the fast-forward shape)
//   mode 2: 4 x 16 B covering one aligned 64 B block
//   mode 3: 16 B in each 64 B half of one aligned 128 B line
//   mode 4: 16 B in two adjacent 128 B lines
//   mode 5..8: as mode 0 with cache-policy bits: nt | sc1 | sc0 sc1 | sc0 sc1 nt
//   mode 9: all 8 x 16 B of one aligned 128 B line (the fat-row shape)
//   mode 10: 2 x 16 B = one aligned 32 B row (the three-step row shape)
//   alloc 0: hipMalloc, 1: hipDeviceMallocUncached, 2: hipDeviceMallocFinegrained,
//         3: hipDeviceMallocContiguous (physically contiguous: larger TLB fragments?)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ void fill(uint4 *t, uint64_t rows) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < rows; i += stride) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull + 0x1234567;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        t[i] = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)i, 0);
    }
}

template <int POLICY>
__device__ __forceinline__ uint4 load_policy(const uint4 *p) {
    uint4 w;
    if (POLICY == 5) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
    else if (POLICY == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
    else if (POLICY == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
    else if (POLICY == 8) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
    else w = *p;
    return w;
}

template <int MODE>
__global__ __launch_bounds__(256) void chase(const uint4 *__restrict__ t, uint64_t rows, uint32_t steps, uint32_t *out) {
    const uint64_t lane = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t j = (lane * 0xD1342543DE82EF95ull + 12345) % rows;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        if (MODE == 2) j &= ~(uint64_t)3;
        if (MODE == 3 || MODE == 9) j &= ~(uint64_t)7;
        if (MODE == 10) j &= ~(uint64_t)1;
        if (MODE == 4) { j &= ~(uint64_t)7; if (j + 16 > rows) j = 0; }
        uint4 w = load_policy<MODE>(t + j);
        if (MODE == 1) { uint4 w2 = t[j + 1 < rows ? j + 1 : j]; acc += w2.z; }
        if (MODE == 2) { uint4 a = t[j + 1], b = t[j + 2], c = t[j + 3]; acc += a.z + b.z + c.z; }
        if (MODE == 3) { uint4 a = t[j + 4]; acc += a.z; }
        if (MODE == 10) { uint4 a = t[j + 1]; acc += a.z; }
        if (MODE == 9) {
            uint4 a = t[j + 1], b = t[j + 2], c = t[j + 3], d = t[j + 4], e = t[j + 5], f = t[j + 6], h = t[j + 7];
            acc += a.z + b.z + c.z + d.z + e.z + f.z + h.z;
        }
        if (MODE == 4) { uint4 a = t[j + 8]; acc += a.z; }
        acc += w.z;
        j = (((uint64_t)w.x | ((uint64_t)w.y << 32)) + s) % rows;   // next row depends on the loaded data
    }
    out[lane] = acc;
}

template <int MODE>
void run(uint32_t blocks, const uint4 *t, uint64_t rows, uint32_t steps, uint32_t *out) {
    chase<MODE><<<blocks, 256>>>(t, rows, steps, out);
}

int main(int argc, char **argv) {
    const uint64_t mib = argc > 1 ? strtoull(argv[1], 0, 10) : 3072;
    const uint64_t lanes = argc > 2 ? strtoull(argv[2], 0, 10) : 10000000;
    const uint32_t steps = argc > 3 ? atoi(argv[3]) : 150;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    const int reps = argc > 5 ? atoi(argv[5]) : 3;
    const int alloc = argc > 6 ? atoi(argv[6]) : 0;
    const uint64_t rows = mib * 1024 * 1024 / 16;
    uint4 *t; uint32_t *out;
    if (alloc == 1) CK(hipExtMallocWithFlags((void **)&t, rows * 16, hipDeviceMallocUncached));
    else if (alloc == 2) CK(hipExtMallocWithFlags((void **)&t, rows * 16, hipDeviceMallocFinegrained));
    else if (alloc == 3) CK(hipExtMallocWithFlags((void **)&t, rows * 16, hipDeviceMallocContiguous));
    else CK(hipMalloc(&t, rows * 16));
    CK(hipMalloc(&out, lanes * 4 + 1024));
    fill<<<4096, 256>>>(t, rows);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t blocks = (uint32_t)((lanes + 255) / 256);
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        switch (mode) {
            case 0: run<0>(blocks, t, rows, steps, out); break;
            case 1: run<1>(blocks, t, rows, steps, out); break;
            case 2: run<2>(blocks, t, rows, steps, out); break;
            case 3: run<3>(blocks, t, rows, steps, out); break;
            case 4: run<4>(blocks, t, rows, steps, out); break;
            case 5: run<5>(blocks, t, rows, steps, out); break;
            case 6: run<6>(blocks, t, rows, steps, out); break;
            case 7: run<7>(blocks, t, rows, steps, out); break;
            case 9: run<9>(blocks, t, rows, steps, out); break;
            case 10: run<10>(blocks, t, rows, steps, out); break;
            default: run<8>(blocks, t, rows, steps, out); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double loads = (double)blocks * 256 * steps;
        printf("{\"table_MiB\": %llu, \"lanes\": %llu, \"steps\": %u, \"mode\": %d, \"alloc\": %d, \"ms\": %.3f, \"Gsteps_per_s\": %.3f, \"steps_total\": %.0f}\n",
               (unsigned long long)mib, (unsigned long long)lanes, steps, mode, alloc, ms, loads / ms / 1e6, loads);
    }
    return 0;
}
