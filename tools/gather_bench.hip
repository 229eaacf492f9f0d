// gather_bench.hip -- calibration micro-benchmark for the query kernel's
// access pattern: dependent random 16-byte loads from a table far larger than
// the Infinity Cache.  Gives (a) the chip's achievable random-line rate, the
// practical ceiling the LF step is measured against, and (b) a known load
// count to calibrate rocprofv3's FETCH_SIZE for this pattern
// (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access
// pattern").   usage: gather_bench <table_MiB> <lanes> <steps> <mode> [reps]
//   mode 0: one dependent 16 B load per step (pure pointer chase)
//   mode 1: 16 B at the random row + 16 B at row+1 (the fast-forward shape)
//   mode 2: one dependent 64 B-aligned 64 B load (4 x uint4) per step
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

template <int MODE>
__global__ __launch_bounds__(256) void chase(const uint4 *__restrict__ t, uint64_t rows, uint32_t steps, uint32_t *out) {
    const uint64_t lane = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t j = (lane * 0xD1342543DE82EF95ull + 12345) % rows;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        if (MODE == 2) j &= ~(uint64_t)3;
        uint4 w = t[j];
        if (MODE == 1) { uint4 w2 = t[j + 1 < rows ? j + 1 : j]; acc += w2.z; }
        if (MODE == 2) { uint4 a = t[j + 1], b = t[j + 2], c = t[j + 3]; acc += a.z + b.z + c.z; }
        acc += w.z;
        j = (((uint64_t)w.x | ((uint64_t)w.y << 32)) + s) % rows;   // next row depends on the loaded data
    }
    out[lane] = acc;
}

int main(int argc, char **argv) {
    const uint64_t mib = argc > 1 ? strtoull(argv[1], 0, 10) : 3072;
    const uint64_t lanes = argc > 2 ? strtoull(argv[2], 0, 10) : 10000000;
    const uint32_t steps = argc > 3 ? atoi(argv[3]) : 150;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    const int reps = argc > 5 ? atoi(argv[5]) : 3;
    const uint64_t rows = mib * 1024 * 1024 / 16;
    uint4 *t; uint32_t *out;
    CK(hipMalloc(&t, rows * 16));
    CK(hipMalloc(&out, lanes * 4 + 1024));
    fill<<<4096, 256>>>(t, rows);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t blocks = (uint32_t)((lanes + 255) / 256);
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        if (mode == 0) chase<0><<<blocks, 256>>>(t, rows, steps, out);
        else if (mode == 1) chase<1><<<blocks, 256>>>(t, rows, steps, out);
        else chase<2><<<blocks, 256>>>(t, rows, steps, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double loads = (double)blocks * 256 * steps;
        printf("{\"table_MiB\": %llu, \"lanes\": %llu, \"steps\": %u, \"mode\": %d, \"ms\": %.3f, \"Gsteps_per_s\": %.3f, \"loads\": %.0f}\n",
               (unsigned long long)mib, (unsigned long long)lanes, steps, mode, ms, loads / ms / 1e6, loads);
    }
    return 0;
}
