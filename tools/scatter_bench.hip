// scatter_bench -- what a random write REQUEST costs by its shape (the output side of the line-row
// kernel: 64 bytes of PML and 32 of col ids per 32 bases, written by lane groups with 16 bytes each).
//   usage: scatter_bench <table_MiB> <iterations> <mode> [reps]
//   mode 0: 32 B per lane pair      (64 lanes x 16 B = 32 segments per instruction)
//   mode 1: 64 B per lane quad      (16 segments)
//   mode 2: 128 B per 8 lanes       (8 whole lines)
//   mode 3: 256 B per 16 lanes      (4 x two lines)
// Every wave writes `iterations` instructions to pseudo-random aligned segments of the table.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

template <int LANES_PER_SEG>
__global__ __launch_bounds__(256) void scatter(uint4 *table, uint64_t n_seg, uint32_t iters) {
    const uint32_t lane = threadIdx.x & 63u, sub = lane % LANES_PER_SEG, grp = lane / LANES_PER_SEG;
    const uint64_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t seg = mix((wave * iters + it) * 64 + grp) % n_seg;
        table[seg * LANES_PER_SEG + sub] = make_uint4(it, lane, (uint32_t)seg, 0u);
    }
}

int main(int argc, char **argv) {
    const uint64_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 8192;
    const uint32_t iters = argc > 2 ? atoi(argv[2]) : 2000;
    const int mode = argc > 3 ? atoi(argv[3]) : 1;
    const int reps = argc > 4 ? atoi(argv[4]) : 3;
    const uint64_t bytes = mib << 20;
    uint4 *t = nullptr;
    if (hipMalloc(&t, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(t, 0, bytes);
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 3;
    const int lanes_per_seg = 2 << mode;
    const uint64_t n_seg = bytes / (16ull * lanes_per_seg);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        switch (mode) {
            case 0: scatter<2><<<blocks, 256>>>(t, n_seg, iters); break;
            case 1: scatter<4><<<blocks, 256>>>(t, n_seg, iters); break;
            case 2: scatter<8><<<blocks, 256>>>(t, n_seg, iters); break;
            default: scatter<16><<<blocks, 256>>>(t, n_seg, iters); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)blocks * 4 * iters, segs = instr * (64 / lanes_per_seg);
        printf("{\"table_MiB\": %llu, \"mode\": %d, \"segment_bytes\": %d, \"ms\": %.3f, \"Gsegments_per_s\": %.2f, \"GB_per_s\": %.1f}\n",
               (unsigned long long)mib, mode, 16 * lanes_per_seg, ms, segs / ms / 1e6, segs * 16 * lanes_per_seg / ms / 1e6);
    }
    return 0;
}
