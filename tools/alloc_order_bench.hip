// alloc_order_bench -- is a large hipMalloc slow by itself, or slow because it waits for memory this
// (or the previous) process has just freed?  An index open spends 4.3 s in the one hipMalloc of the
// final table, after having allocated and freed the refinement levels' arrays.
//   usage: alloc_order_bench <sequence>    sequence of  a<GiB> (hipMalloc, keep)  v<GiB> (the granule store of csrc/dev_vmm.h, keep)
//                                           f (free the oldest kept)  s<seconds> (sleep)  m (hipMemset the newest, whole)
//                                           w (fill the newest with a pattern of its own)  c (check every kept array that was filled)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>

#include <deque>

#include "../col-bwt_amd/csrc/dev_vmm.h"

__global__ void fill_kernel(uint32_t *p, size_t n, uint32_t tag) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = tag ^ (uint32_t)i;
}
__global__ void check_kernel(const uint32_t *p, size_t n, uint32_t tag, unsigned long long *bad) {
    unsigned long long b = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += p[i] != (tag ^ (uint32_t)i);
    if (b) atomicAdd(bad, b);
}

static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

int main(int argc, char **argv) {
    (void)hipFree(nullptr);
    std::deque<void *> kept;
    std::deque<size_t> kept_bytes;
    std::deque<uint32_t> kept_tag;
    uint32_t next_tag = 1;
    unsigned long long *d_bad = nullptr;
    (void)hipMalloc(&d_bad, 8);
    colbwt::VmmScope scope;
    printf("{\"sequence\": [");
    for (int a = 1; a < argc; ++a) {
        const char op = argv[a][0];
        const double v = atof(argv[a] + 1);
        double t = now();
        int ok = 1;
        if (op == 'a') {
            void *p = nullptr;
            ok = hipMalloc(&p, (size_t)(v * (1ull << 30))) == hipSuccess;
            (void)hipDeviceSynchronize();
            if (ok) { kept.push_back(p); kept_bytes.push_back((size_t)(v * (1ull << 30))); kept_tag.push_back(0); }
        } else if (op == 'v') {
            void *p = nullptr;
            ok = colbwt::vmm_alloc(&p, (uint64_t)(v * (1ull << 30))) == hipSuccess;
            (void)hipDeviceSynchronize();
            if (ok) { kept.push_back(p); kept_bytes.push_back((size_t)(v * (1ull << 30))); kept_tag.push_back(0); }
        } else if (op == 'm') {
            if (!kept.empty()) ok = hipMemset(kept.back(), 0x5a, kept_bytes.back()) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
        } else if (op == 'w') {
            if (!kept.empty()) {
                kept_tag.back() = next_tag++ * 0x9E3779B1u;
                hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t *)kept.back(), kept_bytes.back() / 4, kept_tag.back());
                ok = hipDeviceSynchronize() == hipSuccess;
            }
        } else if (op == 'c') {
            unsigned long long bad = 0;
            (void)hipMemset(d_bad, 0, 8);
            for (size_t q = 0; q < kept.size(); ++q)
                if (kept_tag[q]) hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, (const uint32_t *)kept[q], kept_bytes[q] / 4, kept_tag[q], d_bad);
            ok = hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost) == hipSuccess && bad == 0;
            if (bad) fprintf(stderr, "check: %llu words differ\n", bad);
        } else if (op == 'f') {
            if (!kept.empty()) {
                if (!colbwt::vmm_free(kept.front())) ok = hipFree(kept.front()) == hipSuccess;
                kept.pop_front();
                kept_bytes.pop_front();
                kept_tag.pop_front();
            }
        } else if (op == 's') {
            usleep((useconds_t)(v * 1e6));
        }
        printf("%s{\"op\": \"%s\", \"ok\": %d, \"s\": %.3f}", a > 1 ? ", " : "", argv[a], ok, now() - t);
    }
    printf("]}\n");
    return 0;
}
