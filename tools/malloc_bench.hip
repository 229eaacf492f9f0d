// malloc_bench -- what a large device allocation costs by API (an index open of a C2 line-row index
// spends 4 s in one hipMalloc of 134 GB).   usage: malloc_bench <GiB>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

int main(int argc, char **argv) {
    const size_t gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 64, bytes = gib << 30;
    hipFree(nullptr);
    for (int rep = 0; rep < 2; ++rep) {
        void *p = nullptr;
        double t = now();
        hipError_t e = hipMalloc(&p, bytes);
        hipDeviceSynchronize();
        const double t_alloc = now() - t;
        t = now();
        if (e == hipSuccess) hipFree(p);
        printf("{\"api\": \"hipMalloc\", \"GiB\": %zu, \"rep\": %d, \"ok\": %d, \"alloc_s\": %.3f, \"free_s\": %.3f}\n", gib, rep, e == hipSuccess, t_alloc, now() - t);
    }
    for (int rep = 0; rep < 2; ++rep) {
        void *p = nullptr;
        hipStream_t s;
        hipStreamCreate(&s);
        double t = now();
        hipError_t e = hipMallocAsync(&p, bytes, s);
        hipStreamSynchronize(s);
        const double t_alloc = now() - t;
        t = now();
        if (e == hipSuccess) { hipFreeAsync(p, s); hipStreamSynchronize(s); }
        printf("{\"api\": \"hipMallocAsync\", \"GiB\": %zu, \"rep\": %d, \"ok\": %d, \"alloc_s\": %.3f, \"free_s\": %.3f}\n", gib, rep, e == hipSuccess, t_alloc, now() - t);
        hipStreamDestroy(s);
    }
    {   // virtual memory management: reserve, create, map, set access
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        const size_t sz = (bytes + gran - 1) / gran * gran;
        for (int rep = 0; rep < 2; ++rep) {
            void *va = nullptr;
            hipMemGenericAllocationHandle_t h;
            double t = now();
            hipError_t e = hipMemAddressReserve(&va, sz, 0, nullptr, 0);
            if (e == hipSuccess) e = hipMemCreate(&h, sz, &prop, 0);
            const double t_create = now() - t;
            if (e == hipSuccess) e = hipMemMap(va, sz, 0, h, 0);
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            if (e == hipSuccess) e = hipMemSetAccess(va, sz, &acc, 1);
            hipDeviceSynchronize();
            const double t_alloc = now() - t;
            t = now();
            if (e == hipSuccess) { hipMemUnmap(va, sz); hipMemRelease(h); hipMemAddressFree(va, sz); }
            printf("{\"api\": \"hipMemCreate+Map\", \"GiB\": %zu, \"rep\": %d, \"ok\": %d, \"granularity\": %zu, \"create_s\": %.3f, \"alloc_s\": %.3f, \"free_s\": %.3f}\n",
                   gib, rep, e == hipSuccess, gran, t_create, t_alloc, now() - t);
        }
    }
    return 0;
}
