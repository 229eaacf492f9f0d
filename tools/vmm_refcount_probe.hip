// vmm_refcount_probe -- who owns a granule of physical memory across hipMemMap / hipMemUnmap?
// Prints the device's free memory (MiB, relative to the start) after every step of a
// create / (map, unmap) x 3 / release sequence, without and with hipMemRetainAllocationHandle
// before each unmap.   csrc/dev_vmm.h keeps granules across unmap -> map: this says what it must do.
#include <hip/hip_runtime.h>
#include <stdio.h>

static long long base_free = 0;
static void show(const char *what, hipError_t e) {
    size_t f = 0, t = 0;
    (void)hipMemGetInfo(&f, &t);
    if (!base_free) base_free = (long long)f;
    printf("  %-44s %-18s free %+lld MiB\n", what, hipGetErrorName(e), ((long long)f - base_free) / (1 << 20));
}

int main() {
    (void)hipFree(nullptr);
    const size_t gran = 256u << 20;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int retain = 0; retain < 2; ++retain) {
        printf("%s\n", retain ? "with hipMemRetainAllocationHandle before each unmap:" : "plain:");
        show("start", hipSuccess);
        void *va = nullptr;
        show("reserve 256 MiB of addresses", hipMemAddressReserve(&va, gran, gran, nullptr, 0));
        hipMemGenericAllocationHandle_t h;
        show("create", hipMemCreate(&h, gran, &prop, 0));
        for (int c = 0; c < 3; ++c) {
            show("map", hipMemMap(va, gran, 0, h, 0));
            show("set access", hipMemSetAccess(va, gran, &acc, 1));
            show("memset through the mapping", hipMemset(va, c + 1, gran));
            (void)hipDeviceSynchronize();
            if (retain) {
                hipMemGenericAllocationHandle_t h2;
                hipError_t e = hipMemRetainAllocationHandle(&h2, va);
                show(h2 == h ? "retain (same handle)" : "retain (ANOTHER handle)", e);
                if (e == hipSuccess) h = h2;
            }
            show("unmap", hipMemUnmap(va, gran));
        }
        show("release", hipMemRelease(h));
        show("free the addresses", hipMemAddressFree(va, gran));
    }
    return 0;
}
