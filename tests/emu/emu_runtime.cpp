// tests/emu/emu_runtime.cpp -- block scheduler of the SIMT emulator (test only).
#include <chrono>

#include "hip/hip_runtime.h"

namespace emu {
thread_local Ctx ctx;
std::mutex atomic_mu;

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    const uint32_t nthreads = block.x;
    const uint32_t nwaves = (nthreads + 63) / 64;
    for (uint32_t b = 0; b < grid.x; ++b) {
        Barrier blk;
        std::vector<Barrier> waves(nwaves);
        blk.reset(nthreads);
        for (uint32_t w = 0; w < nwaves; ++w) waves[w].reset(std::min(64u, nthreads - w * 64));
        std::vector<std::thread> ts;
        ts.reserve(nthreads);
        for (uint32_t t = 0; t < nthreads; ++t) {
            ts.emplace_back([&, t]() {
                ctx.tid = dim3(t);
                ctx.bid = dim3(b);
                ctx.bdim = block;
                ctx.gdim = grid;
                ctx.block = &blk;
                ctx.wave = &waves[t / 64];
                body();
                ctx.wave->leave();
                ctx.block->leave();
            });
        }
        for (auto &th : ts) th.join();
    }
}
}  // namespace emu
