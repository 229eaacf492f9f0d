// tests/emu/emu_runtime.cpp -- block scheduler of the SIMT emulator (test only).
// One OS thread per GPU thread of a block, taken from a pool that lives for the process (a kernel
// of the index build has thousands of blocks: creating 256 threads for each of them was most
// of the emulator's run time); blocks run one after the other.
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "hip/hip_runtime.h"

namespace emu {
thread_local Ctx ctx;
std::mutex atomic_mu;

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

namespace {
struct Pool {
    std::mutex mu;
    std::condition_variable go, done_cv;
    std::vector<std::thread> workers;
    uint64_t generation = 0;
    uint32_t active = 0, remaining = 0;
    // the block being run
    const std::function<void()> *body = nullptr;
    dim3 grid, block;
    uint32_t block_id = 0;
    Barrier *blk = nullptr;
    std::vector<Barrier> *waves = nullptr;

    void worker(uint32_t t) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                go.wait(lk, [&] { return generation != seen; });
                seen = generation;
                if (t >= active) continue;
            }
            ctx.tid = dim3(t);
            ctx.bid = dim3(block_id);
            ctx.bdim = block;
            ctx.gdim = grid;
            ctx.block = blk;
            ctx.wave = &(*waves)[t / 64];
            (*body)();
            ctx.wave->leave();
            ctx.block->leave();
            std::lock_guard<std::mutex> lk(mu);
            if (--remaining == 0) done_cv.notify_all();
        }
    }
    void ensure(uint32_t n) {
        while (workers.size() < n) {
            const uint32_t t = (uint32_t)workers.size();
            workers.emplace_back([this, t] { worker(t); });
            workers.back().detach();
        }
    }
    void run_block(const std::function<void()> &b, dim3 g, dim3 bl, uint32_t id, Barrier *bar, std::vector<Barrier> *wv) {
        std::unique_lock<std::mutex> lk(mu);
        body = &b;
        grid = g;
        block = bl;
        block_id = id;
        blk = bar;
        waves = wv;
        active = remaining = bl.x;
        ++generation;
        go.notify_all();
        done_cv.wait(lk, [&] { return remaining == 0; });
    }
};
Pool &pool() {
    static Pool *p = new Pool();   // never destroyed: its workers are detached
    return *p;
}
std::mutex launch_mu;              // one kernel at a time (host threads of the product may launch concurrently)
}  // namespace

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    std::lock_guard<std::mutex> one(launch_mu);
    const uint32_t nthreads = block.x;
    const uint32_t nwaves = (nthreads + 63) / 64;
    Pool &P = pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        P.ensure(nthreads);
    }
    for (uint32_t b = 0; b < grid.x; ++b) {
        Barrier blk;
        std::vector<Barrier> waves(nwaves);
        blk.reset(nthreads);
        for (uint32_t w = 0; w < nwaves; ++w) waves[w].reset(std::min(64u, nthreads - w * 64));
        P.run_block(body, grid, block, b, &blk, &waves);
    }
}
}  // namespace emu
