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
//   mode 11: one aligned 32 B row per LANE PAIR, one instruction (lanes 2i / 2i+1 fetch the two
//            halves; 32 chains per wave)
//   mode 12: one aligned 32 B row per lane, 64 chains per wave, two instructions, each
//            pair-coalesced (instruction A serves the even lane's row, B the odd lane's; the
//            halves are exchanged with DPP quad_perm [1,0,3,2])
//   mode 13: one aligned 64 B row per lane, 64 chains per wave, four quad-coalesced instructions
//            (instruction q serves the row of quad lane q; pieces exchanged with DPP broadcasts)
//   mode 14: one aligned 64 B row per LANE QUAD, one instruction (16 chains per wave)
//   mode 15: one aligned 128 B row (a whole line) per lane, 64 chains per wave: eight instructions,
//            each serving the rows of one lane of every 8-lane group (8 distinct lines per
//            instruction, every byte used), registers -> ds_write_b128 -> LDS, rows read back
//            from LDS (the fat-row shape, lane-cooperative)
//   mode 16: as 15 with global_load_lds_dwordx4 (LDS-DMA, no VGPR staging)
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

__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) {   // quad_perm [1,0,3,2]
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}
template <int Q>
__device__ __forceinline__ uint32_t dpp_bcast(uint32_t v) {  // quad_perm [Q,Q,Q,Q]
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, Q | (Q << 2) | (Q << 4) | (Q << 6), 0xF, 0xF, true);
}
template <int Q>
__device__ __forceinline__ uint64_t dpp_bcast64(uint64_t v) {
    return (uint64_t)dpp_bcast<Q>((uint32_t)v) | ((uint64_t)dpp_bcast<Q>((uint32_t)(v >> 32)) << 32);
}

// Cooperative shapes: several lanes fetch one row with ONE instruction (adjacent 16-byte pieces),
// so that the texture addresser sees half / a quarter as many distinct rows per instruction.
template <int MODE>
__global__ __launch_bounds__(256) void chase_coop(const uint4 *__restrict__ t, uint64_t rows, uint32_t steps, uint32_t *out) {
    const uint64_t lane = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t sub = threadIdx.x & (MODE == 11 || MODE == 12 ? 1u : 3u);
    // chain id: shared by the lanes of a group in modes 11 / 14, one per lane in 12 / 13
    const uint64_t chain = MODE == 11 ? lane >> 1 : (MODE == 14 ? lane >> 2 : lane);
    uint64_t j = (chain * 0xD1342543DE82EF95ull + 12345) % rows;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        if (MODE == 11) {
            j &= ~(uint64_t)1;
            const uint4 w = t[j + sub];
            acc += w.z + dpp_xor1(w.z);
            const uint64_t nx = (uint64_t)dpp_bcast<0>(w.x) | ((uint64_t)dpp_bcast<0>(w.y) << 32);   // pair lane 0 = quad lanes 0 / 2
            const uint64_t nx2 = (uint64_t)dpp_bcast<2>(w.x) | ((uint64_t)dpp_bcast<2>(w.y) << 32);
            j = (((threadIdx.x & 2u) ? nx2 : nx) + s) % rows;
        } else if (MODE == 12) {
            j &= ~(uint64_t)1;
            const uint64_t je = (threadIdx.x & 2u) ? dpp_bcast64<2>(j) : dpp_bcast64<0>(j);   // even lane's row
            const uint64_t jo = (threadIdx.x & 2u) ? dpp_bcast64<3>(j) : dpp_bcast64<1>(j);   // odd lane's row
            const uint4 a = t[je + sub];
            const uint4 b = t[jo + sub];
            // even lane: own row = (a, partner's a); odd lane: own row = (partner's b, b)
            const uint32_t pax = dpp_xor1(a.x), pay = dpp_xor1(a.y), paz = dpp_xor1(a.z);
            const uint32_t pbx = dpp_xor1(b.x), pby = dpp_xor1(b.y), pbz = dpp_xor1(b.z);
            const uint32_t fx = sub ? pbx : a.x, fy = sub ? pby : a.y, fz = sub ? pbz : a.z;   // first half
            const uint32_t sz = sub ? b.z : paz;                                               // second half
            acc += fz + sz + (pax ^ pay);
            j = (((uint64_t)fx | ((uint64_t)fy << 32)) + s) % rows;
        } else if (MODE == 13) {
            j &= ~(uint64_t)3;
            const uint4 r0 = t[dpp_bcast64<0>(j) + sub];
            const uint4 r1 = t[dpp_bcast64<1>(j) + sub];
            const uint4 r2 = t[dpp_bcast64<2>(j) + sub];
            const uint4 r3 = t[dpp_bcast64<3>(j) + sub];
            // lane p owns row p: its first piece is quad lane 0's r<p>
            const uint32_t x0 = dpp_bcast<0>(r0.x), y0 = dpp_bcast<0>(r0.y);
            const uint32_t x1 = dpp_bcast<0>(r1.x), y1 = dpp_bcast<0>(r1.y);
            const uint32_t x2 = dpp_bcast<0>(r2.x), y2 = dpp_bcast<0>(r2.y);
            const uint32_t x3 = dpp_bcast<0>(r3.x), y3 = dpp_bcast<0>(r3.y);
            const uint32_t fx = sub == 0 ? x0 : (sub == 1 ? x1 : (sub == 2 ? x2 : x3));
            const uint32_t fy = sub == 0 ? y0 : (sub == 1 ? y1 : (sub == 2 ? y2 : y3));
            acc += r0.z + r1.z + r2.z + r3.z;
            j = (((uint64_t)fx | ((uint64_t)fy << 32)) + s) % rows;
        } else {   // 14
            j &= ~(uint64_t)3;
            const uint4 w = t[j + sub];
            acc += w.z + dpp_xor1(w.z);
            j = (((uint64_t)dpp_bcast<0>(w.x) | ((uint64_t)dpp_bcast<0>(w.y) << 32)) + s) % rows;
        }
    }
    out[lane] = acc;
}

// 128-byte rows through LDS.  Instruction q of a trip fetches, for every 8-lane group g, the row
// wanted by the group's lane q: lane (g, p) loads piece p ^ q of it, so that row (g, q) lies at
// stage[q][8g + (k ^ q)] for piece k and the read-back of 16 lanes hits 16 distinct banks.
template <int MODE>
__global__ __launch_bounds__(256) void chase_line(const uint4 *__restrict__ t, uint64_t lines, uint32_t steps, uint32_t *out) {
    __shared__ uint4 stage[4][8][64];   // 8 KB per wave
    __shared__ uint32_t jx[4][64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g8 = lane & ~7u, p = lane & 7u;
    const uint64_t chain = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t j = (uint32_t)((chain * 0xD1342543DE82EF95ull + 12345) % lines);
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        jx[wave][lane] = j;
        __builtin_amdgcn_wave_barrier();
        const uint4 ja = *reinterpret_cast<const uint4 *>(&jx[wave][g8]);
        const uint4 jb = *reinterpret_cast<const uint4 *>(&jx[wave][g8 + 4]);
        const uint32_t jq[8] = {ja.x, ja.y, ja.z, ja.w, jb.x, jb.y, jb.z, jb.w};
        if (MODE == 15) {
            uint4 r[8];
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) r[q] = t[(uint64_t)jq[q] * 8 + (p ^ q)];
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) stage[wave][q][lane] = r[q];
        } else {
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q)
                __builtin_amdgcn_global_load_lds(t + (uint64_t)jq[q] * 8 + (p ^ q), &stage[wave][q][0], 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_wave_barrier();
        const uint4 w0 = stage[wave][p][g8 + (0 ^ p)];                    // piece 0 of the own row
        const uint4 w1 = stage[wave][p][g8 + (((w0.x >> 3) & 7u) ^ p)];   // a data-dependent piece (a slot)
        // fill() stores the 16-byte piece's own index in .z: every staged piece is checked
        acc += (w0.z != j * 8u) + (w1.z != j * 8u + ((w0.x >> 3) & 7u));
        j = (uint32_t)((((uint64_t)w0.x | ((uint64_t)w0.y << 32)) + s) % lines);
        __builtin_amdgcn_wave_barrier();
    }
    out[chain] = acc;
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
            case 11: chase_coop<11><<<blocks, 256>>>(t, rows, steps, out); break;
            case 12: chase_coop<12><<<blocks, 256>>>(t, rows, steps, out); break;
            case 13: chase_coop<13><<<blocks, 256>>>(t, rows, steps, out); break;
            case 14: chase_coop<14><<<blocks, 256>>>(t, rows, steps, out); break;
            case 15: chase_line<15><<<blocks, 256>>>(t, rows / 8, steps, out); break;
            case 16: chase_line<16><<<blocks, 256>>>(t, rows / 8, steps, out); break;
            default: run<8>(blocks, t, rows, steps, out); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double loads = (double)blocks * 256 * steps / (mode == 11 ? 2 : (mode == 14 ? 4 : 1));   // row fetches
        unsigned long long bad = 0;
        if (mode >= 15 && r == reps - 1) {   // modes 15 / 16 count the staged pieces that were not the wanted ones
            uint32_t *h = (uint32_t *)malloc((size_t)blocks * 256 * 4);
            CK(hipMemcpy(h, out, (size_t)blocks * 256 * 4, hipMemcpyDeviceToHost));
            for (uint64_t k = 0; k < (uint64_t)blocks * 256; ++k) bad += h[k];
            free(h);
        }
        printf("{\"table_MiB\": %llu, \"lanes\": %llu, \"steps\": %u, \"mode\": %d, \"alloc\": %d, \"ms\": %.3f, \"Gsteps_per_s\": %.3f, \"steps_total\": %.0f, \"bad_pieces\": %llu}\n",
               (unsigned long long)mib, (unsigned long long)lanes, steps, mode, alloc, ms, loads / ms / 1e6, loads, bad);
    }
    return 0;
}
