// read_sampler.h -- synthetic reads by backward walk: read[m-1-k] = the
// character at LF^k(p0), p0 uniform in [0, n); 0x01 -> 'A'; substitutions at sub_permille / 1000.
// Generator only: its results are INPUTS of the query, never checked outputs.  A read is a
// function of (seed, read number) and of BWT positions alone, so every HBM layout of one index
// yields the same reads; `View` adapts a layout:
//   n(), rows(), idx(j), load(j) -> Row, ch(Row), lf_row(Row), lf_off(Row), len(j, Row)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace colbwt {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

template <class View>
__device__ __forceinline__ void sample_read(const View &V, uint64_t rd, uint32_t m, uint32_t sub_permille, uint64_t seed,
                                            uint8_t *__restrict__ out) {
    uint64_t st = splitmix64(seed ^ (rd * 0xD1342543DE82EF95ull));
    const uint64_t p0 = st % V.n();
    uint64_t lo = 0, hi = V.rows();   // idx[lo] <= p0 < idx[hi] (sentinel idx[rows] = n)
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (V.idx((uint32_t)mid) <= p0) lo = mid; else hi = mid;
    }
    uint32_t j = (uint32_t)lo;
    auto w = V.load(j);
    uint64_t o = p0 - V.idx(j);
    const char acgt[4] = {'A', 'C', 'G', 'T'};
    for (uint32_t k = 0; k < m; ++k) {
        uint32_t ch = V.ch(w);
        if (ch <= 1) ch = 'A';
        st = splitmix64(st);
        if ((uint32_t)(st % 1000) < sub_permille) {
            const uint32_t cur = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
            const uint32_t pick = (uint32_t)((st >> 32) % 3);
            ch = cur < 4 ? acgt[(cur + 1 + pick) & 3] : acgt[(st >> 40) & 3];
        }
        out[m - 1 - k] = (uint8_t)ch;
        uint64_t t = (uint64_t)V.lf_off(w) + o;   // LF_table::LF (LF_table.hpp:251-262)
        j = V.lf_row(w);
        w = V.load(j);
        for (;;) {
            const uint64_t len = V.len(j, w);
            if (t < len || j >= V.rows() - 1) break;
            t -= len;
            ++j;
            w = V.load(j);
        }
        o = t;
    }
}

}  // namespace colbwt
