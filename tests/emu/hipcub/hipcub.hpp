// tests/emu/hipcub/hipcub.hpp -- the few hipCUB device-wide primitives the product calls, restated
// with <algorithm> for the CPU emulator build (TEST INFRASTRUCTURE ONLY, see ../hip/hip_runtime.h).
// Same calling convention: a null workspace pointer only reports the size.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>

#include "../hip/hip_runtime.h"

namespace hipcub {

template <typename T>
struct DoubleBuffer {
    T *d_buffers[2];
    int selector = 0;
    DoubleBuffer(T *a, T *b) : d_buffers{a, b} {}
    T *Current() { return d_buffers[selector]; }
    T *Alternate() { return d_buffers[selector ^ 1]; }
};

struct Max {
    template <typename T>
    T operator()(const T &a, const T &b) const { return a < b ? b : a; }
};

template <typename T>
struct CountingInputIterator {
    T base;
    explicit CountingInputIterator(T b) : base(b) {}
    T operator[](size_t i) const { return base + (T)i; }
};

struct DeviceRadixSort {
    template <typename K, typename V>
    static void sort_(const K *kin, K *kout, const V *vin, V *vout, size_t n, int begin_bit, int end_bit) {
        std::vector<size_t> order(n);
        std::iota(order.begin(), order.end(), (size_t)0);
        const int width = end_bit - begin_bit;
        auto field = [&](K k) -> unsigned long long {
            const unsigned long long v = (unsigned long long)k >> begin_bit;
            return width >= 64 ? v : v & ((1ull << width) - 1);
        };
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return field(kin[a]) < field(kin[b]); });
        for (size_t i = 0; i < n; ++i) { kout[i] = kin[order[i]]; vout[i] = vin[order[i]]; }
    }
    template <typename K, typename V>
    static hipError_t SortPairs(void *ws, size_t &bytes, DoubleBuffer<K> &keys, DoubleBuffer<V> &vals, size_t n, int begin_bit = 0,
                                int end_bit = sizeof(K) * 8, hipStream_t = nullptr) {
        if (!ws) { bytes = 64; return hipSuccess; }
        sort_(keys.Current(), keys.Alternate(), vals.Current(), vals.Alternate(), n, begin_bit, end_bit);
        keys.selector ^= 1;
        vals.selector ^= 1;
        return hipSuccess;
    }
    template <typename K, typename V>
    static hipError_t SortPairs(void *ws, size_t &bytes, const K *kin, K *kout, const V *vin, V *vout, size_t n, int begin_bit = 0,
                                int end_bit = sizeof(K) * 8, hipStream_t = nullptr) {
        if (!ws) { bytes = 64; return hipSuccess; }
        sort_(kin, kout, vin, vout, n, begin_bit, end_bit);
        return hipSuccess;
    }
};

struct DeviceScan {
    template <typename In, typename Out, typename Op>
    static hipError_t InclusiveScan(void *ws, size_t &bytes, In in, Out out, Op op, size_t n, hipStream_t = nullptr) {
        if (!ws) { bytes = 64; return hipSuccess; }
        for (size_t i = 0; i < n; ++i) out[i] = i ? op(out[i - 1], in[i]) : in[i];
        return hipSuccess;
    }
};

struct DeviceSelect {
    template <typename In, typename Flag, typename Out, typename Count>
    static hipError_t Flagged(void *ws, size_t &bytes, In in, Flag flags, Out out, Count *n_selected, size_t n, hipStream_t = nullptr) {
        if (!ws) { bytes = 64; return hipSuccess; }
        size_t m = 0;
        for (size_t i = 0; i < n; ++i)
            if (flags[i]) out[m++] = in[i];
        *n_selected = (Count)m;
        return hipSuccess;
    }
};

}  // namespace hipcub
