// text_writer.cpp -- see text_writer.h (pml_query.cpp:78-85).
#include "text_writer.h"

#include <string.h>

#include <thread>

namespace colbwt {

namespace {
constexpr size_t kBufBytes = 4u << 20;

// decimal digits of v followed by one space; returns bytes written (<= 11)
inline size_t put_value(char *dst, uint32_t v) {
    char tmp[10];
    int nd = 0;
    do {
        tmp[nd++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    for (int d = 0; d < nd; ++d) dst[d] = tmp[nd - 1 - d];
    dst[nd] = ' ';
    return (size_t)nd + 1;
}
}  // namespace

bool TextWriter::open(const std::string &path) {
    f_ = fopen(path.c_str(), "wb");  // pml_query.cpp:67,70: std::ofstream, truncating
    buf_.resize(kBufBytes + 64);
    used_ = 0;
    ok_ = f_ != nullptr;
    return ok_;
}

bool TextWriter::flush_() {
    if (used_ && f_) ok_ = ok_ && fwrite(buf_.data(), 1, used_, f_) == used_;
    used_ = 0;
    return ok_;
}

template <typename T>
bool TextWriter::record(const std::string &name, const T *vals, uint64_t m) {
    if (!f_) return false;
    if (used_ + name.size() + 3 > kBufBytes) flush_();
    if (name.size() + 3 > kBufBytes) {  // absurdly long name: write through
        flush_();
        ok_ = ok_ && fputc('>', f_) != EOF && fwrite(name.data(), 1, name.size(), f_) == name.size() &&
              fwrite(" \n", 1, 2, f_) == 2;
    } else {
        buf_[used_++] = '>';
        memcpy(&buf_[used_], name.data(), name.size());
        used_ += name.size();
        buf_[used_++] = ' ';
        buf_[used_++] = '\n';
    }
    for (uint64_t k = 0; k < m; ++k) {
        if (used_ + 12 > kBufBytes) flush_();
        used_ += put_value(&buf_[used_], (uint32_t)vals[k]);
    }
    if (used_ + 1 > kBufBytes) flush_();
    buf_[used_++] = '\n';
    return ok_;
}

template <typename T>
bool TextWriter::batch(const std::vector<std::string> &names, const uint64_t *off, const T *vals, uint64_t n_reads,
                       unsigned threads) {
    if (!f_) return false;
    if (n_reads == 0) return ok_;
    const uint64_t total = off[n_reads] - off[0];
    if (threads < 1) threads = 1;
    if (threads > n_reads) threads = (unsigned)n_reads;
    if (total < (1u << 20)) threads = 1;
    std::vector<std::vector<char>> out(threads);
    std::vector<uint64_t> cut(threads + 1, n_reads);
    cut[0] = 0;
    for (unsigned t = 1, k = 0; t < threads; ++t) {  // contiguous shards balanced by base count
        const uint64_t target = off[0] + total * t / threads;
        uint64_t lo = cut[t - 1];
        (void)k;
        while (lo < n_reads && off[lo] < target) ++lo;
        cut[t] = lo;
    }
    auto work = [&](unsigned t) {
        std::vector<char> &b = out[t];
        uint64_t need = 0;
        for (uint64_t k = cut[t]; k < cut[t + 1]; ++k) need += names[k].size() + 4;
        need += (off[cut[t + 1]] - off[cut[t]]) * (sizeof(T) == 1 ? 4 : 6) + 64;
        b.resize(need);
        size_t u = 0;
        for (uint64_t k = cut[t]; k < cut[t + 1]; ++k) {
            b[u++] = '>';
            memcpy(&b[u], names[k].data(), names[k].size());
            u += names[k].size();
            b[u++] = ' ';
            b[u++] = '\n';
            if (u + (off[k + 1] - off[k]) * 11 + 2 > b.size()) b.resize(u + (off[k + 1] - off[k]) * 11 + 64 + b.size() / 2);
            for (uint64_t e = off[k]; e < off[k + 1]; ++e) u += put_value(&b[u], (uint32_t)vals[e]);
            b[u++] = '\n';
        }
        b.resize(u);
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> ts;
        for (unsigned t = 0; t < threads; ++t) ts.emplace_back(work, t);
        for (auto &th : ts) th.join();
    }
    flush_();
    for (unsigned t = 0; t < threads; ++t)
        if (!out[t].empty()) ok_ = ok_ && fwrite(out[t].data(), 1, out[t].size(), f_) == out[t].size();
    return ok_;
}

template bool TextWriter::batch<uint8_t>(const std::vector<std::string> &, const uint64_t *, const uint8_t *, uint64_t,
                                         unsigned);
template bool TextWriter::batch<uint16_t>(const std::vector<std::string> &, const uint64_t *, const uint16_t *, uint64_t,
                                          unsigned);
template bool TextWriter::batch<uint32_t>(const std::vector<std::string> &, const uint64_t *, const uint32_t *, uint64_t,
                                          unsigned);

template bool TextWriter::record<uint8_t>(const std::string &, const uint8_t *, uint64_t);
template bool TextWriter::record<uint16_t>(const std::string &, const uint16_t *, uint64_t);
template bool TextWriter::record<uint32_t>(const std::string &, const uint32_t *, uint64_t);

bool TextWriter::close() {
    if (!f_) return ok_;
    flush_();
    ok_ = (fclose(f_) == 0) && ok_;
    f_ = nullptr;
    return ok_;
}

}  // namespace colbwt
