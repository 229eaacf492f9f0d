// text_writer.cpp -- see text_writer.h (pml_query.cpp:78-85).
#include "text_writer.h"

#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/types.h>
#include <unistd.h>

#include <thread>

namespace colbwt {

namespace {
constexpr size_t kBufBytes = 4u << 20;

// "<decimal digits> " of every value below 10000, 8 bytes each: characters in bytes 0..4,
// length (digits + 1) in byte 7.  PML / col-id values are almost always this small.
struct DigitTable {
    uint64_t e[10000];
    DigitTable() {
        for (uint32_t v = 0; v < 10000; ++v) {
            char t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int nd = 0;
            char tmp[4];
            uint32_t x = v;
            do {
                tmp[nd++] = (char)('0' + x % 10);
                x /= 10;
            } while (x);
            for (int d = 0; d < nd; ++d) t[d] = tmp[nd - 1 - d];
            t[nd] = ' ';
            t[7] = (char)(nd + 1);
            memcpy(&e[v], t, 8);
        }
    }
};
const DigitTable g_digits;

// decimal digits of v followed by one space; returns bytes written (<= 11).  May store up to
// 8 bytes at dst: callers keep that much slack.
inline size_t put_value(char *dst, uint32_t v) {
    if (v < 10000) {
        const uint64_t e = g_digits.e[v];
        memcpy(dst, &e, 8);
        return (size_t)(e >> 56);
    }
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
    fd_ = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);  // pml_query.cpp:67,70: std::ofstream, truncating
    buf_.resize(kBufBytes + 64);
    used_ = 0;
    pos_ = 0;
    ok_ = fd_ >= 0;
    return ok_;
}

bool TextWriter::write_all_(const char *p, size_t n) {
    while (n > 0) {
        ssize_t w = ::pwrite(fd_, p, n, (off_t)pos_);
        if (w < 0 && errno == ESPIPE) w = ::write(fd_, p, n);   // a pipe or terminal (`col-bwt view`): sequential anyway
        if (w <= 0) return false;
        p += w;
        n -= (size_t)w;
        pos_ += (uint64_t)w;
    }
    return true;
}

bool TextWriter::flush_() {
    if (used_ && fd_ >= 0) ok_ = write_all_(buf_.data(), used_) && ok_;
    used_ = 0;
    return ok_;
}

template <typename T>
bool TextWriter::record(const std::string &name, const T *vals, uint64_t m) {
    if (fd_ < 0) return false;
    if (used_ + name.size() + 3 > kBufBytes) flush_();
    if (name.size() + 3 > kBufBytes) {  // absurdly long name: write through
        flush_();
        ok_ = write_all_(">", 1) && write_all_(name.data(), name.size()) && write_all_(" \n", 2) && ok_;
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
    if (fd_ < 0) return false;
    if (n_reads == 0) return ok_;
    const uint64_t total = off[n_reads] - off[0];
    if (threads < 1) threads = 1;
    if (threads > n_reads) threads = (unsigned)n_reads;
    if (total < (1u << 20)) threads = 1;
    if (chunks_.size() < threads) chunks_.resize(threads);
    std::vector<uint64_t> cut(threads + 1, n_reads);
    cut[0] = 0;
    for (unsigned t = 1; t < threads; ++t) {  // contiguous shards balanced by base count
        const uint64_t target = off[0] + total * t / threads;
        uint64_t lo = cut[t - 1];
        while (lo < n_reads && off[lo] < target) ++lo;
        cut[t] = lo;
    }
    std::vector<char> bad(threads, 0);
    auto format = [&](unsigned t) {
        Chunk &c = chunks_[t];
        // worst case: 11 bytes per value ("4294967295 "), name + 4 per read
        uint64_t need = 64;
        for (uint64_t k = cut[t]; k < cut[t + 1]; ++k) need += names[k].size() + 4;
        need += (off[cut[t + 1]] - off[cut[t]]) * (sizeof(T) == 1 ? 4 : sizeof(T) == 2 ? 6 : 11);
        if (c.cap < need) {
            free(c.p);
            c.cap = need + need / 8;
            c.p = (char *)malloc(c.cap);
            if (!c.p) { c.cap = 0; bad[t] = 1; return; }
        }
        char *b = c.p;
        size_t u = 0;
        for (uint64_t k = cut[t]; k < cut[t + 1]; ++k) {
            b[u++] = '>';
            memcpy(b + u, names[k].data(), names[k].size());
            u += names[k].size();
            b[u++] = ' ';
            b[u++] = '\n';
            for (uint64_t e = off[k]; e < off[k + 1]; ++e) u += put_value(b + u, (uint32_t)vals[e]);
            b[u++] = '\n';
        }
        c.used = u;
    };
    // every worker formats its shard, then (once all sizes are known) writes it at its own
    // file offset: the copies into the page cache run side by side too
    std::vector<uint64_t> at(threads + 1, 0);
    auto write = [&](unsigned t) {
        const Chunk &c = chunks_[t];
        const char *p = c.p;
        size_t n = c.used;
        uint64_t o = at[t];
        while (n > 0) {
            const ssize_t w = ::pwrite(fd_, p, n, (off_t)o);
            if (w <= 0) { bad[t] = 1; return; }
            p += w;
            n -= (size_t)w;
            o += (uint64_t)w;
        }
    };
    flush_();
    auto run = [&](auto &&fn) {
        if (threads == 1) {
            fn(0u);
            return;
        }
        std::vector<std::thread> ts;
        for (unsigned t = 1; t < threads; ++t) ts.emplace_back(fn, t);
        fn(0u);
        for (auto &th : ts) th.join();
    };
    run(format);
    at[0] = pos_;
    for (unsigned t = 0; t < threads; ++t) at[t + 1] = at[t] + (bad[t] ? 0 : chunks_[t].used);
    for (unsigned t = 0; t < threads; ++t) ok_ = ok_ && !bad[t];
    if (!ok_) return false;
    run(write);
    for (unsigned t = 0; t < threads; ++t) ok_ = ok_ && !bad[t];
    pos_ = at[threads];
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
    for (Chunk &c : chunks_) free(c.p);
    chunks_.clear();
    if (fd_ < 0) return ok_;
    flush_();
    ok_ = (::close(fd_) == 0) && ok_;
    fd_ = -1;
    return ok_;
}

}  // namespace colbwt
