// bin_writer.cpp -- see bin_writer.h ("Movi-like, unverified" container).
#include "bin_writer.h"

#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <thread>

#include "text_writer.h"

namespace colbwt {

bool BinWriter::open(const std::string &path) {
    close();
    fd_ = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    pos_ = 0;
    ok_ = fd_ >= 0;
    return ok_;
}

bool BinWriter::close() {
    if (fd_ >= 0) {
        if (::close(fd_) != 0) ok_ = false;
        fd_ = -1;
    }
    return ok_;
}

namespace {
bool pwrite_all(int fd, const uint8_t *p, size_t n, uint64_t at) {
    while (n > 0) {
        const ssize_t w = ::pwrite(fd, p, n, (off_t)at);
        if (w <= 0) return false;
        p += w;
        n -= (size_t)w;
        at += (uint64_t)w;
    }
    return true;
}
}  // namespace

template <typename OutT, typename T>
bool BinWriter::batch(const std::vector<std::string> &names, const uint64_t *off, const T *vals, uint64_t n_reads,
                      unsigned threads) {
    if (!ok_ || fd_ < 0) return false;
    if (n_reads == 0) return true;
    rec_off_.resize(n_reads + 1);
    uint64_t at = 0;
    for (uint64_t k = 0; k < n_reads; ++k) {
        rec_off_[k] = at;
        const uint64_t nl = std::min<uint64_t>(names[k].size(), 0xFFFF);
        at += 2 + nl + 8 + (off[k + 1] - off[k]) * sizeof(OutT);
    }
    rec_off_[n_reads] = at;
    if (buf_.size() < at) buf_.resize(at + at / 8);
    threads = std::max(1u, std::min<unsigned>(threads, (unsigned)std::min<uint64_t>(n_reads, 64)));
    std::vector<char> good(threads, 1);
    auto work = [&](unsigned t) {
        // reads [lo, hi): equal shares of the batch's bytes
        const uint64_t b_lo = at / threads * t, b_hi = t + 1 == threads ? at : at / threads * (t + 1);
        const uint64_t lo = (uint64_t)(std::lower_bound(rec_off_.begin(), rec_off_.begin() + n_reads, b_lo) - rec_off_.begin());
        const uint64_t hi = t + 1 == threads
                                ? n_reads
                                : (uint64_t)(std::lower_bound(rec_off_.begin(), rec_off_.begin() + n_reads, b_hi) - rec_off_.begin());
        if (hi <= lo) return;
        for (uint64_t k = lo; k < hi; ++k) {
            uint8_t *p = buf_.data() + rec_off_[k];
            const uint16_t nl = (uint16_t)std::min<uint64_t>(names[k].size(), 0xFFFF);
            memcpy(p, &nl, 2);
            memcpy(p + 2, names[k].data(), nl);
            const uint64_t m = off[k + 1] - off[k];
            memcpy(p + 2 + nl, &m, 8);
            uint8_t *q = p + 2 + nl + 8;                      // unaligned: values go through memcpy
            const T *v = vals + off[k];
            constexpr uint64_t kMax = sizeof(OutT) == 1 ? 0xFFull : 0xFFFFull;
            for (uint64_t e = 0; e < m; ++e) {                // computation order: last base first
                const uint64_t x = (uint64_t)v[m - 1 - e];
                const OutT o = (OutT)(x > kMax ? kMax : x);
                memcpy(q + e * sizeof(OutT), &o, sizeof(OutT));
            }
        }
        if (!pwrite_all(fd_, buf_.data() + rec_off_[lo], rec_off_[hi] - rec_off_[lo], pos_ + rec_off_[lo])) good[t] = 0;
    };
    std::vector<std::thread> ts;
    for (unsigned t = 1; t < threads; ++t) ts.emplace_back(work, t);
    work(0);
    for (auto &th : ts) th.join();
    for (char g : good)
        if (!g) ok_ = false;
    pos_ += at;
    return ok_;
}

template bool BinWriter::batch<uint16_t, uint16_t>(const std::vector<std::string> &, const uint64_t *, const uint16_t *, uint64_t, unsigned);
template bool BinWriter::batch<uint16_t, uint32_t>(const std::vector<std::string> &, const uint64_t *, const uint32_t *, uint64_t, unsigned);
template bool BinWriter::batch<uint8_t, uint8_t>(const std::vector<std::string> &, const uint64_t *, const uint8_t *, uint64_t, unsigned);

bool binary_to_text(const std::string &bin_path, int value_bytes, const std::string &text_path, std::string &err) {
    if (value_bytes != 1 && value_bytes != 2) {
        err = "value_bytes must be 1 (.cid.bin) or 2 (.pml.bin)";
        return false;
    }
    const int fd = ::open(bin_path.c_str(), O_RDONLY);
    if (fd < 0) {
        err = "cannot open " + bin_path;
        return false;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        ::close(fd);
        err = "cannot stat " + bin_path;
        return false;
    }
    const uint64_t len = (uint64_t)st.st_size;
    const uint8_t *data = nullptr;
    if (len) {
        void *m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            ::close(fd);
            err = "cannot map " + bin_path;
            return false;
        }
        data = (const uint8_t *)m;
    }
    TextWriter out;
    bool ok = out.open(text_path);
    if (!ok) err = "cannot create " + text_path;
    uint64_t at = 0;
    std::vector<uint32_t> vals;
    while (ok && at < len) {
        uint16_t nl;
        uint64_t m;
        if (at + 2 > len) { ok = false; break; }
        memcpy(&nl, data + at, 2);
        if (at + 2 + nl + 8 > len) { ok = false; break; }
        const std::string name((const char *)data + at + 2, nl);
        memcpy(&m, data + at + 2 + nl, 8);
        at += 2 + (uint64_t)nl + 8;
        if (m > (len - at) / (uint64_t)value_bytes) { ok = false; break; }
        vals.resize(m);
        for (uint64_t e = 0; e < m; ++e) {                    // back to pattern order
            uint32_t x = 0;
            memcpy(&x, data + at + e * value_bytes, value_bytes);
            vals[m - 1 - e] = x;
        }
        at += m * value_bytes;
        if (!out.record(name, vals.data(), m)) { ok = false; err = "short write on " + text_path; }
    }
    if (!ok && err.empty()) err = "malformed container " + bin_path + " near byte " + std::to_string(at);
    if (data) munmap((void *)data, len);
    ::close(fd);
    if (!out.close() && ok) {
        ok = false;
        err = "short write on " + text_path;
    }
    return ok;
}

}  // namespace colbwt
