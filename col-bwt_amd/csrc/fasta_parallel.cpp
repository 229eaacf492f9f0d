// fasta_parallel.cpp -- see fasta_parallel.h.
#include "fasta_parallel.h"

#include <ctype.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <thread>

namespace colbwt {

ParallelFasta::~ParallelFasta() {
    if (data_ && size_) munmap((void *)data_, size_);
    if (fd_ >= 0) ::close(fd_);
}

bool ParallelFasta::open(const std::string &path) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) return false;
    struct stat st;
    if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2) return false;
    size_ = (uint64_t)st.st_size;
    void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) {
        size_ = 0;
        return false;
    }
    data_ = (const uint8_t *)m;
    (void)madvise(m, size_, MADV_SEQUENTIAL);
    cur_ = 0;
    return data_[0] == '>';   // (a gzip file starts with 0x1f)
}

uint64_t ParallelFasta::next_record_start(uint64_t from) const {
    uint64_t p = from;
    while (p < size_) {
        const void *hit = memchr(data_ + p, '>', size_ - p);
        if (!hit) return size_;
        p = (uint64_t)((const uint8_t *)hit - data_);
        if (p == 0 || data_[p - 1] == '\n') return p;
        ++p;
    }
    return size_;
}

namespace {
struct Piece {
    std::vector<std::string> names;
    std::vector<uint8_t> bases;
    std::vector<uint64_t> ends;      // end of every record in `bases`
    uint64_t max_len = 0;
    bool plain = true;
};

// Records of [lo, hi): lo is a line-start '>', hi a record start or the end of the file.
void parse_piece(const uint8_t *d, uint64_t lo, uint64_t hi, Piece &out) {
    out.bases.reserve((size_t)(hi - lo));
    uint64_t p = lo;
    while (p < hi) {
        // header: '>' name [whitespace comment] '\n'   (io.hpp:24-26: the name ends at the first whitespace)
        ++p;
        uint64_t q = p;
        while (q < hi && !isspace(d[q])) ++q;
        if (q == p && q >= hi) break;                        // '>' at the very end: kseq finds no name, no record
        out.names.emplace_back((const char *)d + p, (size_t)(q - p));
        if (q < hi && d[q] != '\n') {                        // comment: skip the rest of the line
            const void *nl = memchr(d + q, '\n', hi - q);
            q = nl ? (uint64_t)((const uint8_t *)nl - d) : hi;
        }
        p = q < hi ? q + 1 : hi;
        // sequence lines until a line that starts with '>' (or '+' / '@': not plain FASTA)
        const size_t base_len = out.bases.size();
        while (p < hi) {
            const uint8_t c = d[p];
            if (c == '>') break;
            if (c == '+' || c == '@') {
                out.plain = false;
                return;
            }
            if (c == '\n') {                                 // empty line
                ++p;
                continue;
            }
            const void *nl = memchr(d + p, '\n', hi - p);
            const uint64_t e = nl ? (uint64_t)((const uint8_t *)nl - d) : hi;
            out.bases.insert(out.bases.end(), d + p, d + e);
            // kseq strips one trailing '\r' once the record holds more than one byte
            if (out.bases.size() - base_len > 1 && out.bases.back() == '\r') out.bases.pop_back();
            p = e < hi ? e + 1 : hi;
        }
        out.ends.push_back(out.bases.size());
        out.max_len = std::max<uint64_t>(out.max_len, out.bases.size() - base_len);
    }
}
}  // namespace

ParallelFasta::Result ParallelFasta::next_batch(uint64_t target_bases, unsigned threads, std::vector<std::string> &names,
                                                std::vector<uint8_t> &bases, std::vector<uint64_t> &off, uint64_t &max_len) {
    if (cur_ >= size_) return kEnd;
    const uint64_t want = cur_ + target_bases + target_bases / 16 + 4096;
    const uint64_t end = want >= size_ ? size_ : next_record_start(want);
    threads = std::max(1u, std::min(threads, 64u));
    std::vector<uint64_t> cut(threads + 1, end);
    cut[0] = cur_;
    for (unsigned t = 1; t < threads; ++t) {
        const uint64_t at = cur_ + (end - cur_) / threads * t;
        cut[t] = std::max(cut[t - 1], std::min(end, next_record_start(at)));
    }
    std::vector<Piece> pieces(threads);
    {
        std::vector<std::thread> ts;
        auto work = [&](unsigned t) {
            if (cut[t + 1] > cut[t]) parse_piece(data_, cut[t], cut[t + 1], pieces[t]);
        };
        for (unsigned t = 1; t < threads; ++t) ts.emplace_back(work, t);
        work(0);
        for (auto &th : ts) th.join();
    }
    for (const Piece &pc : pieces)
        if (!pc.plain) return kNotPlainFasta;                // nothing appended; cur_ is the batch's first record
    uint64_t nb = 0, nr = 0;
    for (const Piece &pc : pieces) {
        nb += pc.bases.size();
        nr += pc.names.size();
    }
    const uint64_t b0 = bases.size();
    bases.resize(b0 + nb);
    names.reserve(names.size() + nr);
    off.reserve(off.size() + nr);
    std::vector<uint64_t> at(threads + 1, b0);
    for (unsigned t = 0; t < threads; ++t) at[t + 1] = at[t] + pieces[t].bases.size();
    {
        std::vector<std::thread> ts;
        auto copy = [&](unsigned t) {
            if (!pieces[t].bases.empty()) memcpy(bases.data() + at[t], pieces[t].bases.data(), pieces[t].bases.size());
        };
        for (unsigned t = 1; t < threads; ++t) ts.emplace_back(copy, t);
        copy(0);
        for (auto &th : ts) th.join();
    }
    for (unsigned t = 0; t < threads; ++t) {
        Piece &pc = pieces[t];
        for (auto &nm : pc.names) names.push_back(std::move(nm));
        for (uint64_t e : pc.ends) off.push_back(at[t] + e);
        max_len = std::max(max_len, pc.max_len);
    }
    cur_ = end;
    return kBatch;
}

}  // namespace colbwt
