// fastx_reader.cpp -- see fastx_reader.h.  Behavioural mirror of
// PatternProcessor (io.hpp:6-35) over klib's published kseq_read semantics.
#include "fastx_reader.h"

#include <ctype.h>
#include <fcntl.h>
#include <string.h>
#include <unistd.h>

namespace colbwt {

FastxReader::~FastxReader() {
    if (fp_) gzclose(fp_);
    if (raw_fd_ >= 0) ::close(raw_fd_);
}

bool FastxReader::open(const std::string &path) {
    // io.hpp:9 gzopen reads plain files transparently; a regular file that does not start with
    // the gzip magic is read with read(2) directly -- the same bytes
    raw_fd_ = ::open(path.c_str(), O_RDONLY);
    if (raw_fd_ < 0) return false;
    unsigned char magic[2] = {0, 0};
    const ssize_t got = ::pread(raw_fd_, magic, 2, 0);
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        ::close(raw_fd_);
        raw_fd_ = -1;
        fp_ = gzopen(path.c_str(), "r");
        if (!fp_) return false;
        gzbuffer(fp_, 1 << 20);
    }
    buf_.resize(4 << 20);
    begin_ = end_ = 0;
    eof_ = false;
    pending_header_ = 0;
    return true;
}

bool FastxReader::open_at(const std::string &path, uint64_t offset) {
    if (fp_) { gzclose(fp_); fp_ = nullptr; }
    if (raw_fd_ >= 0) ::close(raw_fd_);
    raw_fd_ = ::open(path.c_str(), O_RDONLY);
    if (raw_fd_ < 0 || ::lseek(raw_fd_, (off_t)offset, SEEK_SET) < 0) return false;
    buf_.resize(4 << 20);
    begin_ = end_ = 0;
    eof_ = false;
    pending_header_ = 0;
    return true;
}

int FastxReader::getc_() {
    if (begin_ >= end_) {
        if (eof_) return -1;
        const int got = fp_ ? gzread(fp_, buf_.data(), (unsigned)buf_.size())
                            : (int)::read(raw_fd_, buf_.data(), buf_.size());
        if (got <= 0) {
            eof_ = true;
            return -1;
        }
        begin_ = 0;
        end_ = (size_t)got;
    }
    return buf_[begin_++];
}

bool FastxReader::rest_of_line_(std::vector<uint8_t> *dst, size_t base_len) {
    bool any = false;
    for (;;) {
        if (begin_ >= end_) {
            const int c = getc_();
            if (c < 0) break;
            --begin_;  // put it back; the scan below consumes it
        }
        any = true;
        const void *nl = memchr(buf_.data() + begin_, '\n', end_ - begin_);
        const size_t i = nl ? (size_t)((const uint8_t *)nl - buf_.data()) : end_;
        if (dst) dst->insert(dst->end(), buf_.begin() + begin_, buf_.begin() + i);
        const bool hit = i < end_;
        begin_ = hit ? i + 1 : i;
        if (hit) break;
    }
    if (!any) return false;
    // kseq strips one trailing '\r' when more than one byte has accumulated
    if (dst && dst->size() - base_len > 1 && dst->back() == '\r') dst->pop_back();
    return true;
}

bool FastxReader::next(std::string &name, std::vector<uint8_t> &bases) {
    if (!fp_ && raw_fd_ < 0) return false;
    int c;
    if (!pending_header_) {
        while ((c = getc_()) >= 0 && c != '>' && c != '@') {}
        if (c < 0) return false;
    }
    pending_header_ = 0;

    name.clear();
    bool any = false;
    while ((c = getc_()) >= 0) {
        any = true;
        if (isspace(c)) break;
        name.push_back((char)c);
    }
    if (!any) return false;
    if (c >= 0 && c != '\n') rest_of_line_(nullptr, 0);  // comment

    const size_t base_len = bases.size();
    while ((c = getc_()) >= 0 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;  // empty line
        bases.push_back((uint8_t)c);
        rest_of_line_(&bases, base_len);
    }
    if (c == '>' || c == '@') pending_header_ = c;
    if (c != '+') return true;  // FASTA record

    // FASTQ: skip the '+' line, then consume qualities until they cover the sequence
    while ((c = getc_()) >= 0 && c != '\n') {}
    const size_t seq_len = bases.size() - base_len;
    bool ok = c >= 0;
    if (ok) {
        qual_.clear();
        while (rest_of_line_(&qual_, 0) && qual_.size() < seq_len) {}
        ok = qual_.size() == seq_len;
    }
    if (!ok) {  // kseq_read returns -2: PatternProcessor::read() is false, iteration ends
        bases.resize(base_len);
        eof_ = true;
        begin_ = end_ = 0;
        return false;
    }
    return true;
}

}  // namespace colbwt
