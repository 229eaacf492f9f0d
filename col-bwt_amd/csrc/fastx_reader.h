// fastx_reader.h -- FASTA/FASTQ (optionally gzip) record reader with the
// observable behaviour of the reference's PatternProcessor (io.hpp:6-35),
// i.e. of klib's kseq_read on a gzFile:
//   * name  = header up to the first whitespace, comment dropped (io.hpp:24-26);
//   * bases = sequence lines concatenated verbatim: no case folding, no N
//     handling (col_bwt.hpp:512 consumes raw bytes);
//   * a record ends at a line starting with '>', '@' or '+'; FASTQ qualities
//     are skipped; reading stops at the first malformed FASTQ record
//     (PatternProcessor::read returns false on any negative kseq_read, io.hpp:13-15).
// klib is an un-vendored dependency of the reference (thirdparty/CMakeLists.txt:22-32):
// reader edge cases are "parity unpinned".
#pragma once
#include <stdint.h>
#include <zlib.h>

#include <string>
#include <vector>

namespace colbwt {

class FastxReader {
public:
    FastxReader() = default;
    ~FastxReader();
    FastxReader(const FastxReader &) = delete;
    FastxReader &operator=(const FastxReader &) = delete;

    bool open(const std::string &path);
    // A plain (non-gzip) file from byte `offset` on, which must be the '>' or '@' of a record.
    bool open_at(const std::string &path, uint64_t offset);
    // Appends the next record's bases to `bases` and stores its name; false at
    // end of input (or at the first malformed FASTQ record).
    bool next(std::string &name, std::vector<uint8_t> &bases);

private:
    int getc_();
    // Appends the rest of the current line (without '\n') to `dst`; returns
    // false when nothing at all (not even a newline) could be read.
    bool rest_of_line_(std::vector<uint8_t> *dst, size_t base_len);

    gzFile fp_ = nullptr;
    int raw_fd_ = -1;   // a file without the gzip magic is read directly (what zlib's transparent
                        // mode would hand back, without its extra copy)
    std::vector<uint8_t> buf_;
    size_t begin_ = 0, end_ = 0;
    bool eof_ = false;
    int pending_header_ = 0;  // '>' or '@' already consumed, 0 otherwise
    std::vector<uint8_t> qual_;
};

}  // namespace colbwt
