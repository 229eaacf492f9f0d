// text_writer.h -- the reference's text output (pml_to_vec, pml_query.cpp:65-90):
// per read  '>' name ' ' '\n'  then every value followed by one space, then
// '\n'; k-th value <-> pattern[k].  Byte-identical to
//   fs << '>' << id << " \n"; std::copy(v.begin(), v.end(), std::ostream_iterator<size_t>(fs, " ")); fs << "\n";
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

namespace colbwt {

class TextWriter {
public:
    TextWriter() = default;
    ~TextWriter() { close(); }
    TextWriter(const TextWriter &) = delete;
    TextWriter &operator=(const TextWriter &) = delete;

    bool open(const std::string &path);
    template <typename T>
    bool record(const std::string &name, const T *vals, uint64_t m);
    // The same bytes as calling record() for reads 0..n_reads-1 in order (read k =
    // vals[off[k] .. off[k+1])), formatted by `threads` host threads.  The decimal
    // formatting of ~3-4 bytes per base dominates the reference's query time on
    // match-heavy reads.
    template <typename T>
    bool batch(const std::vector<std::string> &names, const uint64_t *off, const T *vals, uint64_t n_reads,
               unsigned threads);
    bool close();

private:
    bool flush_();
    bool write_all_(const char *p, size_t n);
    int fd_ = -1;
    uint64_t pos_ = 0;                 // file offset of the next byte
    std::vector<char> buf_;
    size_t used_ = 0;
    bool ok_ = true;
    // formatting buffers of batch(), one per worker, kept across calls (fresh pages cost as
    // much as the formatting itself)
    struct Chunk {
        char *p = nullptr;
        size_t cap = 0, used = 0;
    };
    std::vector<Chunk> chunks_;
};

}  // namespace colbwt
