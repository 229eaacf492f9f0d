// bin_writer.h -- binary result container of `col-bwt query` (scripts/col-bwt.py:194-198 names
// its outputs PATTERN.split.pml.bin / PATTERN.split.cid.bin; they are written by the un-vendored
// Movi fork, whose source is not part of the reference tree).  The record shape follows what
// SURVEY.md 8(c) recalls of upstream Movi -- "Movi-like, UNVERIFIED", parity unpinned:
//
//   per read, in file order:   u16 name_len | name bytes | u64 count | count values
//   values are stored in computation order (right to left: the value of the read's LAST base
//   first); .pml.bin holds u16 lengths (saturated at 65535: reads beyond that length keep exact
//   values only in the text files), .cid.bin holds u8 col ids.  Little endian, no file header.
//
// The text files (text_writer.h) stay the bit-exact contract with the reference; this container
// exists because formatting ~4 text bytes per base is what bounds the drop-in end to end.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace colbwt {

class BinWriter {
public:
    BinWriter() = default;
    ~BinWriter() { close(); }
    BinWriter(const BinWriter &) = delete;
    BinWriter &operator=(const BinWriter &) = delete;

    bool open(const std::string &path);
    // Records of reads 0..n_reads-1 (read k = vals[off[k] .. off[k+1])), laid out by `threads`
    // host threads and written with parallel pwrite.  OutT is the stored width (u16 / u8), T the
    // width of `vals` (u16 / u32 / u8): wider values saturate.
    template <typename OutT, typename T>
    bool batch(const std::vector<std::string> &names, const uint64_t *off, const T *vals, uint64_t n_reads, unsigned threads);
    bool close();

private:
    int fd_ = -1;
    uint64_t pos_ = 0;
    bool ok_ = true;
    std::vector<uint8_t> buf_;          // one batch's records (kept across batches)
    std::vector<uint64_t> rec_off_;
};

// Container -> the reference's text format (pml_to_vec, pml_query.cpp:78-85): what `col-bwt view`
// prints.  value_bytes = 2 for .pml.bin, 1 for .cid.bin.  Returns false on a malformed file.
bool binary_to_text(const std::string &bin_path, int value_bytes, const std::string &text_path, std::string &err);

}  // namespace colbwt
