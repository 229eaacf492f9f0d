// fasta_parallel.h -- many-threaded reader for the common case of the pattern file: a plain
// (uncompressed) FASTA.  Same records as FastxReader (the kseq semantics of io.hpp:6-35): the file
// is cut at lines that start with '>' and the pieces are parsed side by side.  The sequential
// parser manages about 1.4 Gbase/s, which bounds pml_query end to end once the results are
// written in binary.  A line that starts with '@' or '+' means FASTQ-style records may follow
// (a quality line may begin with '>'): the reader then reports where it stopped and the caller
// goes on with FastxReader from that record.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace colbwt {

class ParallelFasta {
public:
    ParallelFasta() = default;
    ~ParallelFasta();
    ParallelFasta(const ParallelFasta &) = delete;
    ParallelFasta &operator=(const ParallelFasta &) = delete;

    // true when `path` is a regular, non-gzip file whose first byte is '>' (else use FastxReader)
    bool open(const std::string &path);
    enum Result { kBatch, kEnd, kNotPlainFasta };
    // Appends the next records holding about `target_bases` bases: names, bases (concatenated),
    // off (end offset of every record, relative to bases.size() on entry ... pushed as absolute
    // positions in `bases`), max_len.  kNotPlainFasta: nothing appended, position() is the byte
    // offset of the record to resume from.
    Result next_batch(uint64_t target_bases, unsigned threads, std::vector<std::string> &names, std::vector<uint8_t> &bases,
                      std::vector<uint64_t> &off, uint64_t &max_len);
    uint64_t position() const { return cur_; }

private:
    uint64_t next_record_start(uint64_t from) const;   // first p >= from with data[p-1] == '\n' && data[p] == '>', or size
    const uint8_t *data_ = nullptr;
    uint64_t size_ = 0, cur_ = 0;
    int fd_ = -1;
};

}  // namespace colbwt
