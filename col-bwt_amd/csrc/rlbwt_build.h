// rlbwt_build.h -- FASTA documents -> RLBWT, thresholds, multi-MUMs.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace colbwt {

struct RlbwtResult {
    uint64_t n = 0;                       // BWT length = text length
    int rounds = 0;                       // prefix-doubling rounds
    std::vector<uint8_t> heads;           // one character per BWT run
    std::vector<uint64_t> lens, thr;      // run lengths; threshold position of every run
    std::vector<uint64_t> mum_len, mum_pos;   // multi-MUMs, ascending by position (suffix-array rank)
};

// text[0..n): records, each followed by a separator 1; the last character is the text's only 0.
// doc_start[d]: first character of document d (ascending from 0).  The device does all of it.
int rlbwt_from_text(const uint8_t *text, uint64_t n, const uint64_t *doc_start, uint32_t n_docs, uint64_t min_mum,
                    int device, RlbwtResult &out, std::string &err);

// The text of the FASTA/FASTQ(.gz) files `paths` (one document per file): every record's bases as
// they are, then 1; with revcomp also the record's reverse complement, then 1; a final 0.
// false: a file cannot be read, holds no record, or holds a byte <= 1.
bool text_from_fastas(const std::vector<std::string> &paths, bool revcomp, std::vector<uint8_t> &text,
                      std::vector<uint64_t> &doc_start, std::string &err);

// <prefix>.bwt.heads, .bwt.len, .thr_pos, .col_mums
bool write_rlbwt_files(const std::string &prefix, const RlbwtResult &res, uint32_t n_docs, std::string &err);

}  // namespace colbwt
