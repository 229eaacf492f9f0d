// Host side of the RLBWT / multi-MUM construction: documents -> text, results -> the files
// col_split and build_col_bwt read (formats: SURVEY.md Appendix A; col_bwt.hpp:167-171, 446-448;
// col_split.cpp:90-106).
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>

#include "fastx_reader.h"
#include "rlbwt_build.h"

namespace colbwt {

static uint8_t complement(uint8_t c) {
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        default: return c;   // N and anything else stays
    }
}

// One document: every record as it is + 1 (+ its reverse complement + 1).
static bool document_text(const std::string &path, bool revcomp, std::vector<uint8_t> &text, std::string &err) {
    FastxReader in;
    if (!in.open(path)) { err = "cannot open " + path; return false; }
    std::string name;
    uint64_t records = 0;
    for (;;) {
        const size_t at = text.size();
        if (!in.next(name, text)) break;
        ++records;
        const size_t end = text.size();
        for (size_t i = at; i < end; ++i)
            if (text[i] <= 1) { err = path + ": byte " + std::to_string(text[i]) + " in a sequence"; return false; }
        if (revcomp) {
            text.resize(end + 1 + (end - at) + 1);
            uint8_t *t = text.data();
            t[end] = 1;
            for (size_t i = 0; i < end - at; ++i) t[end + 1 + i] = complement(t[end - 1 - i]);
            t[text.size() - 1] = 1;
        } else {
            text.push_back(1);
        }
    }
    if (!records) { err = path + " holds no record"; return false; }
    return true;
}

bool text_from_fastas(const std::vector<std::string> &paths, bool revcomp, std::vector<uint8_t> &text,
                      std::vector<uint64_t> &doc_start, std::string &err) {
    text.clear();
    doc_start.clear();
    if (paths.empty()) { err = "no input files"; return false; }
    // the files are independent: a thread each (up to 16 at a time), then one copy into place
    const size_t n = paths.size();
    std::vector<std::vector<uint8_t>> part(n);
    std::vector<std::string> errs(n);
    std::vector<char> ok(n, 0);
    const size_t n_threads = std::min<size_t>(n, 16);
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t d; (d = next.fetch_add(1)) < n;) ok[d] = document_text(paths[d], revcomp, part[d], errs[d]);
    };
    {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_threads; ++t) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
    }
    uint64_t total = 1;
    for (size_t d = 0; d < n; ++d) {
        if (!ok[d]) { err = errs[d]; return false; }
        doc_start.push_back(total - 1);
        total += part[d].size();
    }
    text.resize(total);
    next = 0;
    auto place = [&] {
        for (size_t d; (d = next.fetch_add(1)) < n;) {
            memcpy(text.data() + doc_start[d], part[d].data(), part[d].size());
            std::vector<uint8_t>().swap(part[d]);
        }
    };
    {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_threads; ++t) pool.emplace_back(place);
        place();
        for (std::thread &t : pool) t.join();
    }
    text[total - 1] = 0;
    return true;
}

static void le5(std::vector<uint8_t> &buf, uint64_t v) {
    for (int i = 0; i < 5; ++i) buf.push_back((uint8_t)(v >> (8 * i)));
}

static bool write_all(const std::string &path, const uint8_t *p, size_t n) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(p, 1, n, f) == n;
    return (fclose(f) == 0) && ok;
}

bool write_rlbwt_files(const std::string &prefix, const RlbwtResult &res, uint32_t n_docs, std::string &err) {
    std::vector<uint8_t> lens, thr, mums;
    lens.reserve(5 * res.lens.size());
    thr.reserve(5 * res.thr.size());
    mums.reserve(5 + 10 * res.mum_len.size());
    for (size_t j = 0; j < res.lens.size(); ++j) { le5(lens, res.lens[j]); le5(thr, res.thr[j]); }
    le5(mums, n_docs);
    for (size_t j = 0; j < res.mum_len.size(); ++j) { le5(mums, res.mum_len[j]); le5(mums, res.mum_pos[j]); }
    const bool ok = write_all(prefix + ".bwt.heads", res.heads.data(), res.heads.size()) &&
                    write_all(prefix + ".bwt.len", lens.data(), lens.size()) && write_all(prefix + ".thr_pos", thr.data(), thr.size()) &&
                    write_all(prefix + ".col_mums", mums.data(), mums.size());
    if (!ok) err = "cannot write " + prefix + ".bwt.heads / .bwt.len / .thr_pos / .col_mums";
    return ok;
}

}  // namespace colbwt
