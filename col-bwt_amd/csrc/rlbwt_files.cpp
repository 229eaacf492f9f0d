// Host side of the RLBWT / multi-MUM construction: documents -> text, results -> the files
// col_split and build_col_bwt read (formats: SURVEY.md Appendix A; col_bwt.hpp:167-171, 446-448;
// col_split.cpp:90-106).
#include <stdio.h>

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

bool text_from_fastas(const std::vector<std::string> &paths, bool revcomp, std::vector<uint8_t> &text,
                      std::vector<uint64_t> &doc_start, std::string &err) {
    text.clear();
    doc_start.clear();
    for (const std::string &path : paths) {
        FastxReader in;
        if (!in.open(path)) { err = "cannot open " + path; return false; }
        doc_start.push_back(text.size());
        std::string name;
        uint64_t records = 0;
        for (;;) {
            const size_t at = text.size();
            if (!in.next(name, text)) break;
            ++records;
            for (size_t i = at; i < text.size(); ++i)
                if (text[i] <= 1) { err = path + ": byte " + std::to_string(text[i]) + " in a sequence"; return false; }
            const size_t end = text.size();
            text.push_back(1);
            if (revcomp) {
                for (size_t i = end; i > at; --i) text.push_back(complement(text[i - 1]));
                text.push_back(1);
            }
        }
        if (!records) { err = path + " holds no record"; return false; }
    }
    if (paths.empty()) { err = "no input files"; return false; }
    text.push_back(0);
    return true;
}

static bool put5(FILE *f, uint64_t v) {
    uint8_t b[5];
    for (int i = 0; i < 5; ++i) b[i] = (uint8_t)(v >> (8 * i));
    return fwrite(b, 1, 5, f) == 5;
}

bool write_rlbwt_files(const std::string &prefix, const RlbwtResult &res, uint32_t n_docs, std::string &err) {
    struct Out {
        FILE *f = nullptr;
        ~Out() { if (f) fclose(f); }
    } heads, lens, thr, mums;
    heads.f = fopen((prefix + ".bwt.heads").c_str(), "wb");
    lens.f = fopen((prefix + ".bwt.len").c_str(), "wb");
    thr.f = fopen((prefix + ".thr_pos").c_str(), "wb");
    mums.f = fopen((prefix + ".col_mums").c_str(), "wb");
    bool ok = heads.f && lens.f && thr.f && mums.f;
    ok = ok && fwrite(res.heads.data(), 1, res.heads.size(), heads.f) == res.heads.size();
    for (size_t j = 0; ok && j < res.lens.size(); ++j) ok = put5(lens.f, res.lens[j]) && put5(thr.f, res.thr[j]);
    ok = ok && put5(mums.f, n_docs);
    for (size_t j = 0; ok && j < res.mum_len.size(); ++j) ok = put5(mums.f, res.mum_len[j]) && put5(mums.f, res.mum_pos[j]);
    for (Out *o : {&heads, &lens, &thr, &mums})
        if (o->f) { ok = (fclose(o->f) == 0) && ok; o->f = nullptr; }
    if (!ok) err = "cannot write " + prefix + ".bwt.heads / .bwt.len / .thr_pos / .col_mums";
    return ok;
}

}  // namespace colbwt
