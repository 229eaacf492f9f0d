// build_rlbwt -- the step the reference's driver runs as `mumemto mum -K -R -T -l <min> [-r]
// -o <output> (-i <filelist> | fastas...)` (scripts/col-bwt.py:121-145):
//   build_rlbwt [-l min_mum] [-r] [-d device] [-i filelist] -o <prefix> [fastas...]
// writes <prefix>.bwt.heads, .bwt.len, .thr_pos, .col_mums (one document per file; -r adds every
// record's reverse complement, as the driver's own -r says).  See include/colbwt.h for the
// conventions; mumemto itself is not part of the reference tree.
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>

#include <fstream>
#include <string>
#include <vector>

#include "../../include/colbwt.h"

int main(int argc, char *const argv[]) {
    std::string out, list;
    unsigned long long min_mum = 20;   // col-bwt.py:218
    int revcomp = 0, device = 0, c;
    while ((c = getopt(argc, argv, "o:i:l:rd:vK")) != -1) {
        switch (c) {
            case 'o': out = optarg; break;
            case 'i': list = optarg; break;
            case 'l': min_mum = strtoull(optarg, nullptr, 10); break;
            case 'r': revcomp = 1; break;
            case 'd': device = atoi(optarg); break;
            default: break;
        }
    }
    std::vector<std::string> files;
    if (!list.empty()) {
        std::ifstream in(list);
        if (!in) { fprintf(stderr, "[ERROR]: cannot read %s\n", list.c_str()); return 1; }
        std::string line;
        while (std::getline(in, line)) {                       // "path [whitespace anything]" per line
            const size_t e = line.find_first_of(" \t\r");
            if (e != std::string::npos) line.resize(e);
            if (!line.empty()) files.push_back(line);
        }
    } else {
        for (int i = optind; i < argc; ++i) files.push_back(argv[i]);
    }
    if (out.empty() || files.empty()) {
        fprintf(stderr, "usage: build_rlbwt [-l min_mum] [-r] [-d device] [-i filelist] -o <prefix> [fastas...]\n");
        return 1;
    }
    std::vector<const char *> ptrs;
    for (const std::string &f : files) ptrs.push_back(f.c_str());
    printf("[INFO] Number of documents: %zu\n[INFO] Building RLBWT, thresholds and multi-MUMs (min length %llu%s)\n", files.size(), min_mum,
           revcomp ? ", with reverse complements" : "");
    colbwt_rlbwt *res = nullptr;
    const int rc = colbwt_rlbwt_build_files(ptrs.data(), (uint32_t)ptrs.size(), revcomp, min_mum, device, out.c_str(), &res);
    if (rc != COLBWT_OK) {
        fprintf(stderr, "[ERROR]: %s\n", colbwt_rlbwt_error());
        return 1;
    }
    colbwt_rlbwt_view v;
    colbwt_rlbwt_get(res, &v);
    printf("[INFO] \tn = %llu, r = %llu, multi-MUMs: %llu (%d doubling rounds)\n[INFO] Written: %s.bwt.heads, .bwt.len, .thr_pos, .col_mums\n",
           (unsigned long long)v.n, (unsigned long long)v.n_runs, (unsigned long long)v.n_mums, v.rounds, out.c_str());
    colbwt_rlbwt_free(res);
    return 0;
}
