// pml_query -- drop-in command line of the reference's query executable
// (src/pml_query.cpp:92-143) over the MI355X engine (libcolbwt.so).
//
//   pml_query [-v] [-l] [-b|-B] [-d DEVICE[,DEVICE..]] -p <reads.fa|fq[.gz]> <index_prefix>
//
// Same getopt string as the reference ("rvlN:p:m:s:o:", common.hpp:231; the
// build-side options are accepted and ignored) plus -d and -b.  Reads
// <index_prefix>.col_pml, writes <pattern>.pml and <pattern>.cid in the
// reference's text format (pml_query.cpp:65-90).
//   -d 0,1,..  replicates the table on several devices and shards every batch of reads over
//              them (a device may be listed twice);
//   -b         writes <pattern>.pml.bin / <pattern>.cid.bin instead (the containers `col-bwt
//              query` produces; Movi-like, unverified: include/colbwt.h);
//   -B         writes both kinds, text first, from one load of the index.
//
// Deliberate deviations:
//   * a missing index / pattern argument or an unreadable file is fatal (exit 1);
//     the reference prints [ERROR] and continues into undefined behaviour
//     (pml_query.cpp:98-105, common.hpp:98-100);
//   * -l ("long pattern") produces the normal vec-mode files: the reference's
//     streaming mode keeps only the first read (col_bwt.hpp:477-495).
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <string>
#include <vector>

#include "../../include/colbwt.h"

static double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char *const argv[]) {
    std::string pattern;
    bool verbose = false, binary = false, text = true;
    std::vector<int> devices;
    int c;
    while ((c = getopt(argc, argv, "rvlN:p:m:s:o:d:bB")) != -1) {
        switch (c) {
            case 'v': verbose = true; break;
            case 'b': binary = true; text = false; break;
            case 'B': binary = true; text = true; break;
            case 'p': pattern = optarg; break;
            case 'd':
                for (const char *q = optarg; *q;) {
                    char *end = nullptr;
                    devices.push_back((int)strtol(q, &end, 10));
                    if (end == q) { devices.pop_back(); break; }
                    q = *end == ',' ? end + 1 : end;
                }
                break;
            case 'r': case 'l': case 'N': case 'm': case 's': case 'o': break;
            case '?': printf("ERROR: Unknown option.\n"); break;
        }
    }
    if (argc != optind + 1) {
        fprintf(stderr, "[ERROR]: Invalid number of arguments\n");
        fprintf(stderr, "usage: pml_query [-v] [-l] [-b|-B] [-d device[,device..]] -p <pattern FASTA/FASTQ[.gz]> <index_prefix>\n");
        return 1;
    }
    const std::string prefix = argv[optind];
    if (pattern.empty()) {
        fprintf(stderr, "[ERROR]: Pattern file not provided\n");
        return 1;
    }
    const double t_start = now_s();
    printf("[INFO] Loading BWT table supporting LF mapping: \n");
    colbwt_index *idx = nullptr;
    if (devices.empty()) devices.push_back(0);
    if (colbwt_index_open_devices(prefix.c_str(), nullptr, devices.data(), (int)devices.size(), COLBWT_LAYOUT_AUTO, &idx) !=
        COLBWT_OK) {
        fprintf(stderr, "[ERROR]: %s\n", colbwt_last_error());
        return 1;
    }
    colbwt_info info;
    colbwt_index_info(idx, &info);
    if (verbose) {  // col_bwt::bwt_stats, col_bwt.hpp:336-344
        printf("[LOG] Number of Col equal-letter runs: r = %llu\n", (unsigned long long)info.r);
        printf("[LOG] Number of BWT equal-letter runs: bwt_r = %llu\n", (unsigned long long)info.bwt_r);
        printf("[LOG] Length of complete BWT: n = %llu\n", (unsigned long long)info.n);
        printf("[LOG] Rate n/r = %g\n", (double)info.n / (double)info.r);
        printf("[LOG] HBM bytes held by the index (device %u, %u replica%s): %llu each\n", info.device, info.n_devices,
               info.n_devices == 1 ? "" : "s", (unsigned long long)info.device_bytes);
    }
    const double t_loaded = now_s();
    printf("[INFO] \tLoad Complete\n[INFO] \tElapsed time (s): %.6f\n", t_loaded - t_start);

    printf("[INFO] Computing PML Queries: \n");
    colbwt_stats st;
    std::string pml_name, cid_name;
    int qrc = COLBWT_OK;
    if (text) {                                            // pml_query.cpp:124-125
        pml_name = pattern + ".pml";
        cid_name = pattern + ".cid";
        qrc = colbwt_query_file(idx, pattern.c_str(), pml_name.c_str(), cid_name.c_str(), 0, &st);
        if (qrc == COLBWT_OK && binary)
            printf("[INFO] \tPMLs written to: %s\n[INFO] \tCIDs written to: %s\n", pml_name.c_str(), cid_name.c_str());
    }
    if (qrc == COLBWT_OK && binary) {
        pml_name = pattern + ".pml.bin";
        cid_name = pattern + ".cid.bin";
        qrc = colbwt_query_file_binary(idx, pattern.c_str(), pml_name.c_str(), cid_name.c_str(), 0, &st);
    }
    if (qrc != COLBWT_OK) {
        fprintf(stderr, "[ERROR]: %s\n", colbwt_last_error());
        colbwt_index_close(idx);
        return 1;
    }
    const double t_done = now_s();
    printf("[INFO] \tQuery Complete\n[INFO] \tElapsed time (s): %.6f\n", t_done - t_loaded);
    printf("[INFO] \tPMLs written to: %s\n[INFO] \tCIDs written to: %s\n", pml_name.c_str(), cid_name.c_str());
    if (verbose) {
        const double ks = st.kernel_ms * 1e-3;
        printf("[LOG] reads %llu, bases %llu; H2D %.3f ms, kernel %.3f ms, D2H %.3f ms\n",
               (unsigned long long)st.n_reads, (unsigned long long)st.n_bases, st.h2d_ms, st.kernel_ms, st.d2h_ms);
        if (ks > 0)
            printf("[LOG] kernel throughput %.3f Mbase/s, algorithmic %.2f GB/s\n", st.n_bases / ks / 1e6,
                   st.algorithmic_bytes / ks / 1e9);
    }
    printf("[INFO] Done\n[INFO] \tElapsed time (s): %.6f\n", t_done - t_start);
    colbwt_index_close(idx);
    return 0;
}
