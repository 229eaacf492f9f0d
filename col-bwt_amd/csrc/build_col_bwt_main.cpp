// build_col_bwt -- command line of the reference's in-repo builder
// (src/build_col_bwt.cpp:14-63): build_col_bwt <prefix> reads the split RLBWT
// files next to <prefix> and writes <prefix>.col_pml.
#include <stdio.h>

#include "../../include/colbwt.h"

int main(int argc, char **argv) {
    if (argc != 2) {
        fprintf(stderr, "[ERROR]: Invalid number of arguments\nusage: build_col_bwt <prefix>\n");
        return 1;
    }
    printf("[INFO] Building Col BWT supporting PML queries: \n");
    const int rc = colbwt_build_col_pml(argv[1], nullptr);
    if (rc != COLBWT_OK) {
        fprintf(stderr, "[ERROR]: cannot build %s.col_pml (rc=%d): missing or malformed "
                        ".bwt.heads/.bwt.len/.col_ids/.col_runs/.thr_pos\n", argv[1], rc);
        return 1;
    }
    printf("[INFO] \tConstruction Complete\n[INFO] \tSerializing Complete: %s.col_pml\n[INFO] Done\n", argv[1]);
    return 0;
}
