// col_split -- command line of the reference's splitter (src/col_split.cpp:62-140; getopt string
// "rvlN:p:m:s:o:", common.hpp:231): col_split [-m tunnels|all] [-s rate] [-o overlap] [-d device] <prefix>
// reads <prefix>.bwt.heads, .bwt.len, .col_mums and writes <prefix>.col_runs, <prefix>.col_ids.
// build_FL (src/build_FL.cpp) is folded in: no .FL_table file is needed.  -o is accepted and, as in
// the reference (col_split.hpp:215), has no effect.
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>

#include "../../include/colbwt.h"

int main(int argc, char *const argv[]) {
    std::string mode = "tunnels";                 // the reference's default is "" (rejected, :41-43); the driver passes one (col-bwt.py:174)
    int rate = 1, device = 0, c;
    while ((c = getopt(argc, argv, "rvlN:p:m:s:o:d:")) != -1) {
        switch (c) {
            case 'm': mode = optarg; break;
            case 's': rate = atoi(optarg); break;
            case 'd': device = atoi(optarg); break;
            default: break;
        }
    }
    if (argc != optind + 1) {
        fprintf(stderr, "[ERROR]: Invalid number of arguments\nusage: col_split [-m tunnels|all] [-s rate] [-d device] <prefix>\n");
        return 1;
    }
    if (mode != "all" && mode != "tunnels") {
        fprintf(stderr, "[ERROR]: Invalid split mode: %s. Must be one of: all, tunnels\n", mode.c_str());
        return 1;
    }
    printf("[INFO] Splitting runs based on multi-MUM Positions using FL Table\n");
    const int rc = colbwt_col_split(argv[optind], mode == "all" ? COLBWT_SPLIT_ALL : COLBWT_SPLIT_TUNNELS, rate, device);
    if (rc != COLBWT_OK) {
        fprintf(stderr, "[ERROR]: %s\n", colbwt_col_split_error());
        return 1;
    }
    printf("[INFO] \tSplitting Complete\n[INFO] Serializing COL runs bitvector and IDs: %s.col_runs, %s.col_ids\n[INFO] Done\n",
           argv[optind], argv[optind]);
    return 0;
}
