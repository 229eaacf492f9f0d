#!/usr/bin/env python3
"""Extended randomised parity sweep on the GPU box (beyond the 50 tables of the -m gpu suite):
    python tests/fuzz_more.py <first_seed> <last_seed>     # steps of 25 seeds, 3 layouts each
"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from __graft_entry__ import load_package, load_oracle
import helpers
import test_gpu_parity as t
pkg, oracle = load_package(), load_oracle()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for s0 in range(lo, hi, 25):
    t.test_fuzz_random_tables_all_layouts(pkg, oracle, s0)
    print("ok", s0, flush=True)
