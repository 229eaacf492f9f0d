#!/usr/bin/env python3
"""Randomised parity fuzz of the emulated HIP sources (all HBM layouts) vs the
oracle: random move tables (alphabet size 2..9, sub-run splits, thresholds
inside rows, occasional very long rows that the K-step build must cut) and
random / walk reads.  Run by tests/test_emu_parity.py under ASan."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_oracle, load_package  # noqa: E402
import helpers  # noqa: E402

pkg = load_package()
pkg.LIB_PATH = os.path.join(HERE, "libcolbwt_emu.so")
oracle = load_oracle()


def one(seed):
    rng = np.random.default_rng(seed)
    sigma = int(rng.integers(2, 10))
    alpha = bytes(rng.choice(np.arange(1, 128), size=sigma, replace=False).astype(np.uint8).tolist())
    r = int(rng.integers(3, 900))
    max_len = int(rng.choice([1, 2, 9, 40, 300]))
    img = helpers.random_table(rng, r, alphabet=alpha, max_len=max_len, split_prob=float(rng.choice([0, 0.1, 0.5])))
    if seed % 5 == 0:     # a few rows far beyond the 16-bit length field / the 65534 cut of the K-step rows
        t = helpers.unpack_col_pml(img)
        lens = np.diff(np.append(t["idx"].astype(np.int64), t["n"]))
        lens[rng.integers(0, r, size=2)] = rng.integers(65530, 200000, size=2)
        idx = np.concatenate(([0], np.cumsum(lens)[:-1]))
        n = int(lens.sum())
        itv, off = helpers.lf_columns(t["char"], idx, n)
        img = helpers.pack_col_pml(int(t["bwt_r"]), n, t["char"], idx, itv, off & np.uint64(0xFFFF), t["cid"],
                                   rng.integers(0, n, size=r))
    img = bytes(img)
    reads = [rng.choice(np.frombuffer(alpha + b"\xfe", np.uint8), size=int(m)) for m in rng.integers(0, 70, size=40)]
    reads += helpers.backward_walk_reads(img, 25, int(rng.integers(1, 90)), 0.03, seed=seed)
    bases, off = helpers.concat_reads(reads)
    ep, ec = oracle.OracleIndex(img).query_batch(bases, off)
    # line rows (4) and line rows with mismatch lines (5), at varying depths
    k = (4 + seed // 3 % 5) << 8                     # look-ahead depth 4 .. 8
    fat = {0: (4 | k, 5), 1: (5 | k, 6), 2: (6 | k, 4)}[seed % 3]
    for layout in (1, 2, 3) + fat:
        tbl = pkg.ColPml.from_bytes(img, layout=layout)
        p, c, _ = tbl.query_batch(bases, off)
        assert np.array_equal(p, ep) and np.array_equal(c, ec), f"seed {seed} layout {layout} sigma {sigma} r {r}"
        tbl.close()


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    for seed in range(lo, hi):
        one(seed)
    print(f"FUZZ-OK {lo}..{hi}")


if __name__ == "__main__":
    main()
