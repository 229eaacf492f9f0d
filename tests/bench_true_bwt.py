#!/usr/bin/env python3
"""Query rate on a TRUE BWT index of H related sequences (a pangenome in miniature: long
runs, min-LCP thresholds), built by the test helpers' numpy suffix-array construction.

    python tests/bench_true_bwt.py [--haplotypes 32 --length 1000000 --reads 2000000]
The index is small (it fits the caches) -- this measures the kernel on a realistic run /
threshold structure, not the HBM-bound C2 regime; results are checked against the oracle
on a sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # lives under tests/: it calls the oracle as checker
from __graft_entry__ import load_oracle, load_package  # noqa: E402
import helpers  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--haplotypes", type=int, default=32)
    ap.add_argument("--length", type=int, default=1_000_000)
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    a = ap.parse_args()
    import torch
    pkg = load_package()
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    base = rng.choice(acgt, size=a.length)
    seqs = [bytes(base)]
    for _ in range(a.haplotypes - 1):
        s = base.copy()
        mut = rng.random(len(s)) < 0.01
        s[mut] = rng.choice(acgt, size=int(mut.sum()))
        seqs.append(bytes(s))
    t0 = time.time()
    image, text = helpers.true_bwt_index_large(seqs, seed=2, extra_splits=a.length // 50)
    t_build = time.time() - t0
    print(f"index built in {t_build:.0f} s", file=sys.stderr, flush=True)
    hdr = helpers.unpack_col_pml(image)
    dev = torch.device("cuda", 0)
    n, m = a.reads, a.read_len
    body = np.frombuffer(text, np.uint8)
    starts = rng.integers(0, len(body) - m - 1, size=n)
    reads = body[starts[:, None] + np.arange(m)[None, :]].copy()
    reads[reads <= 1] = ord("A")
    mut = rng.random(reads.shape) < 0.01
    reads[mut] = rng.choice(acgt, size=int(mut.sum()))
    d_bases = torch.zeros(n * m + 128, dtype=torch.uint8, device=dev)
    d_bases[:n * m] = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_off = (torch.arange(n + 1, dtype=torch.int64, device=dev) * m)
    d_pml = torch.zeros(n * m + 16, dtype=torch.int16, device=dev)
    d_cid = torch.zeros(n * m + 16, dtype=torch.uint8, device=dev)
    out = {"haplotypes": a.haplotypes, "length": a.length, "n": int(hdr["n"]), "bwt_runs": int(hdr["bwt_r"]),
           "rows": int(hdr["r"]), "reads": n, "read_len": m, "index_build_s": round(t_build, 1)}
    oracle = load_oracle()
    k = 20_000
    ep, ec = oracle.OracleIndex(bytes(image)).query_batch(reads[:k].reshape(-1), np.arange(k + 1, dtype=np.uint64) * np.uint64(m), threads=8)
    for layout in (3, 1):
        tbl = pkg.ColPml.from_bytes(bytes(image), layout=layout)
        best = None
        for _ in range(4):
            st = tbl.query_device(d_bases.data_ptr(), d_off.data_ptr(), n, n * m, d_pml.data_ptr(), d_cid.data_ptr(),
                                  timed=True)
            best = st.kernel_ms if best is None else min(best, st.kernel_ms)
        ok = bool(np.array_equal(d_pml[:k * m].cpu().numpy().view(np.uint16), ep)
                  and np.array_equal(d_cid[:k * m].cpu().numpy(), ec))
        out[f"layout{layout}"] = {"kernel_ms": round(best, 3), "Gbase_s": round(n * m / best / 1e6, 2),
                                  "table_rows": int(tbl.info().table_rows), "matches_oracle_on_sample": ok,
                                  "mean_pml": float(d_pml[:n * m].float().mean())}
        tbl.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
