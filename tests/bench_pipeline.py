#!/usr/bin/env python3
"""The whole pipeline on a pangenome in miniature, every stage the product's own:
FASTA files -> build_rlbwt (RLBWT, thresholds, multi-MUMs on the GPU) -> col_split -> build_col_bwt
-> query of reads sampled from the documents, with the stage times, the index shape and the query
rate on a TRUE index (long runs, real thresholds, col ids from real multi-MUMs).  A sample of the
reads is checked against the oracle.

    python tests/bench_pipeline.py [--docs 32 --length 4000000 --divergence 0.002 --reads 4000000]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_oracle, load_package  # noqa: E402  (lives under tests/: the oracle is the checker)

ACGT = np.frombuffer(b"ACGT", np.uint8)


def write_fasta(path, name, seq, width=80):
    n = len(seq)
    rows = n // width
    with open(path, "wb") as f:
        f.write(b">" + name + b"\n")
        if rows:
            body = np.empty((rows, width + 1), np.uint8)
            body[:, :width] = seq[:rows * width].reshape(rows, width)
            body[:, width] = 10
            f.write(body.tobytes())
        if n > rows * width:
            f.write(seq[rows * width:].tobytes() + b"\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=32)
    ap.add_argument("--length", type=int, default=4_000_000)
    ap.add_argument("--divergence", type=float, default=0.002)
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--error", type=float, default=0.01)
    ap.add_argument("--no-revcomp", action="store_true")
    ap.add_argument("--mode", default="tunnels")
    ap.add_argument("--sub-sample", type=int, default=10)
    ap.add_argument("--tmp", default=None)
    a = ap.parse_args()
    import torch
    pkg = load_package()
    rng = np.random.default_rng(1)
    tmp = tempfile.mkdtemp(dir=a.tmp)
    base = rng.choice(ACGT, size=a.length)
    seqs, paths = [], []
    for d in range(a.docs):
        s = base.copy()
        mut = rng.random(a.length) < a.divergence
        s[mut] = rng.choice(ACGT, size=int(mut.sum()))
        seqs.append(s)
        paths.append(os.path.join(tmp, f"hap{d}.fa"))
        write_fasta(paths[-1], b"hap%d" % d, s)
    prefix = os.path.join(tmp, "idx.fa")
    exe = lambda name: os.path.join(ROOT, "col-bwt_amd", name)
    out = {"docs": a.docs, "length": a.length, "divergence": a.divergence, "revcomp": not a.no_revcomp, "mode": a.mode,
           "sub_sample": a.sub_sample}
    stages = [("build_rlbwt", [exe("build_rlbwt"), "-l", "20", "-o", prefix] + ([] if a.no_revcomp else ["-r"]) + paths),
              ("col_split", [exe("col_split"), "-m", a.mode, "-s", str(a.sub_sample), prefix]),
              ("build_col_bwt", [exe("build_col_bwt"), prefix])]
    for name, cmd in stages:
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True)
        out[name + "_s"] = round(time.time() - t0, 2)
        if r.returncode != 0:
            sys.exit(f"{name} failed: {r.stdout[-2000:]}{r.stderr[-2000:]}")
        print(f"{name}: {out[name + '_s']} s", file=sys.stderr, flush=True)
        for line in r.stderr.splitlines():          # COLBWT_LOAD_TIMING=1: the stage's own breakdown
            if line.startswith("[colbwt load]"):
                print("   " + line, file=sys.stderr, flush=True)
    out["mums"] = (os.path.getsize(prefix + ".col_mums") - 5) // 10
    image = open(prefix + ".col_pml", "rb").read()
    t0 = time.time()
    tbl = pkg.ColPml.load(prefix)
    out["index_open_s"] = round(time.time() - t0, 2)
    info = tbl.info()
    out.update(n=int(info.n), bwt_runs=int(info.bwt_r), rows=int(info.r), n_over_r=round(info.n / info.bwt_r, 1),
               layout=int(info.layout), table_rows=int(info.table_rows), device_GB=round(info.device_bytes / 1e9, 2))
    # reads: substrings of the documents (forward strand) with substitutions
    n, m = a.reads, a.read_len
    which = rng.integers(0, a.docs, size=n)
    starts = rng.integers(0, a.length - m, size=n)
    allseq = np.stack(seqs)
    reads = allseq[which[:, None], starts[:, None] + np.arange(m)[None, :]]
    mut = rng.random(reads.shape) < a.error
    reads[mut] = rng.choice(ACGT, size=int(mut.sum()))
    dev = torch.device("cuda", 0)
    d_bases = torch.zeros(n * m + 128, dtype=torch.uint8, device=dev)
    d_bases[:n * m] = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * m
    d_pml = torch.zeros(n * m + 64, dtype=torch.int16, device=dev)
    d_cid = torch.zeros(n * m + 64, dtype=torch.uint8, device=dev)
    best = None
    for _ in range(4):
        st = tbl.query_device(d_bases.data_ptr(), d_off.data_ptr(), n, n * m, d_pml.data_ptr(), d_cid.data_ptr(), timed=True)
        best = st.kernel_ms if best is None else min(best, st.kernel_ms)
    pml = d_pml[:n * m].cpu().numpy().view(np.uint16)
    cid = d_cid[:n * m].cpu().numpy()
    k = min(n, 20_000)
    oracle = load_oracle()
    ep, ec = oracle.OracleIndex(image).query_batch(reads[:k].reshape(-1), np.arange(k + 1, dtype=np.uint64) * np.uint64(m), threads=8)
    out.update(reads=n, read_len=m, error=a.error, kernel_ms=round(best, 3), Gbase_s=round(n * m / best / 1e6, 2),
               reset_fraction=round(float((pml == 0).mean()), 4), mean_pml=round(float(pml.mean()), 1),
               bases_with_col_id=round(float((cid != 0).mean()), 4),
               matches_oracle_on_sample=bool(np.array_equal(pml[:k * m], ep) and np.array_equal(cid[:k * m], ec)))
    tbl.close()
    print(json.dumps(out))
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
