#!/usr/bin/env python3
"""End-to-end rate of the file entry point (`colbwt_query_file` = `pml_query <index> <fasta>`,
pml_query.cpp:92-143) on the GPU box: FASTA parse -> batches -> GPU -> reference text files.

    python tools/file_path_bench.py [--rows N --reads N --read-len M]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=20_000_000)
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    a = ap.parse_args()
    import torch
    pkg = load_package()
    dev = torch.device("cuda", 0)
    image = pkg.synth_index(a.rows, 8, 0, 42)
    tbl = pkg.ColPml.from_bytes(image)
    n, m = a.reads, a.read_len
    d_bases = torch.zeros(n * m + 128, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    tbl.synth_reads_device(n, m, 10, 43, d_bases.data_ptr(), d_off.data_ptr())
    torch.cuda.synchronize()
    bases = d_bases[:n * m].cpu().numpy().reshape(n, m)
    tmp = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
    fa = os.path.join(tmp, "reads.fa")
    t0 = time.perf_counter()
    with open(fa, "wb") as f:                       # >r<k>\n<bases>\n
        hdr = np.char.add(np.char.add(">r", np.arange(n).astype("U")), "\n").astype("S")
        for lo in range(0, n, 200_000):
            hi = min(n, lo + 200_000)
            parts = []
            for k in range(lo, hi):
                parts.append(hdr[k])
                parts.append(bases[k].tobytes())
                parts.append(b"\n")
            f.write(b"".join(parts))
    t_write = time.perf_counter() - t0
    for mode, run, exts in (("text", tbl.query_file, (".pml", ".cid")),
                            ("binary", tbl.query_file_binary, (".pml.bin", ".cid.bin"))):
        res = []
        for rep in range(3):
            t0 = time.perf_counter()
            st = run(fa)
            dt = time.perf_counter() - t0
            res.append(dt)
        sz = {k: os.path.getsize(fa + k) for k in ("",) + exts}
        print(json.dumps({"mode": mode, "reads": n, "read_len": m, "rows": a.rows, "fasta_bytes": sz[""],
                          "pml_bytes": sz[exts[0]], "cid_bytes": sz[exts[1]], "wall_s": [round(x, 4) for x in res],
                          "Mbase_s": n * m / min(res) / 1e6, "kernel_ms": st.kernel_ms, "h2d_ms": st.h2d_ms, "d2h_ms": st.d2h_ms,
                          "fasta_gen_s": round(t_write, 1), "cpus": len(os.sched_getaffinity(0)),
                          "tmpdir": tmp}), flush=True)
        for k in exts:
            os.remove(fa + k)
    os.remove(fa)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
