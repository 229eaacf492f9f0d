#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (colbwt_query_batch):
reads in host memory -> PML/CID in host memory, C2 workload."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
import torch
pkg = load_package()
rows, n_reads, m = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000, 150
image = pkg.synth_index(rows, 8, 0, 42)
tbl = pkg.ColPml.from_bytes(image)
dev = torch.device("cuda", 0)
d_bases = torch.zeros(n_reads * m + 128, dtype=torch.uint8, device=dev)
d_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
tbl.synth_reads_device(n_reads, m, 10, 43, d_bases.data_ptr(), d_off.data_ptr(), 0)
torch.cuda.synchronize()
bases = d_bases[:n_reads * m].cpu().numpy()
off = d_off.cpu().numpy().astype(np.uint64)
for rep in range(3):
    t0 = time.perf_counter()
    pml, cid, st = tbl.query_batch(bases, off)
    dt = time.perf_counter() - t0
    print(json.dumps({"rep": rep, "wall_s": round(dt, 4), "Gbase_s_wall": round(n_reads * m / dt / 1e9, 3),
                      "h2d_ms": round(st.h2d_ms, 2), "kernel_ms": round(st.kernel_ms, 2), "d2h_ms": round(st.d2h_ms, 2)}), flush=True)
