#!/usr/bin/env python3
"""Ragged long-read batch (ONT-like length spread): time the query with and
without the length-sorted lane assignment (colbwt_query_device_ordered)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
import torch
pkg = load_package()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 600_000
max_len = 19_000
dev = torch.device("cuda", 0)
tbl = pkg.ColPml.from_bytes(pkg.synth_index(rows, 8, 0, 42))
d_full = torch.zeros(n_reads * max_len + 128, dtype=torch.uint8, device=dev)
d_off0 = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
tbl.synth_reads_device(n_reads, max_len, 50, 43, d_full.data_ptr(), d_off0.data_ptr(), 0)
g = torch.Generator(device="cpu"); g.manual_seed(1)
lens = torch.randint(1000, max_len + 1, (n_reads,), generator=g).to(dev)
parts = []
full2d = d_full[:n_reads * max_len].view(n_reads, max_len)
for a in range(0, n_reads, 20000):
    b = min(n_reads, a + 20000)
    m = torch.arange(max_len, device=dev)[None, :] < lens[a:b, None]
    parts.append(full2d[a:b][m])
bases = torch.cat(parts + [torch.zeros(128, dtype=torch.uint8, device=dev)])
del parts, full2d
off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(lens, 0)
nb = int(off[-1].item())
order = torch.argsort(lens, descending=True).to(torch.int32)
del d_full
out = {}
for name, o in (("unordered", None), ("ordered", order)):
    times = []
    for rep in range(3):
        p = torch.zeros(nb + 16, dtype=torch.int16, device=dev); c = torch.zeros(nb + 16, dtype=torch.uint8, device=dev)
        st = tbl.query_device(bases.data_ptr(), off.data_ptr(), n_reads, nb, p.data_ptr(), c.data_ptr(), 2, 0, timed=True,
                              d_order=o.data_ptr() if o is not None else None)
        times.append(st.kernel_ms)
    out[name] = (min(times), int(p[:nb].to(torch.int64).sum().item()), int(c[:nb].to(torch.int64).sum().item()))
print(json.dumps({"reads": n_reads, "bases": nb, "len_range": [1000, max_len],
                  "unordered_ms": out["unordered"][0], "ordered_ms": out["ordered"][0],
                  "Gbase_s_unordered": nb / out["unordered"][0] / 1e6, "Gbase_s_ordered": nb / out["ordered"][0] / 1e6,
                  "same_result": out["unordered"][1:] == out["ordered"][1:]}))
