#!/usr/bin/env python3
"""What the line-row kernel's trips are spent on (GPU box; needs the experiment build
`make -C col-bwt_amd variant TAG=stats VFLAGS=-DCOLBWT_COUNT_TRIPS`).

    python tools/trip_stats.py [--rows N --reads N --read-len M --steps K]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=200_000_000)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sub-permille", type=int, default=10)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--layout", type=int, default=4, help="4 = line rows, 5 / 6 = line rows with (deep) mismatch lines")
    a = ap.parse_args()
    import torch
    pkg = load_package()
    pkg.LIB_PATH = os.path.join(ROOT, "col-bwt_amd", "libcolbwt_stats.so")
    dev = torch.device("cuda", 0)
    image = pkg.synth_index(a.rows, 8, 0, 42)
    tbl = pkg.ColPml.from_bytes(image, layout=a.layout | (a.steps << 8))
    n, m = a.reads, a.read_len
    d_bases = torch.zeros(n * m + 128, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    tbl.synth_reads_device(n, m, a.sub_permille, 43, d_bases.data_ptr(), d_off.data_ptr())
    d_pml = torch.zeros(n * m + 16, dtype=torch.int16, device=dev)
    d_cid = torch.zeros(n * m + 16, dtype=torch.uint8, device=dev)
    L = pkg.lib()
    out = (C.c_ulonglong * 16)()
    stats_fn = L.colbwt_debug_fat2_stats if a.layout in (5, 6) else L.colbwt_debug_fat_stats
    stats_fn(out, 1)
    st = tbl.query_device(d_bases.data_ptr(), d_off.data_ptr(), n, n * m, d_pml.data_ptr(), d_cid.data_ptr(), timed=True)
    torch.cuda.synchronize()
    stats_fn(out, 1)
    if a.layout in (5, 6):
        names = ["row_trips", "fast_forward", "entry_trips", "scan", "absent", "idle", "row_to_entry", "unused"]
    else:
        names = ["live", "fast_forward", "slot", "scan", "absent", "idle", "chunk_ends", "skip_arrivals"]
    d = {k: int(v) for k, v in zip(names, out)}
    wave_trips = max(int(out[8]), 1)
    clocks = {"wave_trips": wave_trips, "boundary": out[12] / wave_trips, "rows_issue": out[13] / wave_trips,
              "flush": out[14] / wave_trips, "other_requests": out[9] / wave_trips, "wait": out[10] / wave_trips,
              "compute": out[11] / wave_trips} if a.layout not in (5, 6) else {}
    d.update(kernel_ms=st.kernel_ms, per_read={k: round(v / n, 3) for k, v in d.items()}, rows=int(tbl.info().table_rows),
             steps=a.steps, layout=a.layout, clocks=clocks, resets=float((d_pml[:n * m] == 0).float().mean().item()))
    print(json.dumps(d))


if __name__ == "__main__":
    main()
