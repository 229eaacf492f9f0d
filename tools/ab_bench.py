#!/usr/bin/env python3
"""A/B timing of libcolbwt build variants on one GPU (experiment tool).

    python tools/ab_bench.py [--rows N --reads N --read-len M --reps K] lib1.so lib2.so ...

Each variant loads the same synthetic index, runs the query on the same
device-resident reads (sampled once), interleaved `reps` times; prints one JSON
line per variant with the HIP-event kernel time and a checksum of the outputs
(all variants must agree).
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


class Stats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("h2d_ms", C.c_double),
                ("kernel_ms", C.c_double), ("d2h_ms", C.c_double), ("algorithmic_bytes", C.c_uint64)]


def bind(path):
    L = C.CDLL(path)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.colbwt_last_error.restype = C.c_char_p
    L.colbwt_index_open_memory_layout.argtypes = [vp, u64, vp, i32, i32, C.POINTER(vp)]
    L.colbwt_index_info.argtypes = [vp, vp]
    L.colbwt_query_device.argtypes = [vp, vp, vp, u64, u64, vp, i32, vp, vp, C.POINTER(Stats)]
    L.colbwt_synth_reads_device.argtypes = [vp, u64, C.c_uint32, C.c_uint32, u64, vp, vp, vp]
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=200_000_000)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sub-permille", type=int, default=10)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--pml-bytes", type=int, default=2, help="2 = u16 PML (reads <= 65535 bases), 4 = u32")
    ap.add_argument("--thr-mode", type=int, default=0, help="0: thresholds uniform in [0,n) (C2 recipe); 1: between runs")
    ap.add_argument("--split-permille", type=int, default=0, help="rows that are sub-run splits (C5: 100)")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    import torch
    pkg = load_package()
    dev = torch.device("cuda", 0)
    image = pkg.synth_index(a.rows, 8, a.split_permille, 42, a.thr_mode)
    n_reads, m = a.reads, a.read_len
    nb = n_reads * m
    d_bases = torch.zeros(nb + 128, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    variants = []
    def layout_arg(lay):                     # "3", or "4:6" = line rows with K = 6 look-ahead steps
        kind, _, steps = lay.partition(":")
        return int(kind or 0) | (int(steps) << 8 if steps else 0)

    for spec in a.libs:                      # "lib.so" or "lib.so@2" (HBM table layout)
        path, _, lay = spec.partition("@")
        L = bind(os.path.abspath(path))
        h = C.c_void_p()
        rc = L.colbwt_index_open_memory_layout(image.ctypes.data, image.size, None, 0, layout_arg(lay), C.byref(h))
        assert rc == 0, L.colbwt_last_error()
        info = pkg.Info()
        L.colbwt_index_info(h, C.byref(info))
        variants.append((os.path.basename(spec), L, h, [], {"layout": int(info.layout), "steps": int(info.layout_shape >> 8),
                                                             "table_rows": int(info.table_rows), "hbm_GB": round(info.device_bytes / 1e9, 1)}))
    name0, L0, h0, _, _ = variants[0]
    assert L0.colbwt_synth_reads_device(h0, n_reads, m, a.sub_permille, 43, d_bases.data_ptr(), d_off.data_ptr(), None) == 0
    torch.cuda.synchronize()
    sums = {}
    for rep in range(a.reps + 1):
        for name, L, h, times, _ in variants:
            d_pml = torch.zeros(nb + 16, dtype=torch.int16 if a.pml_bytes == 2 else torch.int32, device=dev)
            d_cid = torch.zeros(nb + 16, dtype=torch.uint8, device=dev)
            st = Stats()
            rc = L.colbwt_query_device(h, d_bases.data_ptr(), d_off.data_ptr(), n_reads, nb, d_pml.data_ptr(), a.pml_bytes,
                                       d_cid.data_ptr(), None, C.byref(st))
            assert rc == 0, L.colbwt_last_error()
            if rep:
                times.append(st.kernel_ms)
            else:
                step = 1 << 28                      # in slices: an int64 copy of 1e10 values would not fit beside the index
                sums[name] = (sum(int(d_pml[a:min(a + step, nb)].to(torch.int64).sum().item()) for a in range(0, nb, step)),
                              sum(int(d_cid[a:min(a + step, nb)].to(torch.int64).sum().item()) for a in range(0, nb, step)))
            del d_pml, d_cid
    ref = sums[variants[0][0]]
    # a checksum only says something when there is another variant to compare with: with a single
    # variant it would be compared with itself -- null then (results are checked in tests/, not here)
    for k, (name, L, h, times, shape) in enumerate(variants):
        print(json.dumps({"lib": name, "ms": round(float(np.mean(times)), 3), "min_ms": round(min(times), 3),
                          "Gbase_s": round(nb / np.mean(times) / 1e6, 3),
                          "checksum_ok": (sums[name] == ref) if len(variants) > 1 else None,
                          "checksum_against": variants[0][0] if len(variants) > 1 and k else None,
                          "checksum": list(sums[name]),   # sums of all PML values / col ids: comparable across processes
                          "index": shape, "rows": a.rows, "reads": n_reads, "read_len": m, "pml_bytes": a.pml_bytes}), flush=True)


if __name__ == "__main__":
    main()
