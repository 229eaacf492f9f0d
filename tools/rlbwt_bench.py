#!/usr/bin/env python3
"""Times the device construction of RLBWT + thresholds + multi-MUMs (csrc/rlbwt_build.hip) on
related random sequences: python tools/rlbwt_bench.py [--docs 16] [--length 1000000] [--rate 0.002]
[--revcomp].  One JSON line per run (append to profiles/ by hand)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=16)
    ap.add_argument("--length", type=int, default=1_000_000)
    ap.add_argument("--rate", type=float, default=0.002)
    ap.add_argument("--min-mum", type=int, default=20)
    ap.add_argument("--revcomp", action="store_true")
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    pkg = load_package()
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    base = rng.choice(acgt, size=a.length)
    text, starts = bytearray(), []
    for _ in range(a.docs):
        s = base.copy()
        mut = rng.random(a.length) < a.rate
        s[mut] = rng.choice(acgt, size=int(mut.sum()))
        starts.append(len(text))
        text += s.tobytes() + b"\x01"
        if a.revcomp:
            text += s.tobytes().translate(comp)[::-1] + b"\x01"
    text += b"\x00"
    text = bytes(text)
    best, res = None, None
    for _ in range(a.repeat):
        t0 = time.time()
        res = pkg.rlbwt_from_text(text, starts, min_mum=a.min_mum)
        dt = time.time() - t0
        best = dt if best is None else min(best, dt)
    print(json.dumps(dict(docs=a.docs, length=a.length, rate=a.rate, revcomp=a.revcomp, n=res["n"], runs=int(len(res["heads"])),
                          n_over_r=round(res["n"] / len(res["heads"]), 2), mums=int(len(res["mum_len"])),
                          mum_bases=int(res["mum_len"].sum()), rounds=res["rounds"], seconds=round(best, 3),
                          Mchar_s=round(res["n"] / best / 1e6, 1))))


if __name__ == "__main__":
    main()
