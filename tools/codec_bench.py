#!/usr/bin/env python3
"""Times the gather codec kernels (csrc/gather_codec.hip) at the C2 size on one GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    import torch
    pkg = load_package()
    dev = torch.device("cuda", 0)
    n_reads, m = 10_000_000, 150
    nb = n_reads * m
    words = (nb + 31) // 32
    g = torch.Generator(device=dev).manual_seed(1)
    # PML-shaped values: zero with probability 0.15
    pml = (torch.rand(words * 32, device=dev, generator=g) > 0.15).to(torch.int16)
    off = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * m
    zero = torch.zeros(words, dtype=torch.int32, device=dev)
    end = torch.zeros(words, dtype=torch.int32, device=dev)
    out = torch.zeros(words * 32, dtype=torch.int16, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    pkg.read_end_mask_device(off.data_ptr(), n_reads, end.data_ptr(), s)
    res = {}
    for name, fn in (("pack", lambda: pkg.pml_pack_device(pml.data_ptr(), nb, zero.data_ptr(), s)),
                     ("unpack", lambda: pkg.pml_unpack_device(zero.data_ptr(), end.data_ptr(), 0, words, words,
                                                              out.data_ptr(), s))):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            fn()
        b.record()
        torch.cuda.synchronize()
        res[name + "_ms"] = a.elapsed_time(b) / 5
    res["bases"] = nb
    res["pack_GBps"] = (2 * nb + nb / 8) / res["pack_ms"] / 1e6
    res["unpack_GBps"] = (2 * nb + nb / 4) / res["unpack_ms"] / 1e6
    print(json.dumps(res))


if __name__ == "__main__":
    main()
