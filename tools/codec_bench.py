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
    # the col ids: 7 distinct values (the C2 recipe) -> 3 bit planes
    ids = __import__("numpy").array([0, 1, 2, 3, 17, 200, 255], dtype="uint8")
    cid = torch.from_numpy(ids)[torch.randint(0, 7, (words * 32,), generator=torch.Generator().manual_seed(2))].to(dev)
    planes = torch.zeros(3 * words, dtype=torch.int32, device=dev)
    back = torch.zeros(words * 32, dtype=torch.uint8, device=dev)
    for name, fn in (("cid_pack", lambda: pkg.cid_pack_device(cid.data_ptr(), nb, ids, planes.data_ptr(), s)),
                     ("cid_unpack", lambda: pkg.cid_unpack_device(planes.data_ptr(), 0, words, ids, back.data_ptr(), s))):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            fn()
        b.record()
        torch.cuda.synchronize()
        res[name + "_ms"] = a.elapsed_time(b) / 5
    res["cid_round_trip_ok"] = bool(torch.equal(back[:nb], cid[:nb]))
    res["cid_bits"] = 3
    res["bytes_per_base_on_the_wire"] = (1 + 3) / 8
    res["bases"] = nb
    res["pack_GBps"] = (2 * nb + nb / 8) / res["pack_ms"] / 1e6
    res["unpack_GBps"] = (2 * nb + nb / 4) / res["unpack_ms"] / 1e6
    print(json.dumps(res))


if __name__ == "__main__":
    main()
