#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (tools/profile_round.sh) into
profiles/<tag>_kernel_stats.csv + profiles/<tag>_summary.json.

HBM traffic per launch follows MI355X_MICROARCH.md (HBM / rocprofv3 PMC):
FETCH_SIZE and WRITE_SIZE are collected in separate passes and are in KiB;
on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B although every read request is a
128-byte line fill, so the read side is DOUBLED (the guide's gfx950
correction).  That the correction also holds for this kernel's pattern
(random 16-byte row loads) was calibrated with tools/gather_bench
(profiles/r01_gather_calibration.json): a known number of dependent random
16-byte loads reads FETCH_SIZE = 64 B per load, and a second load in the
other 64-byte half of the same 128-byte line adds NO read request, while one
in the next line adds exactly one.  WRITE_SIZE is taken as is (32/64-byte
write requests are tallied at their size).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


KERNEL_SOURCES = ("fat2_query.hip", "fat_cursor.h", "lane_out.h", "fat_query.hip", "fat_layout.h", "fat_build.hip", "lane_io.h", "sk_query.hip", "sk_layout.h", "query_kernels.hip",
                  "lf_device.h", "device_layout.h")


def kernel_sources_sha(root):
    """sha256 over the sources that decide the query kernels' memory traffic: bench.py reports the
    recorded traffic only while this still matches (no git on the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(root, "col-bwt_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def main():
    tag = sys.argv[1]
    if tag.startswith("-"):
        sys.exit(f"summarize_profile.py: '{tag}' is not a profile tag")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(root, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    summary = {"tag": tag, "kernels": {}}
    if stats:
        for r in csv.DictReader(open(stats[0])):
            if "colbwt" in r["Name"]:
                summary["kernels"].setdefault(r["Name"][:80], {}).update(
                    calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), total_ns=float(r["TotalDurationNs"]))
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv"))):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "colbwt" not in r["Kernel_Name"]:
                continue
            a = agg[(r["Kernel_Name"][:80], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
        for (k, c), (v, n) in agg.items():
            summary["kernels"].setdefault(k, {}).setdefault("pmc_per_launch", {})[c] = v / n
    for k, d in summary["kernels"].items():
        p = d.get("pmc_per_launch", {})
        if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
            d["hbm_read_bytes_per_launch"] = p["FETCH_SIZE"] * 1024 * 2   # 128-byte lines tallied at 64
            d["hbm_write_bytes_per_launch"] = p["WRITE_SIZE"] * 1024
            d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
    bt = os.path.join(src, "bench_trace.json")
    if os.path.exists(bt):
        try:
            summary["bench_line_under_trace"] = json.loads(open(bt).read().strip().splitlines()[-1])
        except Exception:
            pass
    json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
    q = sorted([d for k, d in summary["kernels"].items() if "query_kernel" in k],
               key=lambda d: -d.get("total_ns", 0))
    if q and "hbm_bytes_per_launch" in q[0] and len(sys.argv) > 2 and sys.argv[2] == "--set-traffic":
        import subprocess
        try:
            commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
        except Exception:
            commit = None
        kname = [k for k, d in summary["kernels"].items() if d is q[0]][0]
        json.dump({"tag": tag, "kernel": kname, "taken_at_commit": commit, "kernel_sources_sha": kernel_sources_sha(root),
                   "hbm_bytes_per_launch": q[0]["hbm_bytes_per_launch"],
                   "hbm_read_bytes_per_launch": q[0]["hbm_read_bytes_per_launch"],
                   "hbm_write_bytes_per_launch": q[0]["hbm_write_bytes_per_launch"],
                   "read_requests_per_launch": q[0]["pmc_per_launch"].get("TCC_EA0_RDREQ_sum"),
                   "write_requests_per_launch": q[0]["pmc_per_launch"].get("TCC_EA0_WRREQ_sum")},
                  open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(json.dumps(summary["kernels"], indent=1)[:3000])


if __name__ == "__main__":
    main()
