#!/bin/bash
# Runs on the GPU box: gather_bench modes with timing, then PMC (read requests,
# FETCH_SIZE) for each, to calibrate bytes-per-random-load.  Output: gpurun_out/gather_<tag>/
TAG=${1:-g1}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/gather_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
GB=$REPO/tools/gather_bench
for spec in "0 0" "2 0" "3 0" "4 0" "5 0" "6 0" "7 0" "8 0" "0 1" "0 2"; do
  set -- $spec
  $GB 3072 10000000 150 $1 2 $2 >> "$OUT/timing.jsonl" 2>> "$OUT/timing.err"
done
for spec in "0 0" "3 0" "4 0" "5 0" "7 0" "0 1"; do
  set -- $spec
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d "$OUT/pmc_m$1_a$2" -o g -- $GB 3072 4000000 100 $1 1 $2 > /dev/null 2>> "$OUT/pmc.err"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmcf_m$1_a$2" -o g -- $GB 3072 4000000 100 $1 1 $2 > /dev/null 2>> "$OUT/pmc.err"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, json
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(os.path.join(out, "pmc*_m*", "*counter_collection.csv"))):
    key = os.path.basename(os.path.dirname(f)).split("_", 1)[1]
    for r in csv.DictReader(open(f)):
        if "chase" in r["Kernel_Name"]:
            res.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in res.items():
    v["steps_total"] = 4000000 // 256 * 256 * 100 if 4000000 % 256 == 0 else (4000000 // 256 + 1) * 256 * 100
    if "TCC_EA0_RDREQ_sum" in v: v["rdreq_per_step"] = v["TCC_EA0_RDREQ_sum"] / v["steps_total"]
    if "FETCH_SIZE" in v: v["fetch_bytes_per_step"] = v["FETCH_SIZE"] * 1024 / v["steps_total"]
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cat "$OUT/timing.jsonl"
