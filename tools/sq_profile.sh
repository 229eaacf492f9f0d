#!/bin/bash
# SQ instruction / wait counters of the query kernel for both HBM layouts (GPU box).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/sq_${1:-a}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for LAY in ${LAYOUTS:-1 2 3}; do
  export COLBWT_LAYOUT=$LAY
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/l$LAY" -o b -- python3 "$REPO/bench.py" --no-cpu --steps 1 --warmup 0 > "$OUT/l$LAY.json" 2> "$OUT/l$LAY.err"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d "$OUT/m$LAY" -o b -- python3 "$REPO/bench.py" --no-cpu --steps 1 --warmup 0 > "$OUT/m$LAY.json" 2> "$OUT/m$LAY.err"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, json
out = sys.argv[1]; res = {}
for f in sorted(glob.glob(os.path.join(out, "*", "*counter_collection.csv"))):
    key = os.path.basename(os.path.dirname(f))[1:]
    for r in csv.DictReader(open(f)):
        if "query_kernel" in r["Kernel_Name"]:
            res.setdefault("layout" + key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
