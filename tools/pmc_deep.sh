#!/bin/bash
# Deeper PMC passes over the default query kernel (GPU box): where do the wave cycles go?
# usage: tools/pmc_deep.sh <tag>   -> gpurun_out/pmc_<tag>/summary.json
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_${1:-a}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  timeout -k 5 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o b -- python3 "$REPO/bench.py" --no-cpu --steps 1 --warmup 0 > "$OUT/$name.json" 2> "$OUT/$name.err" || return 1
}
# SQ passes first; the TCP / TCC passes run under their own short timeout (a GRBM_* + TA_* pass
# aborted inside rocprofv3 on this pool and then sat until the silence watchdog fired).
pass p1 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS &&
pass p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM
if [ -n "$DEEP_TCP" ]; then
pass p4 TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum &&
pass p5 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum &&
pass p6 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_avr TCC_TAG_STALL_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
fi
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, json
out = sys.argv[1]; res = {}
for f in sorted(glob.glob(os.path.join(out, "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "query_kernel" in r["Kernel_Name"]:
            res[r["Counter_Name"]] = res.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
