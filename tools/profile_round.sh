#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes for
# bench.py and the gather calibration; summaries land in gpurun_out/prof_<tag>/.
# usage: tools/profile_round.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="--no-cpu --steps 3 --warmup 1 $*"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 "$REPO/bench.py" $BENCH_ARGS > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err"
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -o bench -- python3 "$REPO/bench.py" --no-cpu --steps 1 --warmup 0 $* > "$OUT/pmc_$N.json" 2> "$OUT/pmc_$N.err"
  echo "pmc $N rc=$?"
done
