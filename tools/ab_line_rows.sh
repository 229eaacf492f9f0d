#!/bin/bash
# Line rows against three-step rows on the C2 workload, one shape per process (GPU box).
#   usage: tools/ab_line_rows.sh <tag> "<K> <K> ..." [extra ab_bench args]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}; SHAPES=${2:-"8"}; shift; shift
OUT=$REPO/gpurun_out/ab_line_rows_$TAG.jsonl
: > "$OUT"
for s in $SHAPES; do
  python3 "$REPO/tools/ab_bench.py" --reps 3 "$@" "$REPO/col-bwt_amd/libcolbwt.so@3" "$REPO/col-bwt_amd/libcolbwt.so@4:$s" >> "$OUT" 2>> "$OUT.err" || { tail -5 "$OUT.err"; exit 1; }
done
cat "$OUT"
