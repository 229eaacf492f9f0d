#!/bin/bash
# Line rows: reads per chunk of the persistent lanes (GPU box).  usage: tools/ab_chunks.sh <tag> "<R> <R> .."
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}; RS=${2:-"1 2 4 8"}; shift; shift
OUT=$REPO/gpurun_out/ab_chunks_$TAG.jsonl
: > "$OUT"
for R in $RS; do
  echo "{\"chunk_reads\": $R}" >> "$OUT"
  COLBWT_LINE_ROWS_CHUNK=$R python3 "$REPO/tools/ab_bench.py" --reps 3 "$@" "$REPO/col-bwt_amd/libcolbwt.so@3" "$REPO/col-bwt_amd/libcolbwt.so@4" >> "$OUT" 2>> "$OUT.err" || { tail -5 "$OUT.err"; exit 1; }
done
cat "$OUT"
