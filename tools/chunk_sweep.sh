#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/chunk_sweep_final.jsonl
: > "$OUT"
for C in "8,100" "8,50" "8,200" "6,100" "12,100" "16,50" "4,100" "8,100"; do
  echo "{\"chunk\": \"$C\"}" >> "$OUT"
  COLBWT_LINE_ROWS_CHUNK=$C timeout -k 10 200 python3 "$REPO/tools/ab_bench.py" --reps 5 "$REPO/col-bwt_amd/libcolbwt.so@4:8" >> "$OUT" 2>> "$OUT.err" || { tail -5 "$OUT.err"; exit 1; }
done
cut -c1-120 "$OUT"
