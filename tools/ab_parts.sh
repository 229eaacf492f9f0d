#!/bin/bash
# Where a trip's time goes: the line-row kernel with its output flush, and with flush and collector
# push, compiled out (results are wrong; only the times mean something).  Build the variants first:
#   make -C col-bwt_amd variant TAG=noflush VFLAGS=-DCOLBWT_EXP_NO_FLUSH
#   make -C col-bwt_amd variant TAG=nopush VFLAGS="-DCOLBWT_EXP_NO_FLUSH -DCOLBWT_EXP_NO_PUSH"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ab_parts_${1:-a}.jsonl
: > "$OUT"
for lib in ${LIBS:-libcolbwt.so libcolbwt_noflush.so libcolbwt_nopush.so}; do
  timeout -k 10 200 python3 "$REPO/tools/ab_bench.py" --reps 5 "$REPO/col-bwt_amd/$lib@4:8" >> "$OUT" 2>> "$OUT.err" || { tail -5 "$OUT.err"; exit 1; }
done
cat "$OUT"
