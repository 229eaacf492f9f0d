#!/bin/bash
# Where the launch time goes on the way out: the mismatch-line kernel with parts of its output path
# compiled out (results are WRONG; only the times mean something).  Build the variants first:
#   make -C col-bwt_amd variant TAG=noflush  VFLAGS=-DCOLBWT_ABL_NO_FLUSH      # no flush at all (lanes still wait for the flush trip)
#   make -C col-bwt_amd variant TAG=nostores VFLAGS=-DCOLBWT_ABL_NO_STORES     # the flush without its store instructions
# (profiles/r03h_*, r03l_*: also with the stores aimed at a cached kilobyte and with the pushes compiled out)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ab_parts_${1:-a}.jsonl
: > "$OUT"
for lib in ${LIBS:-libcolbwt.so libcolbwt_nostores.so libcolbwt_noflush.so libcolbwt.so}; do
  timeout -k 10 200 python3 "$REPO/tools/ab_bench.py" --reps 5 "$REPO/col-bwt_amd/$lib@5" >> "$OUT" 2>> "$OUT.err" || { tail -5 "$OUT.err"; exit 1; }
done
cat "$OUT"
