#!/bin/bash
# Little's-law curve of the memory system for the query kernel's access shape (GPU box):
# dependent random 2 x 16-byte loads from one 128-byte line of a 16 GiB table, with
# 1/8 .. 8/8 of the resident lanes (one generation, 1500 steps each).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/littles_${1:-a}.jsonl
: > "$OUT"
for lanes in 32768 65536 131072 196608 262144 393216 524288 1048576; do
  "$REPO/tools/gather_bench" 16384 $lanes 1500 3 2 0 >> "$OUT" || exit 1
done
cat "$OUT"
