#!/bin/bash
# Little's-law curve of the memory system for the query kernel's access shape (GPU box):
# dependent random loads from one 128-byte line of a large table, with 1/8 .. 8/8 of the
# resident lanes (one generation, 1500 steps each).
#   usage: tools/gather_littles.sh <tag> [table_MiB=16384] [mode=3]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/littles_${1:-a}.jsonl
MIB=${2:-16384}; MODE=${3:-3}
: > "$OUT"
for lanes in 32768 65536 131072 262144 524288; do
  "$REPO/tools/gather_bench" $MIB $lanes 1500 $MODE 2 0 >> "$OUT" || exit 1
done
cat "$OUT"
