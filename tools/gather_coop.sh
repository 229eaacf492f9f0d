#!/bin/bash
# Lane-cooperative row fetches against the per-lane shapes (GPU box): does the texture
# addresser charge per lane or per distinct row of an instruction?
#   usage: tools/gather_coop.sh <tag> [table_MiB=16384]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/coop_${1:-a}.jsonl
MIB=${2:-16384}
: > "$OUT"
for spec in "0 524288" "10 524288" "3 524288" "2 524288" "11 524288" "11 1048576" "12 524288" "13 524288" "14 524288" "14 2097152" "12 262144" "13 262144"; do
  set -- $spec
  "$REPO/tools/gather_bench" $MIB $2 1500 $1 2 0 >> "$OUT" || exit 1
done
cat "$OUT"
