#!/bin/bash
# 128-byte rows fetched lane-cooperatively through LDS (modes 15 / 16) against one 16-byte load
# per lane (mode 0), by table size and by resident waves (GPU box).
#   usage: tools/gather_fat.sh <tag>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/fat_${1:-a}.jsonl
: > "$OUT"
for spec in "16384 15 196608" "16384 16 196608" "16384 15 262144" "16384 16 262144" "16384 16 131072" "16384 0 524288" \
            "65536 15 196608" "65536 16 196608" "65536 0 524288" "65536 13 524288" \
            "131072 16 196608" "131072 15 196608" "131072 0 524288" "4096 16 196608" "4096 0 524288"; do
  set -- $spec
  "$REPO/tools/gather_bench" $1 $3 1500 $2 2 0 >> "$OUT" || exit 1
done
cat "$OUT"
