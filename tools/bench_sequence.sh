#!/bin/bash
# consecutive bench.py processes on one box: does the kernel time depend on where the table's memory comes from?
O=gpurun_out/${1:-r03w_bench_sequence}.jsonl
: > $O
run() { echo "{\"env\": \"$1\"}" >> $O; env $1 timeout -k 10 200 python bench.py --no-cpu --steps 5 --warmup 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(json.dumps({'ms_per_step': round(d['ms_per_step'], 3), 'index_load_s': d['config']['index_load_s']}))
" >> $O; }
run COLBWT_VMM=1; run COLBWT_VMM=1; run COLBWT_VMM=1
run COLBWT_VMM=0; run COLBWT_VMM=0; run COLBWT_VMM=0
run COLBWT_VMM=1; run COLBWT_VMM=1
run COLBWT_VMM=0; run COLBWT_VMM=1; run COLBWT_VMM=0; run COLBWT_VMM=1
cat $O
