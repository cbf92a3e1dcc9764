#!/bin/bash
# GPU box (via gpurun): the single-launch tableau chain at config 2 over update-tile heights, then a kernel trace.
# Usage: bash tools/fused_sweep.sh <tag>
set -u
TAG=${1:-r02}
OUT=gpurun_out/fused_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for rows in ${ROWS:-0 1 2 4 8}; do
  LPX_FUSED_ROWS=$rows timeout -k 10 120 python3 bench.py --workload tableau1k --steps 4000 --warmup 50 --cpu-pivots 0 > $OUT/rows$rows.json 2> $OUT/rows$rows.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$OUT/rows$rows.json')); print('rows', $rows, round(d['value']), 'pivots/s', 'kernel ms', d['roofline']['avg_launch_ms'])"
done
LPX_GRAPH=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload tableau1k --steps 1000 --warmup 20 --cpu-pivots 0 > $OUT/trace.log 2>&1 || exit 1
find $OUT/trace -name "*kernel_stats.csv" -exec head -8 {} \;
