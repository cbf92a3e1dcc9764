#!/bin/bash
# kernel-trace stats of the secondary workloads (GPU box, via gpurun): bash tools/profile_extra.sh <tag>
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for W in revised10k revised1k tableau1k batch; do
  LPX_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -- python3 bench.py --workload $W --steps 100 --warmup 5 --cpu-pivots 0 > $OUT/bench_$W.log 2>&1 || exit 1
done
