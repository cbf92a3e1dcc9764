#!/bin/bash
# kernel-trace stats + bench line of one workload (GPU box, via gpurun): bash tools/profile_one.sh <workload> <tag>
set -u
W=${1:-revised10k}
TAG=${2:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
LPX_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -- python3 bench.py --workload $W --steps 100 --warmup 5 --cpu-pivots 0 > $OUT/bench_$W.log 2>&1 || exit 1
timeout -k 10 300 python3 -u bench.py --workload $W --steps 500 --warmup 20 > $OUT/bench_$W.json 2> $OUT/bench_$W.err || exit 1
