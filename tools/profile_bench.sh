#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the default bench, then PMC passes for K1's HBM traffic.
# Usage: bash tools/profile_bench.sh <round-tag>
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 50 --warmup 5 --cpu-pivots 0 > $OUT/bench_trace.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 20 --warmup 2 --cpu-pivots 0 > $OUT/bench_fetch.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 20 --warmup 2 --cpu-pivots 0 > $OUT/bench_write.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -20
