#!/bin/bash
# kernel trace of the driver's window (20 steps after 5 warm-up steps) -> timeline of each call's passes
# bash tools/window_trace.sh <tag> ["ENV=1 ..."]
set -u
TAG=$1; shift
export TMPDIR=/tmp
env ${1:-LPX_NONE=0} LPX_BENCH_SUSTAINED_S=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/wtrace_$TAG -- python3 bench.py --headline-only --steps 20 --warmup 5 --cpu-pivots 0 > gpurun_out/wtrace_$TAG.log 2>&1
python tools/window_timeline.py gpurun_out/wtrace_$TAG > gpurun_out/window_$TAG.txt 2>&1
rm -rf gpurun_out/wtrace_$TAG
cat gpurun_out/window_$TAG.txt
