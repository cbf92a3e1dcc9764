#!/bin/bash
# kernel trace of the headline leg under a set of environment knobs: bash tools/pipe_trace.sh <tag> "A=1 B=2"
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
env ${1:-LPX_NONE=0} timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$TAG -- python3 bench.py --headline-only --steps 6 --warmup 2 --cpu-pivots 0 > gpurun_out/trace_$TAG.log 2>&1
python tools/trace_overlap.py gpurun_out/trace_$TAG > gpurun_out/r03_overlap_$TAG.txt 2>&1
rm -rf gpurun_out/trace_$TAG
head -${2:-16} gpurun_out/r03_overlap_$TAG.txt
