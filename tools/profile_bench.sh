#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the default bench (all three legs), then PMC passes for K1's HBM
# traffic on the headline leg alone, then a kernel trace of config 2 (eager launches: one k_t_fused per pivot).
# Usage: bash tools/profile_bench.sh <round-tag>
# (rocprofv3 may crash in its own finalizer after the result files are written: a step counts as done when its CSV exists)
set -u
TAG=${1:-r03}
need() { ls $1 > /dev/null 2>&1 || { echo "missing: $1"; exit 1; }; }
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 --cpu-pivots 0 > $OUT/bench_trace.log 2>&1; need "$OUT/trace/*/*_kernel_stats.csv"
LPX_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_tableau1k -- python3 bench.py --workload tableau1k --steps 2000 --warmup 50 --cpu-pivots 0 > $OUT/bench_tableau1k_trace.log 2>&1; need "$OUT/trace_tableau1k/*/*_kernel_stats.csv"
timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --workload tableau1k --steps 4000 --warmup 100 > $OUT/bench_tableau1k.json 2>> $OUT/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --workload batch --steps 40 --warmup 5 > $OUT/bench_batch.json 2>> $OUT/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --workload config5 --steps 3 --warmup 1 > $OUT/bench_config5.json 2>> $OUT/bench_default.err || exit 1
# PMC passes last: the profiler runs one kernel at a time, so the chain runs in its serial diagnostic mode (same kernels,
# same two streams, nothing side by side); each pass under a short limit of its own, nothing follows a pass that was killed
LPX_TB_PIPE_SERIAL=1 timeout -s KILL 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --headline-only --steps 10 --warmup 2 --cpu-pivots 0 > $OUT/bench_fetch.log 2>&1; [ $? -eq 137 ] && { echo 'PMC pass killed at its limit'; exit 1; }; need "$OUT/pmc_fetch/*/*_counter_collection.csv"
LPX_TB_PIPE_SERIAL=1 timeout -s KILL 100 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --headline-only --steps 10 --warmup 2 --cpu-pivots 0 > $OUT/bench_write.log 2>&1; [ $? -eq 137 ] && { echo 'PMC pass killed at its limit'; exit 1; }; need "$OUT/pmc_write/*/*_counter_collection.csv"
find $OUT -name "*kernel_stats.csv" | head
