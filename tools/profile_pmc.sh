#!/bin/bash
# PMC passes of the headline leg alone (HBM traffic of the pass): FETCH_SIZE and WRITE_SIZE in runs of their own, the chain in
# its serial diagnostic mode (the profiler runs one kernel at a time anyway), no sustained window.  bash tools/profile_pmc.sh <tag>
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
need() { ls $1 > /dev/null 2>&1 || { echo "missing: $1"; exit 1; }; }
for C in FETCH_SIZE:pmc_fetch WRITE_SIZE:pmc_write; do
  c=${C%%:*}; d=${C##*:}
  LPX_BENCH_SUSTAINED_S=0 LPX_TB_PIPE_SERIAL=1 timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $OUT/$d -- python3 bench.py --headline-only --steps 6 --warmup 2 --cpu-pivots 0 > $OUT/bench_$d.log 2>&1
  [ $? -eq 137 ] && { echo "PMC pass $c killed at its limit"; exit 1; }
  need "$OUT/$d/*/*_counter_collection.csv"
done
echo pmc done
