#!/bin/bash
# sweep one environment knob over a bench workload: bash tools/sweep_env.sh <VAR> <workload> v1 v2 ...
set -u
VAR=$1; W=$2; shift 2
for v in "$@"; do
  export $VAR=$v
  echo -n "$VAR=$v  "
  timeout -k 10 200 python3 -u bench.py --workload $W --steps 500 --warmup 20 --cpu-pivots 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['unit'])" || exit 1
done
