#!/bin/bash
# K1 launch-plan knobs on the default bench workload: bash tools/sweep_k1_env.sh "VAR=val VAR2=val" ...
set -u
for cfg in "$@"; do
  echo -n "$cfg  "
  env $cfg timeout -k 10 200 python3 -u bench.py --steps 100 --warmup 5 --cpu-pivots 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['unit'], round(d['roofline']['achieved']), 'GB/s', round(d['roofline']['avg_launch_ms'],4), 'ms')" || exit 1
done
