#!/bin/bash
# K1 with tall persistent-style tiles (rows/block = rows / ~24) on the revised10k workload: bash tools/k1_tall.sh
set -u
for cfg in "1 1" "417 4" "417 8" "417 2" "500 4" "313 4" "209 4"; do
  set -- $cfg
  echo "rows/block=$1 unroll=$2"
  LPX_RANK1_ROWS=$1 LPX_RANK1_UNROLL=$2 LPX_RANK1_MINWG=1 timeout -k 10 200 python3 -u bench.py --workload revised10k --steps 300 --warmup 20 --cpu-pivots 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline'])" || exit 1
done
