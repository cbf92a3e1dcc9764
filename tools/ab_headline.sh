#!/bin/bash
# A/B of an environment switch on the headline leg, alternating in one session on one box (boxes differ by ~1 %).
# Usage (GPU box): bash tools/ab_headline.sh LPX_RANK1_HOIST 0 1
VAR=$1; A=$2; B=$3
for rep in 1 2 3; do
  for v in $A $B; do
    env $VAR=$v timeout -k 10 200 python3 bench.py --headline-only --steps 100 --warmup 10 --cpu-pivots 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', round(d['value'],1), 'pivots/s  K1', round(d['roofline']['avg_launch_ms'],4), 'ms  frac', round(d['roofline']['frac'],4))"
  done
done
