#!/bin/bash
# the headline window at several lengths, with and without the HIP events around the pass: time = a + b K ?
set -u
for ev in 1 0; do for K in 20 40 80 160; do
LPX_BENCH_K1_EVENTS=$ev LPX_BENCH_SUSTAINED_S=0.5 timeout -k 10 200 python3 bench.py --headline-only --steps $K --warmup 5 --cpu-pivots 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('events', $ev, 'K', $K, 'ms total %.3f' % (d['ms_per_step']*d['steps']), 'ms/step %.4f' % d['ms_per_step'], 'pivots/s %.0f' % d['value'], 'pass ms %.4f' % d['roofline']['avg_launch_ms'], 'sustained %.0f' % d['sustained']['value'])
" || exit 1
done; done
