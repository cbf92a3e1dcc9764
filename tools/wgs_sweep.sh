#!/bin/bash
# workgroups of the pass (band height): time per pass and step per LPX_TB_WGS
set -u
for wgs in 20480 10240 7168 5120 3584 2560; do
LPX_TB_WGS=$wgs LPX_BENCH_EAGER=0 LPX_BENCH_SUSTAINED_S=0.8 timeout -k 10 200 python3 bench.py --headline-only --steps 40 --warmup 5 --cpu-pivots 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wgs', $wgs, 'ms/step %.4f' % d['ms_per_step'], 'pivots/s %.0f' % d['value'], 'pass ms %.4f' % d['roofline']['avg_launch_ms'], 'frac %.3f' % d['roofline']['frac'], 'sustained %.0f' % d['sustained']['value'])
" || exit 1
done
