#!/bin/bash
# the pass in panel order (TbFlushArgs::panel): headline leg per panel width and workgroup count
set -u
for wgs in 20480 10240; do for panel in 0 24 40 56 80; do
LPX_TB_PANEL=$panel LPX_TB_WGS=$wgs LPX_BENCH_SUSTAINED_S=0.6 timeout -k 10 200 python3 bench.py --headline-only --steps 40 --warmup 5 --cpu-pivots 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wgs', $wgs, 'panel', $panel, 'ms/step %.4f' % d['ms_per_step'], 'pivots/s %.0f' % d['value'], 'pass ms %.4f' % d['roofline']['avg_launch_ms'], 'frac %.3f' % d['roofline']['frac'], 'sustained %.0f' % d['sustained']['value'])
" || exit 1
done; done
echo "--- in place, serial (the pass alone on the chip)"
for panel in 0 40 80; do
LPX_TB_PIPE_SERIAL=1 LPX_TB_PANEL=$panel LPX_BENCH_SUSTAINED_S=0 timeout -k 10 200 python3 bench.py --headline-only --steps 20 --warmup 5 --cpu-pivots 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('serial panel', $panel, 'ms/step %.4f' % d['ms_per_step'], 'pass ms %.4f' % d['roofline']['avg_launch_ms'], 'frac %.3f' % d['roofline']['frac'])
" || exit 1
done
