#!/bin/bash
# headline leg under sets of environment knobs, one per line of the argument list: "A=1 B=2" ...
set -u
OUT=gpurun_out/r03_pipe_var.log
: > $OUT
for V in "$@"; do
  echo "== $V" >> $OUT
  env $V timeout -k 10 120 python bench.py --headline-only --cpu-pivots 100 2>>$OUT.err | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print(round(r['value']), 'pivots/s; ms_per_step', round(r['ms_per_step'],3), 'pass ms', round(r['roofline']['avg_launch_ms'],3), 'frac', round(r['roofline']['frac'],3), 'share', round(r['roofline']['share_of_step'],3))
" >> $OUT || exit 1
done
cat $OUT
