#!/bin/bash
# headline leg over the number of CUs reserved for the decision kernels of the pipelined chain (GPU box, via gpurun)
set -u
OUT=gpurun_out/r03_pipe_cus.log
: > $OUT
for R in ${@:-32 48 64 0}; do
  echo "== LPX_TB_PIPE_CUS=$R" >> $OUT
  LPX_TB_PIPE_CUS=$R timeout -k 10 120 python bench.py --headline-only --cpu-pivots 100 2>>$OUT.err | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print(round(r['value']), 'pivots/s; ms_per_step', round(r['ms_per_step'],3), 'pass ms', round(r['roofline']['avg_launch_ms'],3), 'frac', round(r['roofline']['frac'],3), 'share', round(r['roofline']['share_of_step'],3))
" >> $OUT || exit 1
done
cat $OUT
