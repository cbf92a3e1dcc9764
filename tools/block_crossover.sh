#!/bin/bash
# Where does the deferred-update chain overtake one update per pivot?  Run on the GPU box: bash tools/block_crossover.sh
# (tableau1k workload at other sizes: steps are single pivots there; LPX_TABLEAU_BLOCK_MIN_MB=0 forces the chain on)
for mn in "1000 2000" "1500 3000" "2000 4000" "3000 6000" "5000 10000" "8000 16000"; do
  set -- $mn
  for mode in off 8 16 32; do
    if [ $mode = off ]; then export LPX_TABLEAU_BLOCK=0; else export LPX_TABLEAU_BLOCK=$mode LPX_TABLEAU_BLOCK_MIN_MB=0; fi
    v=$(timeout -k 10 120 python3 bench.py --workload tableau1k --m $1 --n $2 --steps 1024 --warmup 64 --cpu-pivots 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']))")
    echo "m=$1 n=$2 block=$mode: $v pivots/s"
  done
done
