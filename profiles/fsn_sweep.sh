#!/bin/bash
# usage: fsn_sweep.sh "<knob values>"  -> frames/s per (SE_FSN_BIG, dtype)
for k in $1; do for d in f32 bf16x3; do
  SE_FSN_BIG=$k timeout -k 10 300 python bench.py --model fullsubnet --dtype $d --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('BIG=$k', '$d', round(j['value'],1), round(j['ms_per_step'],1))" || exit 1
done; done
