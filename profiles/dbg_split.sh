#!/bin/bash
# where a k_conv_p launch spends its time: timing-only variants (results wrong) with one phase removed
mkdir -p gpurun_out/split
for d in 0 1 2 4 3 7; do
  SE_CONVP_DBG=$d SE_PIPELINE=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/split/dbg_$d.json 2> gpurun_out/split/dbg_$d.err
  echo "dbg $d done"
done
