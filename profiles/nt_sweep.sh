#!/bin/bash
# per-layer tiling sweep of k_conv_p: one bench run per forced NT (layers without that candidate keep their default)
mkdir -p gpurun_out/sweep
SE_CONVP_VERBOSE=1 SE_PIPELINE=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/sweep/nt_default.json 2> gpurun_out/sweep/nt_default.err
for nt in 1 2 3 4 6 8 10 12; do
  SE_CONVP_NT=$nt SE_PIPELINE=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/sweep/nt_$nt.json 2> gpurun_out/sweep/nt_$nt.err
  echo "nt $nt done"
done
