#!/bin/bash
# per-layer tiling sweep of k_conv_p at the headline workload, pipelined stage order (the bench default): frames/s per forced NT
# usage: bash profiles/nt_layer_sweep.sh "<layer names>" "<NT values>" [bench args]
layers=$1; nts=$2; shift 2
SE_CONVP_VERBOSE=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" 2> gpurun_out/nt_verbose.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('default', round(j['value']))"
grep "^\[conv_p\]" gpurun_out/nt_verbose.err | sort -u | cut -c1-200
for l in $layers; do for nt in $nts; do
  env SE_CONVP_NT_$l=$nt timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$l NT=$nt', round(j['value']))" || exit 1
done; done
