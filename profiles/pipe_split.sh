#!/bin/bash
# what bounds the three-stream pipeline: timing-only runs (results wrong) with one stage's work removed.
# SE_DBG_SKIP only exists in a debug build: make -C speech_enhancement_mi_amd/csrc -B -j8 CXXFLAGS='-O3 -std=c++17 -fPIC -fno-slp-vectorize -DSE_DEBUG_KNOBS'
# (the default library refuses to start when the variable is set)
mkdir -p gpurun_out/pipe
for d in 0 1 2 3 4 8 12 15; do
  SE_DBG_SKIP=$d python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('skip=$d pipelined ms/step', round(d['ms_per_step'],2))" 
  SE_PIPELINE=0 SE_DBG_SKIP=$d python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('skip=$d serial    ms/step', round(d['ms_per_step'],2))" 
done
