#!/bin/bash
# kernel timeline of FullSubNet streaming (steady-state part): usage: bash profiles/timeline_fsn.sh <tag> [bench args]
tag=$1; shift
out=gpurun_out/tl_fsn_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/$out/trace -- python3 $R/bench.py --model fullsubnet --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $R/$out/run.log 2>&1
cd $R
python3 profiles/timeline.py $out/trace gpurun_out/timeline_fsn_$tag.txt k_lstm_step_big 0.6
rm -rf $out/trace
