#!/bin/bash
# kernel timeline of the training step (steady-state half of the run): how much of the step has no kernel in flight
out=gpurun_out/tl_train
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/$out/trace -- python3 $R/bench.py --mode train --steps 12 --warmup 3 > $R/$out/run.log 2>&1
cd $R
python3 profiles/timeline.py $out/trace gpurun_out/timeline_train.txt k_tola 0.6
rm -rf $out/trace
