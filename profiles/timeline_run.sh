#!/bin/bash
# usage: bash profiles/timeline_run.sh <tag> [bench args]   -> gpurun_out/timeline_<tag>.txt (pipelined stage order, the bench default)
tag=$1; shift
out=gpurun_out/tl_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/$out/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $R/$out/run.log 2>&1
cd $R
python3 profiles/timeline.py $out/trace gpurun_out/timeline_$tag.txt
rm -rf $out/trace
