#!/bin/bash
# rocprofv3 kernel stats of one bench configuration, serial stage order.  usage: bash profiles/stats_only.sh <tag> [bench args]
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SE_PIPELINE=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $R/$out/stats.log 2>&1
cd $R
python3 profiles/summarize.py stats $out/stats $out/kernel_stats.csv
rm -rf $out/stats
head -16 $out/kernel_stats.csv | cut -c1-150
