#!/bin/bash
# L2 hit/miss pass (separate from the MFMA / HBM passes of pmc_run.sh).  usage: bash profiles/pmc_cache.sh <tag> [bench args]
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SE_PIPELINE=0
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/$out/pmc_l2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $R/$out/pmc_l2.log 2>&1
echo "pmc l2 done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$out/pmc_mfma -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $R/$out/pmc_mfma.log 2>&1
echo "pmc mfma done"
cd $R
python3 profiles/summarize.py pmc $out/pmc_l2 $out/pmc_l2.csv
python3 profiles/summarize.py pmc $out/pmc_mfma $out/pmc_mfma.csv
rm -rf $out/pmc_l2 $out/pmc_mfma
cat $out/pmc_l2.csv
