#!/bin/bash
# VALU / LDS / wait counters of one bench configuration (for kernels that are not MFMA-bound).  usage: bash profiles/pmc_kernel.sh <tag> [bench args]
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SE_PIPELINE=0
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/$out/pmc_valu -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $R/$out/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $R/$out/pmc_mem -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $R/$out/pmc_mem.log 2>&1
cd $R
python3 profiles/summarize.py pmc $out/pmc_valu $out/pmc_mem $out/pmc_kernel.csv
rm -rf $out/pmc_valu $out/pmc_mem
grep -E "preconv|conv_small|gln_ew" $out/pmc_kernel.csv | cut -c1-160
