#!/bin/bash
# rocprofv3 passes of the default bench (single stream so that kernels do not overlap): kernel stats + PMC groups.
# usage: bash profiles/pmc_run.sh <tag> [extra bench args]   (the secondary legs of the default run are profiled by their own invocations:
#   WORKLOAD_KEY=fullsubnet/b256/f32 bash profiles/pmc_run.sh r03_fsn_b256 --model fullsubnet;  ... --model student --batch 1024 --nfft 400 --dtype bf16x3;  ... --mode train)
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SE_PIPELINE=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $R/$out/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$out/pmc_mfma -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $R/$out/pmc_mfma.log 2>&1
echo "pmc mfma done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $R/$out/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $R/$out/pmc_write.log 2>&1
echo "pmc write done"
cd $R
python3 profiles/summarize.py stats $out/stats $out/kernel_stats.csv
python3 profiles/summarize.py pmc $out/pmc_mfma $out/pmc_mfma.csv
python3 profiles/summarize.py pmc $out/pmc_fetch $out/pmc_write $out/pmc_hbm.csv
python3 profiles/summarize.py hbm-json $out/pmc_hbm.csv "${WORKLOAD_KEY:-crn/b256/nfft512/f32}" "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of profiles/pmc_run.sh $tag (serial stage order)" $out/pmc_hbm_current.json
rm -rf $out/stats $out/pmc_mfma $out/pmc_fetch $out/pmc_write
ls -la $out
