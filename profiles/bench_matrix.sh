#!/bin/bash
# every bench line quoted in DESIGN.md, one JSON line per run under gpurun_out/matrix/ (run from the repo root on the GPU box)
out=gpurun_out/matrix
mkdir -p $out
run() { name=$1; shift; timeout -k 10 600 python bench.py "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(tail -c 300 $out/$name.json | head -c 0)"; python - <<PY
import json
try:
    d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1])
    print("   ", d["metric"], round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],2), "roofline", d["roofline"]["kernel"] if d.get("roofline") else None, round(d["roofline"]["frac"],3) if d.get("roofline") else None)
except Exception as ex:
    print("    ERR", ex)
PY
}
run crn_f32_default
run crn_bf16x3 --dtype bf16x3 --no-cpu-baseline
run crn_f16 --dtype f16 --no-cpu-baseline
run crn_nfft400 --nfft 400 --no-cpu-baseline
run crn_b1024 --batch 1024 --no-cpu-baseline
run crn_b1 --batch 1 --steps 20 --warmup 3 --no-cpu-baseline
run student_b1024_f32 --model student --batch 1024 --no-cpu-baseline
run student_b1024_bf16x3 --model student --batch 1024 --dtype bf16x3 --no-cpu-baseline
run student_b1024_f16 --model student --batch 1024 --dtype f16 --no-cpu-baseline
run crn_elu --model crn_elu --no-cpu-baseline
run fullsubnet --model fullsubnet --no-cpu-baseline
run train_hip --mode train --no-cpu-baseline
run train_hip_accum1 --mode train --accum 1 --no-cpu-baseline
run crn_elu_bf16x3 --model crn_elu --dtype bf16x3 --no-cpu-baseline
