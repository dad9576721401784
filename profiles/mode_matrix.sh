#!/bin/bash
# headline workload, f32 / bf16x3 x pipelined / serial stage order: frames/s and ms per step
for d in f32 bf16x3; do for p in 1 0; do
  SE_PIPELINE=$p timeout -k 10 300 python bench.py --dtype $d --no-cpu-baseline --no-secondary --steps 5 --warmup 2 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$d pipeline=$p', round(j['value']), round(j['ms_per_step'],2))" || exit 1
done; done
