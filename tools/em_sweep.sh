#!/bin/bash
# caller stage of the joint-calling stream (50 individuals x 257 haplotypes) under each step kernel / workgroup size
for o in em_step=full em_step_waves=4 em_step_waves=8 em_step_waves=16; do
  timeout -k 10 200 python bench.py --workload joint --steps 10 --warmup 3 --no-stream --no-cpu --no-e2e --option $o 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$o', d['caller_stage'])" || exit 1
done
