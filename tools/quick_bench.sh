#!/bin/bash
# wgs and panel bench lines without the CPU and file legs: value, step, DP kernel, anchoring kernel
for w in wgs panel; do
  timeout -k 10 200 python bench.py --no-cpu --no-e2e --no-stream --workload $w "$@" 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w: %.0f GCUPS  step %.4f ms  align %.4f ms  anchor %.4f ms' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['anchor_kernel_ms']))" || exit 1
done
