#!/bin/bash
# A/B timing of library variants on one GPU, interleaved rounds in one call.
#   tools/ab_bench.sh <variant.so> ...   (paths relative to the repo root)
for round in 1 2; do
  for v in "$@"; do
    PLATGPU_LIB=$PWD/$v timeout -k 10 200 python bench.py --no-cpu --steps 10 --windows ${WINDOWS:-32768} --workload ${WORKLOAD:-wgs} 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', 'kernel_ms %.4f anchor_ms %.4f step_ms %.4f frac %.3f' % (r['kernel_ms'], r['anchor_kernel_ms'], d['ms_per_step'], r['frac']))" || exit 1
  done
done
