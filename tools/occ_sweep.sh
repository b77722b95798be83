#!/bin/bash
# time the alignment kernel at reduced occupancy (extra LDS per workgroup)
for pad in 0 50000 78000 150000; do
  timeout -k 10 200 python bench.py --no-cpu --no-e2e --no-stream --steps 10 --option align_lds_pad=$pad 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('lds_pad $pad kernel_ms %.4f' % r['kernel_ms'])" || exit 1
done
