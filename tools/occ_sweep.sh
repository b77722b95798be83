#!/bin/bash
# time the alignment kernel at reduced occupancy (extra LDS per workgroup)
for pad in 0 50000 78000 150000; do
  PG_ALIGN_LDS_PAD=$pad timeout -k 10 200 python bench.py --no-cpu --steps 10 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('lds_pad $pad kernel_ms %.4f' % r['kernel_ms'])" || exit 1
done
