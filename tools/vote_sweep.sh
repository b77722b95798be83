#!/bin/bash
for w in 1 2 4; do
  timeout -k 10 200 python bench.py --no-cpu --no-e2e --no-stream --steps 10 --option vote_waves=$w 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('vote_waves $w anchor_ms %.4f kernel_ms %.4f step %.4f' % (r['anchor_kernel_ms'], r['kernel_ms'], d['ms_per_step']))" || exit 1
done
