#!/bin/bash
# k_vote5: workgroup count sweep (each workgroup loops over haplotypes)
for wl in wgs panel; do
  for g in 512 1024 2048 4096 8192 1000000; do
    timeout -k 10 200 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu --no-e2e --no-stream --option vote_grid=$g > /tmp/vg.json || exit 1
    python - <<PY
import json
d = json.loads(open("/tmp/vg.json").read().strip().splitlines()[-1])
print("${wl} grid ${g}: anchor %.3f ms" % d["roofline_anchor"]["kernel_ms"])
PY
  done
done
