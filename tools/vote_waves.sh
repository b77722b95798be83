#!/bin/bash
# k_vote5: wavefronts per workgroup (64 pairs each)
for wl in wgs panel joint; do
  for w in 1 2 4; do
    timeout -k 10 200 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu --no-e2e --no-stream --option vote_waves=$w > /tmp/vw.json || exit 1
    python - <<PY
import json
d = json.loads(open("/tmp/vw.json").read().strip().splitlines()[-1])
print("${wl} waves ${w}: anchor %.3f ms" % d["roofline_anchor"]["kernel_ms"])
PY
  done
done
