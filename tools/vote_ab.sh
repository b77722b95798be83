#!/bin/bash
# anchoring kernel A/B on the three pair streams: position table in LDS against bit-parallel counts
set -e
mkdir -p gpurun_out
for wl in wgs panel joint; do
  for k in table bits; do
    timeout -k 10 200 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu --no-e2e --no-stream --option vote_kernel=$k > gpurun_out/vote_ab_${wl}_${k}.json
    python - <<PY
import json
d = json.loads(open("gpurun_out/vote_ab_${wl}_${k}.json").read().strip().splitlines()[-1])
print("${wl} ${k}: anchor %.3f ms (%d named + %d everywhere of %d pairs), align %.3f ms, step %.3f ms, %.0f GCUPS" % (d["roofline_anchor"]["kernel_ms"], d["roofline_anchor"]["pairs_counted_over_named_positions"], d["roofline_anchor"]["pairs_counted_over_every_position"], d["roofline_anchor"]["pairs"], d["roofline"]["kernel_ms"], d["ms_per_step"], d["value"]))
PY
  done
done
