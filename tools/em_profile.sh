#!/bin/bash
# per-kernel times of the joint-calling stream's caller stage, for each EM step kernel
root=$PWD; cd /tmp && export TMPDIR=/tmp
for o in full sym; do
  rm -rf $root/gpurun_out/em_prof_$o
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/em_prof_$o -o t -- python3 $root/bench.py --workload joint --steps 3 --warmup 1 --no-stream --no-cpu --no-e2e --option em_step=$o > /dev/null 2>&1 || exit 1
  f=$(find $root/gpurun_out/em_prof_$o -name '*kernel_stats.csv' | head -1)
  echo "== $o"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %6s total_us %10.1f avg_us %9.2f" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3))
PY
done
