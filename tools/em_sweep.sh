#!/bin/bash
# caller stage of the joint-calling stream (50 individuals x 257 haplotypes) under the options given, one run per argument
# (an argument is a comma-separated list of NAME=VALUE)
for o in "$@"; do
  opts=""; for kv in ${o//,/ }; do opts="$opts --option $kv"; done
  timeout -k 10 200 python bench.py --workload joint --steps 10 --warmup 3 --no-stream --no-cpu --no-e2e $opts 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['caller_stage']; print('%-40s setup %.3f ms  em %.3f ms  iters %.2f/%d' % ('$o', c['setup_ms'], c['em_ms'], c['em_iterations_mean'], c['em_iterations_max']))" || exit 1
done
