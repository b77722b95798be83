#!/bin/bash
# The round's whole measurement set (gpurun, repository root): VALU instruction counts of the DP kernel at three read
# lengths, the default bench line with kernel statistics and counter passes, the panel and joint bench lines, the
# anchoring kernel's counters on the panel stream, per-kernel statistics of the joint stream.  usage: tools/round_set.sh r02
set -u
tag=${1:-r02}
root=$PWD; out=$root/gpurun_out/$tag; mkdir -p "$out"
tools/valu_count.sh $tag > "$out/valu_insts.txt" 2>&1 || { echo "valu_count failed"; exit 1; }
tools/profile_round.sh $tag > "$out/profile_round.txt" 2>&1 || { echo "profile_round failed"; exit 2; }
for w in panel joint; do
  timeout -k 10 300 python3 bench.py --workload $w --no-e2e --no-stream > "$out/bench_$w.json" 2> "$out/bench_$w.err" || { echo "bench $w failed"; exit 3; }
done
tools/pmc_vote.sh panel $tag/vote > "$out/vote_pmc.txt" 2>&1 || { echo "pmc_vote failed"; exit 4; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_joint" -o p -- python3 "$root/bench.py" --workload joint --no-cpu --no-e2e --no-stream > /dev/null 2> "$out/stats_joint.err" || { echo "joint stats failed"; exit 5; }
cd "$root"
find "$out/stats_joint" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats_joint.csv" \;
tail -4 "$out/valu_insts.txt"; head -c 1500 "$out/bench.json"; echo; head -12 "$out/kernel_stats.csv"
