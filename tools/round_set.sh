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
# the BGZF inflater: timing against the host library, kernel statistics and counters of the same program
timeout -k 10 200 python3 tools/time_inflate.py 2000 > "$out/inflate_time.txt" 2>&1 || { echo "time_inflate failed"; exit 6; }
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_inflate" -o p -- python3 "$root/tools/time_inflate.py" 2000 > /dev/null 2> "$out/stats_inflate.err" || { echo "inflate stats failed"; exit 7; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$out/pmc_inflate" -o p -- python3 "$root/tools/time_inflate.py" 2000 > /dev/null 2> "$out/pmc_inflate.err" || { echo "inflate pmc failed"; exit 8; }
cd "$root"
find "$out/stats_inflate" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats_inflate.csv" \;
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(out + "/pmc_inflate/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_bgzf_inflate" in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
with open(out + "/pmc_inflate.csv", "w") as f:
    f.write("kernel,counter,mean_per_dispatch,dispatches\n")
    for k, (v, n) in sorted(acc.items()):
        f.write("k_bgzf_inflate,%s,%.1f,%d\n" % (k, v / max(n, 1), n))
PY
tail -4 "$out/valu_insts.txt"; head -c 1500 "$out/bench.json"; echo; head -12 "$out/kernel_stats.csv"
