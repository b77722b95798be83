#!/bin/bash
# Collects the round's measurement set on the GPU box (run through gpurun from the repository root):
#   bench line, rocprofv3 kernel statistics of the same command, three counter passes.
# usage: tools/profile_round.sh r01
set -u
tag=${1:-r01}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 400 python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o p -- python3 "$root/bench.py" --no-cpu --no-e2e --no-stream > "$out/stats_bench.json" 2> "$out/stats.err" || exit 2
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o p -- python3 "$root/bench.py" --no-cpu --no-e2e --no-stream --steps 4 --warmup 0 > /dev/null 2> "$out/pmc_fetch.err" || exit 3
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o p -- python3 "$root/bench.py" --no-cpu --no-e2e --no-stream --steps 4 --warmup 0 > /dev/null 2> "$out/pmc_write.err" || exit 4
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_sq" -o p -- python3 "$root/bench.py" --no-cpu --no-e2e --no-stream --steps 4 --warmup 0 > /dev/null 2> "$out/pmc_sq.err" || exit 5
cd "$root"
python3 tools/pmc_summarize.py "$out/pmc_summary.csv" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_sq" || exit 6
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
cat "$out/bench.json"
cat "$out/kernel_stats.csv"
