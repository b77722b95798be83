#!/bin/bash
# VALU instructions one wavefront of k_align2 executes, for three read lengths (gpurun, repository root):
# SQ_INSTS_VALU / SQ_WAVES per dispatch.  usage: tools/valu_count.sh r02
set -u
tag=${1:-r02}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for L in 100 150 250; do
  timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d "$out/pmc_valu_$L" -o p -- python3 "$root/bench.py" --no-cpu --no-e2e --no-stream --steps 2 --warmup 0 --read-len $L > /dev/null 2> "$out/pmc_valu_$L.err" || exit 3
  python3 "$root/tools/pmc_summarize.py" "$out/valu_L$L.csv" "$out/pmc_valu_$L" || exit 4
  grep "k_align2," "$out/valu_L$L.csv"
done
