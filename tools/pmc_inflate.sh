#!/bin/bash
# counters of the inflate kernel on a simulated BAM stretch (gpurun, repository root). usage: tools/pmc_inflate.sh tag
set -u
tag=${1:-pmc_inflate}
root=$PWD; out=$root/gpurun_out/$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d "$out/p1" -o p -- python3 "$root/tools/time_inflate.py" 2000 > /dev/null 2> "$out/p1.err" || exit 3
timeout -s KILL 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES --output-format csv -d "$out/p2" -o p -- python3 "$root/tools/time_inflate.py" 2000 > /dev/null 2> "$out/p2.err" || exit 4
python3 "$root/tools/pmc_summarize.py" "$out/pmc.csv" "$out/p1" "$out/p2" || exit 5
grep "inflate" "$out/pmc.csv"
