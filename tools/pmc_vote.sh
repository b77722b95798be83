#!/bin/bash
# counters of the anchoring kernel on one workload (gpurun, repository root). usage: tools/pmc_vote.sh panel tag
set -u
w=${1:-panel}; tag=${2:-vote}
root=$PWD; out=$root/gpurun_out/$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d "$out/p1" -o p -- python3 "$root/bench.py" --no-cpu --no-e2e --no-stream --steps 3 --warmup 0 --workload $w > /dev/null 2> "$out/p1.err" || exit 3
timeout -s KILL 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$out/p2" -o p -- python3 "$root/bench.py" --no-cpu --no-e2e --no-stream --steps 3 --warmup 0 --workload $w > /dev/null 2> "$out/p2.err" || exit 4
python3 "$root/tools/pmc_summarize.py" "$out/pmc.csv" "$out/p1" "$out/p2" || exit 5
grep "k_vote" "$out/pmc.csv"
