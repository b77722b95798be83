#!/bin/bash
# counter passes over the joint-calling stream's EM step kernels (gpurun, repository root)
root=$PWD; out=$root/gpurun_out/em_pmc; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for o in full sym; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $out/${o}_$i -o p -- python3 $root/bench.py --workload joint --steps 1 --warmup 0 --no-stream --no-cpu --no-e2e --option em_step=$o > /dev/null 2> $out/${o}_$i.err || exit 3
  done
  python3 $root/tools/pmc_summarize.py $out/$o.csv $out/${o}_1 $out/${o}_2 $out/${o}_3 $out/${o}_4 $out/${o}_5 || exit 4
  grep -E "^kernel|k_em_step" $out/$o.csv
done
