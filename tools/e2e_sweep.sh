#!/bin/bash
# single-process file-to-file rate under a few helper-thread counts and lock hand-over intervals (gpurun, repository root)
export TF_REPEATS=16 TF_PAIRED=1 TF_READLEN=150 PLATYPUS_AMD_TIMINGS=1
for cfg in "6 0.005" "8 0.005" "12 0.005" "16 0.005" "12 0.0005" "12 0.02"; do
  set -- $cfg
  TF_ARGS="--readAhead $1" TF_SWITCH=$2 timeout -k 10 300 python tools/time_files.py 2000 30 > gpurun_out/e2e_sweep_$1_$2.txt 2>&1 || exit 1
  echo "readAhead $1 switch $2: $(grep '^rep' gpurun_out/e2e_sweep_$1_$2.txt | tr '\n' ' ')"
  grep "^worker" gpurun_out/e2e_sweep_$1_$2.txt | tail -n 1 | cut -c1-260
done
