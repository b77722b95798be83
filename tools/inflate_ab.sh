#!/bin/bash
# Device inflater: its GPU tests, its timing against the host library, and the file-to-file run with the BGZF blocks
# inflated on the GPU and by the host library (one process and eight).  Logs under gpurun_out/.
export TF_PAIRED=1 TF_READLEN=150 TF_REPEATS=16 TF_DUMP_AFTER=150
timeout -k 10 100 python -u -m pytest tests/test_gpu_inflate.py -x -q --timeout 60 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 150 python -u tools/time_inflate.py 2000 > gpurun_out/infl_time.log 2>&1 || exit 1
for mode in "device" "host" "device --workersPerGPU 8" "host --workersPerGPU 8"; do
  echo "inflate $mode"
  TF_ARGS="--inflate $mode" timeout -k 10 170 python -u tools/time_files.py 4000 30 > gpurun_out/files_ab.log 2>&1 || exit 1
  grep -E "^rep|Timeout" gpurun_out/files_ab.log
done
