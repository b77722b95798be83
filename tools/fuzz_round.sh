#!/bin/bash
# The randomised checks on the tree's kernels, each for a time budget (seconds, default 90), logs under gpurun_out/<tag>:
# alignment parity, caller stage, whole-region VCF text, file-to-file VCF text, pipeline against planted truth, device inflater
# against zlib.
# usage: tools/fuzz_round.sh [tag] [seconds] [first seed] [checks ...]
tag=${1:-r02}; secs=${2:-90}; seed=${3:-5000}
shift 3 2>/dev/null
checks=${@:-parity caller vcf files pipeline inflate}
out=gpurun_out/$tag; mkdir -p $out
for f in $checks; do
  timeout -k 10 $((secs + 120)) python3 -u tests/fuzz_$f.py $secs $seed > $out/fuzz_$f.txt 2>&1
  rc=$?
  echo "fuzz_$f rc=$rc"; tail -2 $out/fuzz_$f.txt
  if [ $rc -ne 0 ]; then exit $rc; fi          # nothing more on the GPU after a check that failed or was killed
done
