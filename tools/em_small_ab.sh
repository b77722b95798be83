#!/bin/bash
# small-window EM: the reference-order kernel (default) against the regrouped one, per shape (windows haplotypes reads individuals)
for shape in "32768 4 32 1" "32768 2 32 1" "8192 3 32 1" "4000 2 100 50" "1000 4 100 50" "2000 8 100 50" "4000 5 100 50" "2000 6 60 12" "8192 8 40 1"; do
  for opt in "em_small=exact" "em_small=fast"; do
    echo -n "$opt: "; timeout -k 5 60 python3 tools/em_shape.py $shape $opt || exit 1
  done
done
