#!/bin/bash
# copies the summaries of gpurun_out/<tag> (tools/round_set.sh) into profiles/ under the round's names
set -eu
tag=${1:-r02}; src=gpurun_out/$tag; dst=profiles
cp $src/bench.json $dst/${tag}_bench.json
cp $src/bench_panel.json $dst/${tag}_bench_panel.json
cp $src/bench_joint.json $dst/${tag}_bench_joint.json
cp $src/kernel_stats.csv $dst/${tag}_kernel_stats.csv
cp $src/kernel_stats_joint.csv $dst/${tag}_kernel_stats_joint.csv
cp $src/pmc_summary.csv $dst/${tag}_pmc_summary.csv
cp $src/vote/pmc.csv $dst/${tag}_pmc_vote_panel.csv
{ echo "read_len,kernel,counter,mean_per_dispatch,dispatches"; for L in 100 150 250; do grep "k_align2," $src/valu_L$L.csv | sed "s/^/$L,/"; done; } > $dst/${tag}_valu_insts.csv
cp $src/inflate_time.txt $dst/${tag}_inflate_time.txt
cp $src/kernel_stats_inflate.csv $dst/${tag}_kernel_stats_inflate.csv
cp $src/pmc_inflate.csv $dst/${tag}_pmc_inflate.csv
ls -la $dst | grep ${tag}_
