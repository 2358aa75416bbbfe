#!/bin/bash
# Copy what tools/collect_profiles.sh left under gpurun_out/prof_<tag>/ into the tracked profiles/ directory, named per round:
#     bash tools/publish_profiles.sh <tag> <round prefix, e.g. r03>
set -e
src=gpurun_out/prof_${1:?tag}; pre=profiles/${2:?prefix}
for k in overlapped serial c3 c4 oo_overlapped hmc; do
  [ -f $src/$k/run_kernel_stats.csv ] && cp $src/$k/run_kernel_stats.csv ${pre}_${k}_kernel_stats.csv
done
for k in overlapped serial c3 c4 oo_overlapped hmc c4_full c4_full_in_turn final final_default_stream; do
  [ -f $src/bench_$k.log ] && grep -h '^{' $src/bench_$k.log | tail -1 > ${pre}_${k}_bench.json
done
for k in headline c3 c4 c4full; do [ -f $src/pmc_summary_$k.json ] && cp $src/pmc_summary_$k.json ${pre}_pmc_summary_$k.json; done
[ -f $src/pmc_overlap_check.json ] && cp $src/pmc_overlap_check.json ${pre}_overlap_pmc.json
[ -f $src/hmc_timeline.txt ] && cp $src/hmc_timeline.txt ${pre}_hmc_timeline.txt
ls -la ${pre}_*
