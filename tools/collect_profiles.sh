#!/bin/bash
# Run on the GPU box (through gpurun):  bash tools/collect_profiles.sh <tag>
# Produces under gpurun_out/prof_<tag>/: rocprofv3 --kernel-trace --stats summaries of bench.py with the two halves
# overlapped (the timed configuration) and serialised, the bench JSON lines, and one --pmc pass per counter group.
# python is named directly after "--" (the profiler preloads the GPU runtime: no env/bash hops).
set -o pipefail
tag=${1:-run}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B="bench.py --steps 3 --warmup 1 --cpu-samples 0 --no-host-io --no-other"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/overlapped -o run -- python $B > $out/bench_overlapped.log 2>&1 || exit 1
export FINROM_NO_OVERLAP=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/serial -o run -- python $B > $out/bench_serial.log 2>&1 || exit 1
B1="bench.py --steps 1 --warmup 1 --cpu-samples 0 --no-profile"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$out/pmc_$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --output-format csv -d $d -o run -- python $B1 > $d.log 2>&1 || exit 1
  echo "pmc $grp done"
done
unset FINROM_NO_OVERLAP
# the opt-in offline/online form of the reduced operator: kernel stats of the same step (overlapped and serialised)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/oo_overlapped -o run -- python $B --projection offline_online > $out/bench_oo_overlapped.log 2>&1 || exit 1
export FINROM_NO_OVERLAP=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/oo_serial -o run -- python $B --projection offline_online > $out/bench_oo_serial.log 2>&1 || exit 1
unset FINROM_NO_OVERLAP
python tools/pmc_summary.py $out > $out/summary.log 2>&1
cat $out/summary.log
