#!/bin/bash
# Run on the GPU box (through gpurun):  bash tools/collect_profiles.sh <tag>
# Produces under gpurun_out/prof_<tag>/: rocprofv3 --kernel-trace --stats summaries of bench.py (the timed configuration with
# the two halves side by side, and serialised), the bench JSON lines, one --pmc pass per counter group (halves serialised),
# and kernel stats of BASELINE configs[2] / [3] / [4].  python is named directly after "--" (the profiler preloads the GPU
# runtime: no env/bash hops).
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
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  d=$out/pmc_$(echo $grp | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $grp --output-format csv -d $d -o run -- python $B1 > $d.log 2>&1 || exit 1
  echo "pmc $grp done"
done
unset FINROM_NO_OVERLAP
# the opt-in offline/online form of the reduced operator
rocprofv3 --kernel-trace --stats --output-format csv -d $out/oo_overlapped -o run -- python $B --projection offline_online > $out/bench_oo_overlapped.log 2>&1 || exit 1
# BASELINE configs[2] (nine parameters, r = 120) and configs[3] (Gaussian field, m = 20, r = 200; 20k samples of the 125k shard)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3 -o run -- python $B --params nine --r 120 > $out/bench_c3.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c4 -o run -- python $B --params field --m 20 --r 200 --samples 20000 > $out/bench_c4.log 2>&1 || exit 1
python tools/pmc_summary.py $out > $out/summary.log 2>&1
cat $out/summary.log
