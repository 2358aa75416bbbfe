#!/bin/bash
# Run on the GPU box (through gpurun):  bash tools/collect_profiles.sh <tag>
# Produces under gpurun_out/prof_<tag>/: rocprofv3 --kernel-trace --stats summaries of bench.py (headline: the two halves side by
# side, and in turn), the bench JSON lines, --pmc passes (one counter group per run; halves in turn, because counter collection
# serialises kernels anyway), ONE --pmc pass WITH the kernel trace in the side-by-side configuration (does the profiler let the
# two kernels overlap? -- see pmc_overlap_check), and kernel stats + HBM counters of BASELINE configs[2] / [3], the offline/online
# form and the HMC rehearsal.  python is named directly after "--" (the profiler preloads the GPU runtime: no env / bash hops).
set -o pipefail
tag=${1:-run}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B="bench.py --steps 3 --warmup 1 --cpu-samples 0 --no-host-io --no-other"
B1="bench.py --steps 1 --warmup 1 --cpu-samples 0 --no-profile"
C3="--params nine --r 120"
C4="--params field --m 20 --r 200 --samples 20000"
echo "[1] headline kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/overlapped -o run -- python $B > $out/bench_overlapped.log 2>&1 || exit 1
FINROM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/serial -o run -- python $B > $out/bench_serial.log 2>&1 || exit 1
pmc_pass() {   # <subdir> <bench args...>: HBM / L2 / SQ counter groups, halves in turn
  sub=$1; shift; mkdir -p $out/$sub
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
    d=$out/$sub/pmc_$(echo $grp | tr ' ' '_' | cut -c1-60)
    FINROM_NO_OVERLAP=1 rocprofv3 --pmc $grp --output-format csv -d $d -o run -- python $B1 "$@" > $d.log 2>&1 || { echo "pmc $grp failed"; tail -3 $d.log; return 1; }
    echo "  pmc [$sub] $grp done"
  done
}
echo "[2] headline counters"; pmc_pass headline || exit 1
# ([3] of round 3 -- counters + kernel trace side by side -- answered its question: rocprofv3 serialises the kernels while it collects
#  counters, profiles/r03_overlap_pmc.json; not repeated)
echo "[4] configs[2] nine / r = 120"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3 -o run -- python $B $C3 > $out/bench_c3.log 2>&1 || exit 1
pmc_pass c3 $C3 || exit 1
echo "[5] configs[3] field / m = 20 / r = 200 (20k of the 125k shard)"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c4 -o run -- python $B $C4 > $out/bench_c4.log 2>&1 || exit 1
pmc_pass c4 $C4 || exit 1
echo "[5b] the full 125k shard of configs[3]: HBM counters only (two pieces of 62.5k per step)"
mkdir -p $out/c4full
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  d=$out/c4full/pmc_$(echo $grp | tr ' ' '_')
  FINROM_NO_OVERLAP=1 rocprofv3 --pmc $grp --output-format csv -d $d -o run -- python $B1 --params field --m 20 --r 200 --samples 125000 > $d.log 2>&1 || { echo "pmc c4full $grp failed"; tail -3 $d.log; exit 1; }
  echo "  pmc [c4full] $grp done"
done
echo "[6] offline/online form, HMC rehearsal, full cfg4 shard"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/oo_overlapped -o run -- python $B --projection offline_online > $out/bench_oo_overlapped.log 2>&1 || exit 1
python bench.py --workload hmc --steps 10000 --warmup 100 --hmc-trace > $out/bench_hmc.log 2>&1 || exit 1      # (with cpu_baseline: default --cpu-samples)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/hmc -o run -- python bench.py --workload hmc --steps 1000 --warmup 50 --cpu-samples 0 > $out/bench_hmc_traced.log 2>&1 || exit 1
python tools/hmc_timeline.py $out/hmc/run_kernel_trace.csv > $out/hmc_timeline.txt 2>&1; rm -f $out/hmc/run_kernel_trace.csv
python bench.py --steps 2 --warmup 1 --cpu-samples 0 --no-host-io --no-other --params field --m 20 --r 200 --samples 125000 > $out/bench_c4_full.log 2>&1 || exit 1
FINROM_NO_OVERLAP=1 python bench.py --steps 2 --warmup 1 --cpu-samples 0 --no-host-io --no-other --params field --m 20 --r 200 --samples 125000 > $out/bench_c4_full_in_turn.log 2>&1 || exit 1
echo "[7] the driver's own command"
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_final.log 2>&1 || exit 1
python bench.py --gpus 1 --steps 20 --warmup 5 --stream default --cpu-samples 0 --no-other > $out/bench_final_default_stream.log 2>&1 || exit 1
python tools/pmc_summary.py $out > $out/summary.log 2>&1
cat $out/summary.log
