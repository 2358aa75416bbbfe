#!/bin/bash
# Run on the GPU box: bash tools/corun_experiments.sh <outdir>
# What the projection kernel loses beside the FOM band sweep (headline workload), decomposed by removing one resource at a time
# (timing experiments: the runs with FINROM_CLOCK_PROBE=2 / FINROM_BAND_NOMEM=1 produce garbage results on purpose):
#   A  default (both halves side by side)            B  projection without its table stream (every fetch hits slot 0)
#   C  sweep without its L / y / w stream            D  both            E  halves in turn (FINROM_NO_OVERLAP=1)
out=${1:-gpurun_out/corun}
mkdir -p $out
run() {
  tag=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-samples 0 --no-host-io --no-other > $out/$tag.json 2> $out/$tag.err || { echo "$tag failed"; tail -3 $out/$tag.err; return 1; }
  python - $out/$tag.json $tag <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k, s = d["kernels_avg_ms"], d["kernels_serial_ms"]
print(f"{sys.argv[2]:4s} step {d['ms_per_step']:6.2f} ms  {d['value']/1e6:5.2f} M pairs/s | beside: proj {k.get('rom_proj_mfma',0):6.2f} sweep {k.get('fom_chol_solve',0):5.2f} | alone: proj {s.get('rom_proj_mfma',0):6.2f} sweep {s.get('fom_chol_solve',0):5.2f}")
PY
}
run A FINROM_X=0 && run B FINROM_CLOCK_PROBE=2 && run C FINROM_BAND_NOMEM=1 && run D FINROM_CLOCK_PROBE=2 FINROM_BAND_NOMEM=1 && run E FINROM_NO_OVERLAP=1
