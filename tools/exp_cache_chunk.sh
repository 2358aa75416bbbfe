# usage: bash tools/exp_cache_chunk.sh "<row_cache> <fwd_chunk> <proj_lds 0|1>" ...   (bench.py one-liners per configuration)
set -o pipefail
for cfg in "$@"; do
  set -- $cfg
  echo "=== ROW_CACHE=$1 CHUNK=$2 PROJ_LDS=$3"
  if [ "$3" = "1" ]; then export FINROM_PROJ_LDS=1; else unset FINROM_PROJ_LDS; fi
  FINROM_ROW_CACHE=$1 FINROM_FWD_CHUNK=$2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-samples 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['kernels_avg_ms'], d['kernels_serial_ms'])"
done
