# usage: bash tools/exp_env.sh "VAR=val VAR=val ..." ...   -- one bench.py run per argument with those variables set
for cfg in "$@"; do
  echo "=== $cfg"
  env $cfg timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-samples 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['kernels_avg_ms'], d['kernels_serial_ms'])"
done
