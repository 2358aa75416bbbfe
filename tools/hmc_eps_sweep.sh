#!/bin/bash
# acceptance of the configs[4] rehearsal against the leapfrog step size (GPU box): bash tools/hmc_eps_sweep.sh
for eps in ${EPS_LIST:-0.005 0.01 0.02 0.03 0.05}; do
  timeout -k 10 300 python bench.py --workload hmc --steps 2000 --warmup 20 --cpu-samples 0 --hmc-eps $eps 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c = d['config']
print('eps', c['eps'], 'acceptance', round(c['acceptance'], 3), 'graph', c['hip_graph'], 'evals/s', round(d['value']), 'ms/step', round(d['ms_per_step'], 4), 'one-chain ms/call', round(d['single_chain_latency_ms_per_call'], 4), d['kernels_avg_ms'])"
done
