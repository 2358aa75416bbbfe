"""finrom_fom_gradient throughput (Fin.gradient for a batch, fom/forward_solve.py:293-322): 100k value-and-gradient evaluations.
usage (GPU box): python tools/fom_grad_bench.py [params=five|nine|field] [m=12] [S=100000]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from bayesianinferencedl_amd import _ffi
from bayesianinferencedl_amd.fom.forward_solve import Fin
from bayesianinferencedl_amd.fom.thermal_fin import get_space

params = sys.argv[1] if len(sys.argv) > 1 else "five"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 12
S = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
V = get_space(None, m=m)
fin = Fin(V)
dim = {"five": 5, "nine": 9, "field": V.dim()}[params]
rng = np.random.default_rng(0)
X = torch.from_numpy(np.exp(0.3 * rng.standard_normal((S, dim))) if params == "field" else rng.uniform(0.1, 10.0, (S, dim))).cuda()
data = torch.from_numpy(rng.uniform(0.1, 0.6, 9)).cuda()
kw = dict(params=None if params == "field" else params)
for _ in range(2):
    res = fin.gradient_batch(X, data, **kw)
torch.cuda.synchronize()
L = _ffi.lib(); L.finrom_profile_reset(); L.finrom_profile_enable(1)
t0 = time.perf_counter()
for _ in range(5):
    res = fin.gradient_batch(X, data, **kw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
L.finrom_profile_enable(0)
prof = {k: round(v[1] / v[0], 3) for k, v in _ffi.profile_read().items() if v[0]}
print(f"{params} m={m} S={S}: {dt * 1e3:.2f} ms per batch = {S / dt / 1e6:.2f} M value+gradient/s, path {fin._engine(params if params != 'field' else 'field').last_path()}, "
      f"failed {int((res['info'] != 0).sum())}; kernel ms per launch: {prof}", flush=True)
