"""One-sample grad_romml latency (BASELINE configs[4] call pattern) and, with FINROM_CLOCK_PROBE=7, the split-K kernel's phase clocks.
usage (GPU box): [FINROM_CLOCK_PROBE=7] python tools/hmc_probe.py [r] [n_calls]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench
from bayesianinferencedl_amd import _ffi
from bayesianinferencedl_amd.fom.forward_solve import Fin
from bayesianinferencedl_amd.fom.thermal_fin import get_space
from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
from bayesianinferencedl_amd.rom.basis import pod_basis

r = int(sys.argv[1]) if len(sys.argv) > 1 else 81
n_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
V = get_space(None, m=12)
fin = Fin(V)
phi = pod_basis(fin, r, n_snapshots=200, low=0.1, high=10.0, params="nine", seed=1)
rom = AffineROMFin(V, bench.hmc_error_model(V.dim()), phi)
rng = np.random.default_rng(0)
rom.set_data(rng.uniform(0.2, 1.0, 9))
K = np.exp(0.1 * rng.standard_normal((4, V.dim())))
for S in (1, 4):
    for _ in range(3):
        rom.grad_romml_batch(K[:S])
    L = _ffi.lib(); L.finrom_profile_reset(); L.finrom_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(n_calls):
        rom.grad_romml_batch(K[:S])
    dt = (time.perf_counter() - t0) / n_calls
    L.finrom_profile_enable(0)
    prof = {k: round(v[1] / v[0] * 1e3, 1) for k, v in _ffi.profile_read().items() if v[0]}
    print(f"S={S} r={r}: {dt * 1e3:.3f} ms per call; kernel us per launch: {prof}", flush=True)
