"""Latency of the scalar (one-sample) call surface: the MAP / HMC call pattern of BASELINE configs[4].
usage: python tools/scalar_latency.py [n_calls]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bayesianinferencedl_amd.fem import Function
from bayesianinferencedl_amd.fom.forward_solve import Fin
from bayesianinferencedl_amd.fom.thermal_fin import get_space
from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
from bayesianinferencedl_amd.rom.basis import pod_basis
from bayesianinferencedl_amd.deep_learning.dl_model import ResBnFcModel

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
V = get_space(40)
fin = Fin(V)
phi = pod_basis(fin, 80)
model = ResBnFcModel(V.dim(), 9, 5, 50)
rom = AffineROMFin(V, model, phi)
rng = np.random.default_rng(0)
rom.set_data(rng.uniform(0.2, 1.0, 9))
ks = [Function(V, np.exp(0.3 * rng.standard_normal(V.dim()))) for _ in range(8)]


def bench(name, f):
    for k in ks:
        f(k)
    t0 = time.perf_counter()
    for i in range(n_calls):
        f(ks[i % 8])
    print(f"{name}: {1e3 * (time.perf_counter() - t0) / n_calls:.3f} ms per call")


bench("AffineROMFin.forward_reduced + qoi_reduced", lambda k: rom.qoi_reduced(rom.forward_reduced(k)))
bench("AffineROMFin.grad_reduced", lambda k: rom.grad_reduced(k))
bench("AffineROMFin.grad_romml", lambda k: rom.grad_romml(k))
bench("Fin.forward + qoi_operator", lambda k: fin.qoi_operator(fin.forward(k)[0]))
bench("Fin.gradient", lambda k: fin.gradient(k, rom.data))
