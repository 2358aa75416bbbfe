"""FOM forward solve at lattice m: the frontal band sweep against the schedule interpreter it replaces, same inputs (nodal field),
QoI-only and with w.     python tools/fom_band_vs_interp.py [m] [S]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesianinferencedl_amd.engine as E
from bayesianinferencedl_amd.fom.forward_solve import Fin
from bayesianinferencedl_amd.fom.thermal_fin import get_space

m = int(sys.argv[1]) if len(sys.argv) > 1 else 24
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
V = get_space(None, m=m)
X = torch.from_numpy(np.exp(0.5 * np.random.default_rng(0).standard_normal((S, V.dim())))).cuda()


def timed(fin, want_w, reps=5):
    eng = fin._engine("field")
    eng.set_small_max(0)
    for _ in range(2):
        res = fin.forward_batch(X, want_w=want_w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = fin.forward_batch(X, want_w=want_w)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, eng.last_path(), res


fin_b = Fin(V)
E.USE_BAND = False
fin_i = Fin(V)
fin_i._engine("field")
E.USE_BAND = True
out = {}
for want_w in (False, True):
    tb, pb, rb = timed(fin_b, want_w)
    ti, pi, ri = timed(fin_i, want_w)
    qb, qi = rb["qoi"].cpu().numpy(), ri["qoi"].cpu().numpy()
    rel = np.max(np.linalg.norm(qb - qi, axis=1) / np.linalg.norm(qi, axis=1))
    print(f"m={m} n={V.dim()} S={S} want_w={want_w}: {pb} {tb:.2f} ms | {pi} {ti:.2f} ms | x{ti / tb:.2f} | max rel qoi diff {rel:.2e}", flush=True)
