#!/usr/bin/env python3
"""Generates tests/golden/fin_m4_r8.npz with the CPU oracle (oracle/fin_oracle.py).

The reference itself cannot run here (dolfin/mshr/petsc4py/tensorflow absent) and holds no golden
vectors for this path, so these are ORACLE-generated regression vectors: tiny mesh
m = 4 (n = 245), orthonormal POD basis r = 8, 16 conductivity samples of each input kind.
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fin_oracle as O  # noqa: E402


def main():
    m, r, S = 4, 8, 16
    prob = O.FinProblem(m)
    fo = O.FinOracle(prob)
    rng = np.random.default_rng(0)
    Y = np.array([fo.forward(fo.nine_param_to_function(rng.uniform(0.1, 3.5, 9))) for _ in range(40)])
    phi = O.pod_basis(Y, r)
    ro = O.AffineROMOracle(prob, phi)
    k5 = rng.uniform(0.1, 10.0, (S, 5))
    k9 = rng.uniform(0.1, 10.0, (S, 9))
    chol = O.make_cov_chol(prob.coords, 'm52', 1.6)
    xi = rng.standard_normal((S, prob.n))
    fields = O.sample_fields(chol, xi)
    out = {"m": m, "phi": phi, "k5": k5, "k9": k9, "xi": xi, "fields": fields}
    for name, X, lift in (("five", k5, fo.five_param_to_function), ("nine", k9, fo.nine_param_to_function),
                          ("field", fields, lambda x: x)):
        W = np.array([fo.forward(lift(x)) for x in X])
        WR = np.array([ro.forward_reduced(lift(x)) for x in X])
        out[f"w_{name}"] = W
        out[f"qoi_{name}"] = W @ fo.B_obs.T
        out[f"w_r_{name}"] = WR
        out[f"qoi_r_{name}"] = WR @ ro.B_obs_phi.T
        out[f"theta_{name}"] = np.array([prob.S @ lift(x) for x in X])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fin_m4_r8.npz"), **out)
    print("wrote fin_m4_r8.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
