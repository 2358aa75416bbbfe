"""Worst-case search of the greedy basis sampler.

The reference finds the parameter with the largest ROM error, G(kappa) = 1/2 |y(kappa) - y_r(kappa)|^2, by L-BFGS-B from
one random start with one FOM + one dense reduced solve per cost evaluation (rom/error_optimization.py:14-37, 76-111).
On the device the same question is answered by evaluating G for thousands of candidates per library call
(finrom_solve_pairs: FOM and LSPG ROM halves side by side) and refining around the incumbent.

Differences from the reference, on purpose: y_r comes from the least-squares Petrov-Galerkin ROM of the hot path
(psi = A(kappa) Phi for every candidate, rom/averaged_affine_ROM.py:289-304), whereas the reference's cost freezes
psi = A(kappa_0) Phi at the optimiser's starting point (:20-21); and G is the scalar 1/2 sum of squares, whereas
the reference's `cost_functional` returns the element-wise vector (:86)."""
from __future__ import annotations

import numpy as np

from ..pairs import FinPairSolver

BOUNDS = (0.1, 10.0)          # rom/error_optimization.py:26


def rom_error_batch(kappa, phi, solver, params="five"):
    """G(kappa_s) = 1/2 |y - y_r|^2 for a batch kappa [S, 5|9] -> ndarray [S]; failed samples give -inf."""
    pairs = FinPairSolver(solver.V, phi, params=params, solver=solver)
    res = pairs.solve_pairs(np.ascontiguousarray(kappa, dtype=np.float64))
    g = 0.5 * np.sum(np.asarray(res["err"]) ** 2, axis=1)
    g[np.asarray(res["info"]) != 0] = -np.inf
    return g


def _search(k_0, phi, solver, params, n_candidates, n_refine, rng):
    rng = np.random.default_rng() if rng is None else rng
    lo, hi = BOUNDS
    dim = len(k_0)
    cand = np.vstack([np.asarray(k_0, float)[None, :], rng.uniform(lo, hi, (n_candidates - 1, dim))])
    g = rom_error_batch(cand, phi, solver, params)
    best, g_best = cand[np.argmax(g)].copy(), float(np.max(g))
    width = 0.25 * (hi - lo)
    for _ in range(n_refine):                      # shrinking boxes around the incumbent
        cand = np.clip(best + rng.uniform(-width, width, (max(n_candidates // 4, 8), dim)), lo, hi)
        g = rom_error_batch(cand, phi, solver, params)
        if np.max(g) > g_best:
            best, g_best = cand[np.argmax(g)].copy(), float(np.max(g))
        width *= 0.35
    return best, g_best


def optimize_five_param(k_0, phi, solver, n_candidates=4096, n_refine=3, rng=None):
    """-> (z_star as a Function, G(z_star)); same call shape as the reference's optimiser (:14-37)."""
    best, g_best = _search(k_0, phi, solver, "five", n_candidates, n_refine, rng)
    return solver.five_param_to_function(best), g_best


def optimize_nine_param(k_0, phi, solver, n_candidates=4096, n_refine=3, rng=None):
    best, g_best = _search(k_0, phi, solver, "nine", n_candidates, n_refine, rng)
    return solver.nine_param_to_function(best), g_best
