"""Model-constrained adaptive (greedy) sampling of the trial basis, same call surface as the reference's
rom/model_constr_adaptive_sampling.py: `sample` (:4-48) repeatedly asks `optimizer` for the parameter with the largest
ROM error, solves the FOM there and enriches the basis by one Gram-Schmidt step (`enrich`, :50-68)."""
from __future__ import annotations

import time

import numpy as np

from ..fem import as_nodal
from .basis import enrich  # noqa: F401  (re-exported: the reference defines it in this module)


def sample(basis, random_initial, optimizer, solver, tol=1.0e-14, maxiter=80, verbose=False):
    """basis [n, k]; random_initial() -> starting parameter; optimizer(z_0, basis, solver) -> (z_star, G(z_star));
    solver.forward(z_star)[0] is the FOM state.  Returns the enriched basis [n, k + iterations]."""
    iterations = 0
    g_z_star = 1e30
    while g_z_star > tol and iterations < maxiter:
        z_0 = random_initial()
        t_i = time.time()
        prev = g_z_star
        z_star, g_z_star = optimizer(z_0, basis, solver)
        iterations += 1
        w = as_nodal(solver.forward(z_star)[0]).reshape(-1, 1)
        basis = enrich(basis, w)
        if verbose:
            print(f"sampler iteration {iterations}: {time.time() - t_i:.3f} s, error {g_z_star:.3e}, improvement {prev - g_z_star:.3e}")
    return basis


def initial_pod_basis(solver, basis_size=5, samples=10, low=0.1, high=1.0, rng=None):
    """Start basis of the reference's driver (rom/generate_reduced_basis_five_param.py:31-49): `samples` FOM snapshots for
    kappa ~ U(low, high)^5 (one batched device solve), combined with the leading eigenvectors of their Gram matrix."""
    rng = np.random.default_rng() if rng is None else rng
    kappa = rng.uniform(low, high, (samples, 5))
    Y = np.asarray(solver.forward_batch(kappa, want_w=True, params="five")["w"])
    e, v = np.linalg.eigh(Y @ Y.T)
    order = np.argsort(e)[::-1][:basis_size]
    return np.ascontiguousarray((v[:, order].T @ Y).T)


def generate_five_param_basis(V, basis_size=5, samples=10, tol=1.0e-14, maxiter=80, n_candidates=4096, seed=None,
                              verbose=False):
    """The reference's rom/generate_reduced_basis_five_param.py as a function: POD start basis + greedy sampling with the
    batched worst-case search.  Returns the basis [n, basis_size + iterations] (write it with np.savetxt(..., delimiter=','))."""
    from ..fom.forward_solve import Fin
    from .error_optimization import optimize_five_param
    rng = np.random.default_rng(seed)
    solver = Fin(V)
    basis = initial_pod_basis(solver, basis_size, samples, rng=rng)

    def random_initial():
        return rng.uniform(0.1, 1.0, 5)                      # :51-57

    def optimizer(z_0, phi, s):
        return optimize_five_param(z_0, phi, s, n_candidates=n_candidates, rng=rng)
    return sample(basis, random_initial, optimizer, solver, tol=tol, maxiter=maxiter, verbose=verbose)
