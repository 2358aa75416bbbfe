"""Reduced bases for the fin.  The reference ships bases tied to its irreproducible mshr
mesh (data/basis_*_param.txt, 1446 x 81), so a basis for OUR mesh is built
with the reference's POD recipe (rom/generate_reduced_basis_nine_param.py:296-318:
snapshots of the FOM for k ~ U(0.1, 3.5)^9) using the batched device FOM, then
orthonormalised (parity on raw w_r needs orthonormal columns)."""
from __future__ import annotations

import os

import numpy as np


def pod_basis(solver, r, n_snapshots=400, low=0.1, high=3.5, params="nine", seed=1):
    rng = np.random.default_rng(seed)
    dim = {"nine": 9, "five": 5}[params]
    kappa = rng.uniform(low, high, size=(n_snapshots, dim))
    Y = np.asarray(solver.forward_batch(kappa, want_w=True, params=params)["w"])   # batched device FOM
    # one-time host setup.  One BLAS thread: a threaded SVD blocks differently for different thread counts, and ranks of a
    # multi-GPU run (OMP_NUM_THREADS=1 each) must build the SAME basis as a lone process, bit for bit
    from ..fem import deterministic_blas
    with deterministic_blas():
        _, _, Vt = np.linalg.svd(Y, full_matrices=False)
    return np.ascontiguousarray(Vt[:r].T)


def load_basis_csv(path):
    """Reader for the reference's basis format: np.savetxt(..., delimiter=',')."""
    return np.loadtxt(path, delimiter=",")


def load_or_build_basis(V, solver, path=None, r=81):
    if path is not None and os.path.exists(path):
        phi = load_basis_csv(path)
        if phi.shape[0] == V.dim():
            return phi
    return pod_basis(solver, r)


def enrich(basis, w):
    """Gram-Schmidt enrichment of a basis with one snapshot, as the reference's greedy sampler
    does it (rom/model_constr_adaptive_sampling.py:50-68): note its loop stops at column k-2
    (``range(0, k-1)``), i.e. the last existing column is NOT projected out; kept as is."""
    U = np.hstack((np.asarray(basis, float), np.asarray(w, float).reshape(-1, 1)))
    k = basis.shape[1]
    for j in range(k - 1):
        U[:, -1] -= (U[:, -1] @ U[:, j]) / (U[:, j] @ U[:, j]) * U[:, j]
    U[:, -1] /= np.sqrt(U[:, -1] @ U[:, -1])
    return U
