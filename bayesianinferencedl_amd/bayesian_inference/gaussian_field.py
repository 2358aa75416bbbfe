"""Gaussian-random-field prior of the conductivity: host-side factor of the covariance.

Call surface of the reference's ``make_cov_chol(V, kern_type, length)``
(bayesian_inference/gaussian_field.py:9-31).  This is one-time setup, kept on the host and
routed through the same SciPy entry points the reference uses (``pdist``/``squareform``
for the distances, ``scipy.linalg.cholesky`` for the UPPER factor): the Matern-5/2
covariance carries no nugget and is close to singular, so where the factorisation happens
matters for reproducing the reference's samples.  The per-sample work
``k = exp(0.5 * U^T xi)`` (deep_learning/generate_fin_dataset.py:87-88) runs on the GPU
(engine.FieldSampler -> finrom_sampler_draw)."""
import numpy as np
import scipy.linalg
from scipy.spatial.distance import pdist, squareform


def _matern(nu_sqrt, poly):
    def kern(d, length):
        t = nu_sqrt * d / length
        return poly(t) * np.exp(-t)
    return kern


_KERNELS = {
    # :17-21  squared exponential with a 1e-5 nugget
    'sq_exp': lambda d, length: np.exp(-(1 / (2 * length ** 2)) * d ** 2) + 1e-5 * np.eye(len(d)),
    # :22-25  Matern 5/2, no nugget
    'm52': _matern(np.sqrt(5), lambda t: 1 + t + t * t / 3),
    # :26-29  Matern 3/2 (the reference's fall-through branch)
    'm32': _matern(np.sqrt(3), lambda t: 1 + t),
}


def make_cov_chol(V, kern_type='m52', length=1.6):
    xy = V.tabulate_dof_coordinates().reshape((-1, 2))[V.dofmap().dofs(), :]
    kern = _KERNELS.get(kern_type, _KERNELS['m32'])
    from ..fem import deterministic_blas
    with deterministic_blas():           # one LAPACK thread: the same factor, bit for bit, in every rank of a multi-GPU run
        return scipy.linalg.cholesky(kern(squareform(pdist(xy)), length))
