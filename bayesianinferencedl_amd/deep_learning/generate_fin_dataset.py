"""Batched, GPU-resident version of the reference's dataset generator
``gen_affine_avg_rom_dataset`` (deep_learning/generate_fin_dataset.py:62-111): draw S
Gaussian-random-field conductivities, solve the FOM and the sub-fin-averaged affine ROM for
each, record the QoIs and the ROM error.  Same name, arguments, return value and .npy file
names; the per-sample Python loop of the reference (:83-100) is replaced by batched device
calls (sampler -> FOM -> sub-fin average -> ROM -> difference)."""
from __future__ import annotations

import logging
import os

import numpy as np

from ..bayesian_inference.gaussian_field import make_cov_chol
from ..engine import FieldSampler
from ..fom.forward_solve import Fin
from ..fom.thermal_fin import get_space
from ..pairs import FinPairSolver
from ..rom.averaged_affine_ROM import AffineROMFin
from ..rom.basis import load_or_build_basis

log = logging.getLogger(__name__)


def gen_affine_avg_rom_dataset(dataset_size, resolution=40, genrand=False, *, phi=None, seed=None,
                               out_dir='../data', basis_path='../data/basis_nine_param.txt', batch=65536):
    V = get_space(resolution)
    solver = Fin(V, genrand)                                           # :66
    if phi is None:
        phi = load_or_build_basis(V, solver, basis_path)               # :67 (see rom/basis.py)
    chol = make_cov_chol(V, length=1.6)                                # :69
    solver_r = AffineROMFin(V, None, phi, genrand)                     # :76 (err_model unused by the forward path)
    pairs = FinPairSolver(V, phi, genrand, "field", solver, solver_r)
    sampler = FieldSampler(chol)

    qoi_errors = np.zeros((dataset_size, solver_r.n_obs))
    qois = np.zeros((dataset_size, solver_r.n_obs))
    z_s = np.zeros((dataset_size, V.dim()))
    rng = np.random.RandomState(seed)      # the reference uses the unseeded global state (:87)
    for s0 in range(0, dataset_size, batch):
        s1 = min(dataset_size, s0 + batch)
        norm = rng.randn(s1 - s0, V.dim())                             # :87
        nodal_vals = sampler(norm)                                     # :88  exp(0.5 * chol.T @ norm)
        res = pairs.solve_pairs(nodal_vals)                            # :93-97
        z_s[s0:s1] = nodal_vals
        qoi_errors[s0:s1] = res["err"]                                 # :99
        qois[s0:s1] = res["qoi"]                                       # :100

    if out_dir is not None and os.path.isdir(out_dir):                 # :102-110
        if dataset_size > 1000:
            np.save(os.path.join(out_dir, 'z_aff_avg_tr_avg_obs_3'), z_s)
            np.save(os.path.join(out_dir, 'errors_aff_avg_tr_avg_obs_3'), qoi_errors)
            np.save(os.path.join(out_dir, 'qois_avg_tr_avg_obs_3'), qois)
        if dataset_size < 600:
            np.save(os.path.join(out_dir, 'z_aff_avg_eval_avg_obs_3'), z_s)
            np.save(os.path.join(out_dir, 'errors_aff_avg_eval_avg_obs_3'), qoi_errors)
            np.save(os.path.join(out_dir, 'qois_avg_eval_avg_obs_3'), qois)
    elif out_dir is not None:
        log.warning("output directory %s does not exist; dataset not saved", out_dir)
    return (z_s, qoi_errors)
