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


def _tags(dataset_size):
    """File-name tags of the reference (:102-110): > 1000 samples -> training set, < 600 -> evaluation set (both: neither)."""
    return (["tr"] if dataset_size > 1000 else []) + (["eval"] if dataset_size < 600 else [])


def gen_affine_avg_rom_dataset(dataset_size, resolution=40, genrand=False, *, phi=None, seed=None,
                               out_dir='../data', basis_path='../data/basis_nine_param.txt', batch=65536,
                               device_rng=False, _stop_after_batches=None):
    """Returns (z_s [S, n], qoi_errors [S, n_obs]) like the reference and writes the three .npy files with its names.

    Host memory is bounded by one batch: when the files are written (out_dir exists and the size selects a file set) the
    outputs are memory-mapped .npy files filled shard by shard (one shard = `batch` samples, flushed as it completes), and the
    returned arrays are those maps -- 1 M samples x 4101 dofs is 33 GB of z_s that never sits in RAM.  With a `seed` the run is
    resumable: a side file records the completed shards and a rerun with the same arguments continues after the last one (the
    random stream is advanced past them).  device_rng=True draws xi on the device (Philox keyed by the global sample index,
    needs a seed) instead of np.random.RandomState(seed).randn per shard (:87)."""
    import json
    V = get_space(resolution)
    solver = Fin(V, genrand)                                           # :66
    if phi is None:
        phi = load_or_build_basis(V, solver, basis_path)               # :67 (see rom/basis.py)
    chol = make_cov_chol(V, length=1.6)                                # :69
    solver_r = AffineROMFin(V, None, phi, genrand)                     # :76 (err_model unused by the forward path)
    pairs = FinPairSolver(V, phi, genrand, "field", solver, solver_r)
    sampler = FieldSampler(chol)
    if device_rng and seed is None:
        raise ValueError("device_rng needs a seed (the device stream is keyed by (seed, global sample index))")

    n, n_obs = V.dim(), solver_r.n_obs
    tags = _tags(dataset_size) if out_dir is not None and os.path.isdir(out_dir) else []
    if out_dir is not None and not os.path.isdir(out_dir):
        log.warning("output directory %s does not exist; dataset not saved", out_dir)
    done, prog_path, meta = 0, None, None
    if tags:
        first = tags[0]
        names = {"z": f"z_aff_avg_{first}_avg_obs_3.npy", "err": f"errors_aff_avg_{first}_avg_obs_3.npy", "qoi": f"qois_avg_{first}_avg_obs_3.npy"}
        paths = {k: os.path.join(out_dir, v) for k, v in names.items()}
        shapes = {"z": (dataset_size, n), "err": (dataset_size, n_obs), "qoi": (dataset_size, n_obs)}
        # what a resumed run must share with the interrupted one: sizes, stream, AND the operators the samples go through -- a
        # rerun with another basis, observation set or covariance factor must not continue into the same files
        import hashlib
        hsh = hashlib.sha256()
        for a_ in (phi, chol):
            hsh.update(np.ascontiguousarray(a_, dtype=np.float64).tobytes())
        meta = {"dataset_size": int(dataset_size), "seed": seed, "batch": int(batch), "n": int(n), "n_obs": int(n_obs),
                "device_rng": bool(device_rng), "genrand": bool(genrand), "operators_sha256": hsh.hexdigest()}
        prog_path = os.path.join(out_dir, f".gen_affine_avg_rom_dataset_{first}.progress.json")
        if seed is not None and os.path.exists(prog_path) and all(os.path.exists(p_) for p_ in paths.values()):
            try:
                with open(prog_path) as f:
                    prev = json.load(f)
            except (OSError, ValueError):                   # killed while the side file was being written: start over
                prev = {}
            if {k: prev.get(k) for k in meta} == meta:
                done = int(prev.get("done", 0))
        mode = "r+" if done else "w+"
        arr = {k: np.lib.format.open_memmap(paths[k], mode=mode, dtype=np.float64, shape=None if done else shapes[k]) for k in paths}
        z_s, qoi_errors, qois = arr["z"], arr["err"], arr["qoi"]
    else:
        qoi_errors = np.zeros((dataset_size, n_obs))
        qois = np.zeros((dataset_size, n_obs))
        z_s = np.zeros((dataset_size, n))
    rng = np.random.RandomState(seed)      # the reference uses the unseeded global state (:87)
    for ib, s0 in enumerate(range(0, dataset_size, batch)):
        s1 = min(dataset_size, s0 + batch)
        if ib < done:
            if not device_rng:
                rng.randn(s1 - s0, n)                                  # advance the stream past a completed shard
            continue
        if _stop_after_batches is not None and ib >= _stop_after_batches:
            break                                                      # (tests: an interrupted run)
        if device_rng:
            nodal_vals = sampler.draw(seed, s0, s1 - s0)               # :87-88 on the device
        else:
            norm = rng.randn(s1 - s0, n)                               # :87
            nodal_vals = sampler(norm)                                 # :88  exp(0.5 * chol.T @ norm)
        res = pairs.solve_pairs(nodal_vals)                            # :93-97
        z_s[s0:s1] = nodal_vals
        qoi_errors[s0:s1] = res["err"]                                 # :99
        qois[s0:s1] = res["qoi"]                                       # :100
        if tags:
            for a_ in (z_s, qoi_errors, qois):
                a_.flush()
            if seed is not None:                            # (through a temporary file: a kill leaves the old or the new one)
                with open(prog_path + ".tmp", "w") as f:
                    json.dump(dict(meta, done=ib + 1), f)
                os.replace(prog_path + ".tmp", prog_path)
    if tags and prog_path and os.path.exists(prog_path) and (_stop_after_batches is None):
        os.remove(prog_path)
    return (z_s, qoi_errors)
