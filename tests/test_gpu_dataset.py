"""End-to-end dataset generator (deep_learning/generate_fin_dataset.py:62-111 call surface): sampler ->
FOM -> sub-fin averages -> ROM -> error, .npy names and shapes of the reference (:102-110)."""
import os

import numpy as np
import pytest

from oracle import fin_oracle as O

pytestmark = pytest.mark.gpu


def test_gen_affine_avg_rom_dataset_small(tmp_path, problems):
    from bayesianinferencedl_amd.deep_learning.generate_fin_dataset import gen_affine_avg_rom_dataset
    from bayesianinferencedl_amd.fom.thermal_fin import get_space
    import bayesianinferencedl_amd.fom.thermal_fin as tf
    # resolution 14 -> lattice divisor m = 4 (n = 245): small enough for the oracle to follow
    V = get_space(14)
    assert V.dim() == 245
    prob = problems(4)
    fo = O.FinOracle(prob)
    rng = np.random.default_rng(0)
    Y = np.array([fo.forward(fo.nine_param_to_function(rng.uniform(0.1, 3.5, 9))) for _ in range(40)])
    phi = O.pod_basis(Y, 8)
    S = 200
    z_s, qoi_errors = gen_affine_avg_rom_dataset(S, resolution=14, phi=phi, seed=11, out_dir=str(tmp_path), batch=128)
    assert z_s.shape == (S, 245) and qoi_errors.shape == (S, 9)
    # S < 600 -> the reference's "eval" file names (:106-110)
    for name, shape in (("z_aff_avg_eval_avg_obs_3", (S, 245)), ("errors_aff_avg_eval_avg_obs_3", (S, 9)),
                        ("qois_avg_eval_avg_obs_3", (S, 9))):
        a = np.load(os.path.join(tmp_path, name + ".npy"))
        assert a.shape == shape
    # same draws on the CPU: np.random.RandomState(seed).randn per batch, exp(0.5 chol.T @ xi)
    chol = O.make_cov_chol(prob.coords, 'm52', 1.6)
    rs = np.random.RandomState(11)
    xi = np.concatenate([rs.randn(128, 245), rs.randn(S - 128, 245)])
    fields = O.sample_fields(chol, xi)
    assert np.max(np.abs(z_s - fields) / fields) < 1e-12
    _, err, q, qr = O.gen_affine_avg_rom_dataset(prob, phi, fields[:24])
    assert np.max(np.abs(qoi_errors[:24] - err)) < 1e-10 * np.max(np.abs(q))
    qois = np.load(os.path.join(tmp_path, "qois_avg_eval_avg_obs_3.npy"))
    assert np.max(np.linalg.norm(qois[:24] - q, axis=1) / np.linalg.norm(q, axis=1)) < 1e-10


def test_basis_csv_roundtrip(tmp_path, spaces):
    """Bases travel as np.savetxt(..., delimiter=',') files (rom/generate_reduced_basis_five_param.py:69)."""
    from bayesianinferencedl_amd.rom.basis import load_basis_csv, load_or_build_basis, pod_basis
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    V = spaces(4)
    solver = Fin(V)
    phi = pod_basis(solver, 6, n_snapshots=30, params="nine", seed=3)
    assert np.allclose(phi.T @ phi, np.eye(6), atol=1e-12)
    path = os.path.join(tmp_path, "basis_nine_param.txt")
    np.savetxt(path, phi, delimiter=",")
    assert np.array_equal(load_basis_csv(path), np.loadtxt(path, delimiter=","))
    assert np.allclose(load_or_build_basis(V, solver, path), phi, rtol=0, atol=1e-17)
