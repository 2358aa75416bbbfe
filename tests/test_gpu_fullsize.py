"""BASELINE.json-size batches on the GPU, checked through size-independent properties
(no oracle run at this size): mirror symmetry of the fin, determinism, heat balance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MIRROR9 = [8, 7, 6, 5, 4, 3, 2, 1, 0]      # fin1<->fin9, fin2<->fin8, ... (same height, other side)


@pytest.fixture(scope="module")
def setup(spaces):
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    from bayesianinferencedl_amd.rom.basis import pod_basis
    V = spaces(12)
    solver = Fin(V)
    phi = pod_basis(solver, 80, n_snapshots=400, low=0.1, high=10.0, params="five", seed=1)
    assert np.allclose(phi.T @ phi, np.eye(80), atol=1e-10)
    return V, solver, AffineROMFin(V, None, phi), phi


def test_config2_five_param_100k(setup):
    """configs[1]: five-param fin, n = 1597, r = 80, 100k samples."""
    from bayesianinferencedl_amd.pairs import FinPairSolver
    V, solver, solver_r, phi = setup
    S = 100_000
    X = np.random.default_rng(3).uniform(0.1, 10.0, (S, 5))
    pairs = FinPairSolver(V, phi, False, "five", solver, solver_r)
    res = pairs.solve_pairs(X)
    assert (res["info"] == 0).all()
    # a five-parameter conductivity is left/right symmetric, and so are mesh and operators; the POD
    # basis is symmetric only up to the round-off of its SVD, hence the looser ROM tolerance
    for key, tol in (("qoi", 1e-10), ("qoi_r", 1e-6)):
        q = res[key]
        assert np.isfinite(q).all() and (q > 0).all()
        assert np.max(np.abs(q - q[:, MIRROR9]) / np.abs(q)) < tol
    assert np.array_equal(res["err"], res["qoi"] - res["qoi_r"])
    # the hotter the root region conducts, the cooler the post: centre QoI decreases with k5
    again = pairs.solve_pairs(X)
    for key in ("qoi", "qoi_r", "w_r"):
        assert np.array_equal(res[key], again[key]), "batched kernels must be deterministic"
    # sub-fin averages of the interpolated field: centre average is exactly k5
    assert np.max(np.abs(res["theta"][:, 4] - X[:, 4])) < 1e-12 * 10


def test_config3_nine_param_mirror(setup):
    """configs[2]-style: nine independent fin conductivities, r = 120; mirroring the parameters mirrors the QoIs."""
    from bayesianinferencedl_amd.pairs import FinPairSolver
    from bayesianinferencedl_amd.rom.basis import pod_basis
    V, solver, _, _ = setup
    phi = pod_basis(solver, 120, n_snapshots=400, low=0.1, high=3.5, params="nine", seed=2)
    S = 20_000
    X = np.random.default_rng(4).uniform(0.1, 3.5, (S, 9))
    pairs = FinPairSolver(V, phi, False, "nine", solver, None)
    a = pairs.solve_pairs(X)
    b = pairs.solve_pairs(X[:, MIRROR9])
    assert (a["info"] == 0).all() and (b["info"] == 0).all()
    assert np.max(np.abs(a["qoi"] - b["qoi"][:, MIRROR9]) / np.abs(a["qoi"])) < 1e-10
    # the ROM (sub-fin-averaged operator + projection) carries a genuine model error that the
    # reference learns with a network; here only: finite, positive and of the right magnitude
    assert np.isfinite(a["qoi_r"]).all()
    assert np.median(np.abs(a["qoi"] - a["qoi_r"]) / np.abs(a["qoi"])) < 0.2


def test_heat_balance_20k(setup):
    V, solver, _, _ = setup
    ops = V.operators()
    X = np.random.default_rng(5).uniform(0.1, 10.0, (20_000, 9))
    res = solver.forward_batch(X, want_w=True, params="nine")
    robin_colsum = np.asarray(ops.csr(ops.robin_vals).sum(0)).ravel()
    assert np.max(np.abs(res["w"] @ robin_colsum - 1.0)) < 1e-11      # heat in (=1) == heat out


def test_external_observations_40(setup):
    """n_obs = 40 one-hot boundary observations (fom/forward_solve.py:215-228)."""
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    V, _, _, _ = setup
    fin = Fin(V, external_obs=True)
    X = np.random.default_rng(6).uniform(0.1, 10.0, (300, 9))
    res = fin.forward_batch(X, want_w=True, params="nine")
    assert res["qoi"].shape == (300, 40)
    assert np.array_equal(res["qoi"], res["w"] @ fin.B_obs.T)


def test_config4_gaussian_field_m20_r200(spaces, problems):
    """configs[3]-style: Gaussian-random-field conductivity on the m = 20 mesh (n = 4101), r = 200 (13 blocks: the
    8-waves-per-sample projection kernel with the factorisation fused across the waves), 6144 samples through the four-wave
    band sweep (fom_band_ldsw_kernel<7, 22, 8, 4>; the path is asserted); checks against the oracle on samples spread over
    lanes and blocks, heat balance on everything."""
    from oracle import fin_oracle as O
    from bayesianinferencedl_amd.bayesian_inference.gaussian_field import make_cov_chol
    from bayesianinferencedl_amd.engine import FieldSampler
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    from bayesianinferencedl_amd.pairs import FinPairSolver
    from bayesianinferencedl_amd.rom.basis import pod_basis
    m = 20
    V = spaces(m); prob = problems(m)
    assert V.dim() == 4101
    solver = Fin(V)
    phi = pod_basis(solver, 200, n_snapshots=600, low=0.1, high=3.5, params="nine", seed=5)
    S = 6144                                                   # > 4096: the throughput schedule, not the small-batch one
    xi = np.random.default_rng(5).standard_normal((S, V.dim()))
    K = np.asarray(FieldSampler(make_cov_chol(V, length=1.6))(xi))
    assert K.shape == (S, 4101) and (K > 0).all()
    res = FinPairSolver(V, phi, False, "field", solver, None).solve_pairs(K, want_w=True)
    assert (np.asarray(res["info"]) == 0).all()
    assert solver._engine("field").last_path() == "band_lds_4wave"
    bal = np.asarray(res["w"]) @ np.asarray(prob.BiM.sum(0)).ravel()       # heat in = heat out
    assert np.max(np.abs(bal - 1.0)) < 1e-10
    fo = O.FinOracle(prob); ro = O.AffineROMOracle(prob, phi)
    for i in (0, 777, 1234, S // 2, S // 2 + 17, 4099, 5000 + 41, S - 1):      # lanes 0, 9, 18, 0, 17, 3, 49, 63
        q = fo.qoi_operator(fo.forward(K[i])); qr = ro.qoi_reduced(ro.forward_reduced(K[i]))
        assert np.linalg.norm(np.asarray(res["qoi"])[i] - q) < 1e-10 * np.linalg.norm(q)
        assert np.linalg.norm(np.asarray(res["qoi_r"])[i] - qr) < 1e-10 * np.linalg.norm(qr)
