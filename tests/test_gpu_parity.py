"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.
Tolerance: 1e-10 relative (BASELINE.json north_star, fp64) on QoI_FOM, QoI_ROM, w, Phi w_r;
raw w_r only with an orthonormal basis."""
import numpy as np
import pytest

from oracle import fin_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(autouse=True, params=["small-batch schedule", "throughput schedule"])
def fom_schedule(request, monkeypatch):
    """Every test here runs under both families of FOM schedules.  "throughput schedule": the frontal band sweep on every mesh
    with a band plan (m <= 20; since round 3 it serves small forward batches too), the band adjoint for gradients, no small-batch
    schedule installed.  "small-batch schedule": no band plan installed, so batches of <= 512 samples take the latency-oriented
    kernel (fom_small_kernel: the forward path of meshes without window sizes, and every handle's small-batch GRADIENT path) and
    larger ones the schedule interpreter.  (Which kernel ran is asserted where a test is ABOUT a kernel: tests/test_gpu_band.py,
    test_fwd_chunk_16_..., test_interpreter_forward_path_....)"""
    import bayesianinferencedl_amd.engine as E
    if request.param == "throughput schedule":
        monkeypatch.setattr(E, "SMALL_MAX", 0)
    else:
        monkeypatch.setattr(E, "USE_BAND", False)
    return request.param


def rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return np.max(np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1))


def oracle_basis(prob, r, seed=1, n_snap=None):
    rng = np.random.default_rng(seed)
    fo = O.FinOracle(prob)
    n_snap = n_snap or max(3 * r, 40)
    Y = np.array([fo.forward(fo.nine_param_to_function(rng.uniform(0.1, 3.5, 9))) for _ in range(n_snap)])
    return O.pod_basis(Y, r)


@pytest.mark.parametrize("m,S", [(4, 70), (12, 130)])
def test_fom_field_parity(problems, spaces, m, S):
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    prob = problems(m); V = spaces(m)
    fo = O.FinOracle(prob)
    rng = np.random.default_rng(5)
    K = np.exp(0.5 * rng.standard_normal((S, prob.n)))
    fin = Fin(V)
    res = fin.forward_batch(K, want_w=True)
    n_check = min(S, 24)
    W = np.array([fo.forward(K[i]) for i in range(n_check)])
    Q = W @ fo.B_obs.T
    assert (res["info"] == 0).all()
    assert rel(res["w"][:n_check], W) < TOL
    assert rel(res["qoi"][:n_check], Q) < TOL
    # size-independent property on the whole batch: heat in = heat out, Bi 1^T M_Gamma w = 1
    bal = np.asarray(res["w"]) @ np.asarray(prob.BiM.sum(0)).ravel()
    assert np.max(np.abs(bal - 1.0)) < 1e-11


@pytest.mark.parametrize("params,dim", [("nine", 9), ("five", 5)])
def test_fom_param_parity(problems, spaces, params, dim):
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    m = 12
    prob = problems(m); V = spaces(m)
    fo = O.FinOracle(prob)
    rng = np.random.default_rng(3)
    X = rng.uniform(0.1, 10.0, (67, dim))
    res = Fin(V).forward_batch(X, want_w=True, params=params)
    lift = fo.nine_param_to_function if params == "nine" else fo.five_param_to_function
    W = np.array([fo.forward(lift(X[i])) for i in range(16)])
    assert rel(res["w"][:16], W) < TOL
    assert rel(res["qoi"][:16], W @ fo.B_obs.T) < TOL


@pytest.mark.parametrize("m,r", [(4, 8), (12, 80), (12, 81), (12, 96), (12, 33), (12, 100), (12, 120), (12, 136), (12, 160), (12, 170), (12, 200), (4, 150)])
def test_rom_parity(problems, spaces, m, r):
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, r)
    ro = O.AffineROMOracle(prob, phi)
    rng = np.random.default_rng(4)
    TH = rng.uniform(0.1, 3.5, (130, 9))               # > 64 samples: the throughput kernels (small batches: see the end)
    rom = AffineROMFin(V, None, phi)
    res = rom.forward_nine_param_reduced_batch(TH, want_state=True)
    assert (res["info"] == 0).all()
    n_check = 12
    WR, AR, BR = [], [], []
    for i in range(n_check):
        w_r, A_r, B_r, _ = ro.forward_nine_param_reduced(TH[i], return_parts=True)
        WR.append(w_r); AR.append(A_r); BR.append(B_r)
    WR = np.array(WR); AR = np.array(AR); BR = np.array(BR)
    assert rel(res["A_r"][:n_check].reshape(n_check, -1), AR.reshape(n_check, -1)) < 1e-12
    assert rel(res["B_r"][:n_check], BR) < 1e-12
    assert rel(res["qoi_r"][:n_check], WR @ ro.B_obs_phi.T) < TOL
    assert rel(res["w_r"][:n_check] @ phi.T, WR @ phi.T) < TOL
    assert rel(res["w_r"][:n_check], WR) < 1e-8          # orthonormal basis; cond(A_r) ~ 1e7
    # without the A_r/B_r outputs the reduced matrix is factored inside the projection kernel
    # (in-register blocked Cholesky, r <= 96): a different code path, same contract
    fused = rom.forward_nine_param_reduced_batch(TH)
    assert (fused["info"] == 0).all()
    assert rel(fused["qoi_r"][:n_check], WR @ ro.B_obs_phi.T) < TOL
    assert rel(fused["w_r"][:n_check] @ phi.T, WR @ phi.T) < TOL
    assert rel(fused["w_r"], res["w_r"]) < 1e-8
    # only the reduced QoI wanted (what the sample-pair path asks for): bases wider than 96 then factor A_r and form the QoI
    # inside the registers of the projection kernel's waves (fused_solve_mw) -- a third code path, same contract
    qonly = rom.forward_nine_param_reduced_batch(TH, want_w=False)
    assert "w_r" not in qonly and (qonly["info"] == 0).all()
    assert rel(qonly["qoi_r"][:n_check], WR @ ro.B_obs_phi.T) < TOL
    assert rel(qonly["qoi_r"], res["qoi_r"]) < TOL
    # batches of <= 64 samples (one-sample call patterns) split a sample's k-steps over four waves for 48 < r <= 96
    # (rom_proj_entry_splitk): another summation order, same contract, with and without the A_r / B_r outputs
    small = rom.forward_nine_param_reduced_batch(TH[:7], want_state=True)
    assert (small["info"] == 0).all()
    assert rel(small["A_r"].reshape(7, -1), AR[:7].reshape(7, -1)) < 1e-12 and rel(small["B_r"], BR[:7]) < 1e-12
    assert rel(small["qoi_r"], WR[:7] @ ro.B_obs_phi.T) < TOL and rel(small["w_r"] @ phi.T, WR[:7] @ phi.T) < TOL
    small = rom.forward_nine_param_reduced_batch(TH[:7])
    assert rel(small["qoi_r"], WR[:7] @ ro.B_obs_phi.T) < TOL and rel(small["w_r"] @ phi.T, WR[:7] @ phi.T) < TOL


def test_pairs_field_parity(problems, spaces):
    """The dataset loop body (generate_fin_dataset.py:83-100) on Gaussian-field samples."""
    from bayesianinferencedl_amd.pairs import FinPairSolver
    from bayesianinferencedl_amd.bayesian_inference.gaussian_field import make_cov_chol
    from bayesianinferencedl_amd.engine import FieldSampler
    m = 4
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, 8)
    chol = make_cov_chol(V, length=1.6)
    from bayesianinferencedl_amd.fem import deterministic_blas
    with deterministic_blas():           # the product factors with one LAPACK thread (same factor in every rank)
        assert np.allclose(chol, O.make_cov_chol(prob.coords, length=1.6), rtol=0, atol=0)
    rng = np.random.default_rng(6)
    xi = rng.standard_normal((40, prob.n))
    fields = FieldSampler(chol)(xi)
    assert rel(fields, O.sample_fields(chol, xi)) < 1e-12
    res = FinPairSolver(V, phi).solve_pairs(fields)
    z, err, q, qr = O.gen_affine_avg_rom_dataset(prob, phi, fields)
    assert rel(res["qoi"], q) < TOL
    assert rel(res["qoi_r"], qr) < TOL
    assert np.max(np.abs(res["err"] - err)) < 1e-10 * np.max(np.abs(q))
    assert rel(res["theta"], fields @ prob.S.T) < 1e-13


def test_scalar_call_surface(problems, spaces):
    """Reference-style one-sample calls (generate_fin_dataset.py:90-97)."""
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    from bayesianinferencedl_amd.fem import Function
    m = 4
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, 8)
    fo = O.FinOracle(prob); ro = O.AffineROMOracle(prob, phi)
    solver = Fin(V); solver_r = AffineROMFin(V, None, phi)
    rng = np.random.default_rng(8)
    z = Function(V)
    z.vector().set_local(np.exp(0.3 * rng.standard_normal(prob.n)))
    x, y, A, B, C = solver.forward(z)
    assert y is None and A is None
    w_r = solver_r.forward_reduced(z)
    qoi = solver.qoi_operator(x); qoi_r = solver_r.qoi_reduced(w_r)
    k = z.vector()[:]
    assert rel(x.vector()[:], fo.forward(k)) < TOL
    assert rel(qoi, fo.qoi_operator(fo.forward(k))) < TOL
    assert rel(qoi_r, ro.qoi_reduced(ro.forward_reduced(k))) < TOL
    # the dense reduced state the reference leaves behind (:296-297) is materialised on demand
    _, A_r_o, B_r_o, _ = ro.forward_nine_param_reduced(ro.subfin_avg_op(k), True)
    assert rel(solver_r._A_r, A_r_o) < 1e-12 and rel(solver_r._B_r[None, :], B_r_o[None, :]) < 1e-12
    assert rel(solver.subfin_avg_op(z), prob.S @ k) < 1e-13
    # dense mass / stiffness attributes of the reference (fom :172-173): area 9, constants in the stiffness null space
    assert abs(solver.M.sum() - 9.0) < 1e-12 and np.abs(solver.K @ np.ones(prob.n)).max() < 1e-12
    assert rel(solver_r.forward(z).vector()[:], ro.forward(k)) < TOL
    k5 = rng.uniform(0.1, 1.0, 5)
    assert rel(solver.forward_five_param(k5)[0].vector()[:], fo.forward_five_param(k5)) < TOL


@pytest.mark.parametrize("m,r", [(4, 100), (12, 120), (4, 137), (4, 150), (12, 200)])
def test_half_block_cover_gives_the_aligned_kernels_bits(problems, spaces, m, r, monkeypatch):
    """Bases whose last 16-column block is at most half full (r mod 16 in 1..8) run the multi-wave projection kernels with the
    last tile column and the diagonal tiles re-paired (rom_proj_device.h::HalfCover: fewer MFMAs per k-step) and move the pieces
    into the aligned block triangle afterwards.  Every entry of A_r is still the same sum of the same products in the same
    order: the stored A_r, the separate solve and the in-register factorisation + QoI must agree BIT FOR BIT with the aligned
    kernels (FINROM_PROJ_NO_HALF=1, read per call).  r = 100, 137, 150: NB = 7, 9, 10 (four and eight waves per sample);
    the oracle pins the values (psi^T psi, rom/averaged_affine_ROM.py:291-297)."""
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, r)
    ro = O.AffineROMOracle(prob, phi)
    rom = AffineROMFin(V, None, phi)
    rng = np.random.default_rng(29)
    S = 37
    TH = np.exp(rng.uniform(np.log(0.1), np.log(10.0), (S, 9)))
    full = rom._rom.solve(TH, want_state=True)
    qonly = rom._rom.solve(TH, want_w=False)
    monkeypatch.setenv("FINROM_PROJ_NO_HALF", "1")
    full_a = rom._rom.solve(TH, want_state=True)
    qonly_a = rom._rom.solve(TH, want_w=False)
    monkeypatch.delenv("FINROM_PROJ_NO_HALF")
    for key in ("A_r", "B_r", "w_r", "qoi_r", "info"):    # (a POD basis this wide has noise-level columns: flags may be set -- in both forms)
        assert np.array_equal(full[key], full_a[key], equal_nan=key != "info"), key
    assert np.array_equal(qonly["qoi_r"], qonly_a["qoi_r"], equal_nan=True) and np.array_equal(qonly["info"], qonly_a["info"])
    for i in (0, S - 1):
        w_r, A_r, B_r, psi = ro.forward_nine_param_reduced(TH[i], return_parts=True)
        assert np.max(np.abs(full["A_r"][i] - A_r)) < 1e-12 * np.abs(A_r).max()
        assert np.array_equal(full["A_r"][i], full["A_r"][i].T)


@pytest.mark.parametrize("m,r", [(12, 80), (12, 33), (4, 16)])
def test_grouped_projection_against_the_ungrouped_loop_and_the_oracle(problems, spaces, m, r, monkeypatch):
    """The one-wave projection kernel (r <= 80) accumulates psi^T psi grouped by sub-domain, each group divided by its
    conductivity, and rescales the accumulators where the group changes (rom_proj_device.h::proj_main_grouped; reference
    rom/averaged_affine_ROM.py:291-297 forms psi = A(theta) Phi and the product).  Same sums up to rounding: against a handle
    created with FINROM_PROJ_UNGROUPED=1 (table order, every row multiplied by its conductivity), against the oracle, over the
    dataset's two decades of conductivities; a sample whose conductivities cannot be divided by (zero, subnormal, huge, NaN) takes the ungrouped
    loop inside the same launch -- bit for bit what the ungrouped handle gives, flags included."""
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, r)
    ro = O.AffineROMOracle(prob, phi)
    rom_g = AffineROMFin(V, None, phi)
    monkeypatch.setenv("FINROM_PROJ_UNGROUPED", "1")
    rom_u = AffineROMFin(V, None, phi)
    monkeypatch.delenv("FINROM_PROJ_UNGROUPED")
    rng = np.random.default_rng(23)
    S = 301                                                # (> 64: the batch kernel; a ragged last workgroup)
    TH = np.exp(rng.uniform(np.log(0.1), np.log(10.0), (S, 9)))      # (the dataset's range, generate_fin_dataset.py:75)
    odd = {5: 0.0, 17: 1e-300, 40: 1e200, 77: np.nan, 130: -0.0, 300: np.inf, 222: 1e61, 223: 1e-61}
    for i, v in odd.items():
        TH[i, i % 9] = v
    TH[200, 3] = -2.5                                      # a negative conductivity divides like any other: grouped
    TH[201, 0], TH[201, 8] = 1e50, 1e-50                   # ... and so do 100 decades between two sub-domains (ratio squared: 1e200)
    g = rom_g._rom.solve(TH, want_state=True)
    u = rom_u._rom.solve(TH, want_state=True)
    assert np.array_equal(g["info"], u["info"])
    plain = np.array([i not in odd for i in range(S)])
    assert (g["info"][plain & (np.arange(S) != 200) & (np.arange(S) != 201)] == 0).all()
    sc201 = np.abs(u["A_r"][201]).max()
    assert np.isfinite(sc201) and np.max(np.abs(g["A_r"][201] - u["A_r"][201])) < 1e-12 * sc201
    for key in ("A_r", "B_r", "w_r", "qoi_r"):
        for i in odd:                                      # the ungrouped loop, in the same launch
            assert np.array_equal(g[key][i], u[key][i], equal_nan=True), (key, i)
    ok = plain & (g["info"] == 0)
    scale = np.abs(u["A_r"][ok]).max(axis=(1, 2), keepdims=True)
    assert not np.array_equal(g["A_r"][ok], u["A_r"][ok]), "the grouped loop did not run"
    assert np.max(np.abs(g["A_r"][ok] - u["A_r"][ok]) / scale) < 2e-13
    assert np.allclose(g["qoi_r"][ok], u["qoi_r"][ok], rtol=1e-7, atol=1e-10)
    for i in (0, 1, 150, 299):
        w_r, A_r, B_r, psi = ro.forward_nine_param_reduced(TH[i], return_parts=True)
        assert np.max(np.abs(g["A_r"][i] - A_r)) < 1e-12 * np.abs(A_r).max()
        assert rel(g["qoi_r"][i], ro.B_obs_phi @ w_r) < 1e-7
    # the QoI-only form (what finrom_solve_pairs launches: factorisation and substitutions inside the kernel)
    gq = rom_g._rom.solve(TH, want_w=False); uq = rom_u._rom.solve(TH, want_w=False)
    assert np.array_equal(gq["info"], uq["info"])
    assert np.allclose(gq["qoi_r"][ok], uq["qoi_r"][ok], rtol=1e-7, atol=1e-10)
    for i in odd:
        assert np.array_equal(gq["qoi_r"][i], uq["qoi_r"][i], equal_nan=True), i


@pytest.mark.parametrize("m,r,S", [(12, 80, 200), (12, 120, 70), (12, 33, 65), (4, 150, 64)])
def test_rom_gradient_batched_contraction(problems, spaces, m, r, S):
    """Batches of >= 64 samples contract v_r^T G_pi w_r on the matrix cores (rom_grad_contract_kernel, 16 samples per wave);
    smaller ones keep the contraction inside the substitution kernel.  Same numbers, and the oracle on a few samples."""
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, r)
    ro = O.AffineROMOracle(prob, phi)
    rom = AffineROMFin(V, None, phi)
    rng = np.random.default_rng(19)
    data = rng.uniform(0.1, 1.0, 9)
    ro.set_data(data); rom.set_data(data)
    TH = rng.uniform(0.2, 3.0, (S, 9))
    big = rom.grad_reduced_batch(None, theta=TH)
    assert (big["info"] == 0).all()
    small = [rom.grad_reduced_batch(None, theta=TH[i:i + 7]) for i in range(0, S, 7)]
    g_small = np.concatenate([x["g_theta"] for x in small]); J_small = np.concatenate([x["J"] for x in small])
    # (two kernels, two summation orders for A_r -- the batch kernel accumulates the k-steps grouped by sub-domain, the split-K
    # kernel of the small batches in table order: J agrees to the conditioning of A_r times the rounding unit)
    assert rel(big["g_theta"], g_small) < 1e-9 and np.max(np.abs(big["J"] - J_small) / J_small) < 1e-11
    # per-sample observations through the batched path
    D = rng.uniform(0.1, 1.0, (S, 9))
    res2 = rom._rom.grad(TH, D)
    w_r, A_r, B_r, psi = ro.forward_nine_param_reduced(TH[S - 1], return_parts=True)
    resid = D[S - 1] - ro.B_obs_phi @ w_r
    assert abs(res2["J"][S - 1] - 0.5 * resid @ resid) < 1e-10 * (0.5 * resid @ resid)


def test_info_flags_non_spd(spaces):
    """A negative conductivity makes A(k) indefinite: info != 0 and NaN outputs, no crash."""
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    V = spaces(4)
    X = np.array([[1.0] * 9, [-1.0] * 9, [2.0] * 9])
    res = Fin(V).forward_batch(X, want_w=False, params="nine")
    assert res["info"].tolist() == [0, 1, 0]
    assert np.isnan(res["qoi"][1]).all() and np.isfinite(res["qoi"][[0, 2]]).all()


@pytest.mark.parametrize("r", [80, 120, 200])
def test_rom_info_flags_a_sample_that_cannot_be_factored(problems, spaces, r):
    """A NaN parameter poisons that sample's A_r: its pivot test fails -> info != 0 and NaN QoIs for it alone, on the path
    that returns w_r and on the QoI-only path (fused in registers for r > 96)."""
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    prob = problems(12); V = spaces(12)
    rom = AffineROMFin(V, None, oracle_basis(prob, r))
    TH = np.random.default_rng(2).uniform(0.1, 3.5, (7, 9))
    TH[3, 4] = np.nan
    for want_w in (True, False):
        res = rom.forward_nine_param_reduced_batch(TH, want_w=want_w)
        ok = np.arange(7) != 3
        assert res["info"][3] != 0 and (res["info"][ok] == 0).all()
        assert np.isnan(res["qoi_r"][3]).all() and np.isfinite(res["qoi_r"][ok]).all()


@pytest.mark.parametrize("m,r", [(4, 8), (12, 80), (12, 50), (12, 90), (12, 120), (12, 150), (12, 200)])
def test_rom_adjoint_gradient_parity(problems, spaces, m, r):
    """AffineROMFin.grad_reduced (rom/averaged_affine_ROM.py:335-356) against the oracle restatement."""
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    from bayesianinferencedl_amd.fem import Function
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, r)
    ro = O.AffineROMOracle(prob, phi)
    rom = AffineROMFin(V, None, phi)
    rng = np.random.default_rng(9)
    data = rng.uniform(0.1, 1.0, 9)
    ro.set_data(data); rom.set_data(data)
    K = np.exp(0.3 * rng.standard_normal((6, prob.n)))
    res = rom.grad_reduced_batch(K)
    assert (res["info"] == 0).all()
    for s in range(3):
        g_ref, J_ref = ro.grad_reduced(K[s])
        g = res["g_theta"][s] @ rom.dsigma_dk
        assert abs(res["J"][s] - J_ref) < 1e-10 * abs(J_ref)
        assert np.linalg.norm(g - g_ref) < 1e-8 * np.linalg.norm(g_ref)
    # scalar call surface: (dJ_dk [n], J)
    z = Function(V); z.vector().set_local(K[0])
    g0, J0 = rom.grad_reduced(z)
    g_ref, J_ref = ro.grad_reduced(K[0])
    assert g0.shape == (prob.n,) and abs(J0 - J_ref) < 1e-10 * abs(J_ref)
    assert np.linalg.norm(g0 - g_ref) < 1e-8 * np.linalg.norm(g_ref)
    # per-sample observations
    D = rng.uniform(0.1, 1.0, (6, 9))
    res2 = rom._rom.grad(rom.subfin_avg_batch(K), D)
    ro.set_data(D[4])
    g_ref, J_ref = ro.grad_reduced(K[4])
    assert abs(res2["J"][4] - J_ref) < 1e-10 * abs(J_ref)
    assert np.linalg.norm(res2["g"][4] @ rom.dsigma_dk - g_ref) < 1e-8 * np.linalg.norm(g_ref)


def test_fom_adjoint_gradient_parity(problems, spaces):
    """Fin.gradient (fom/forward_solve.py:293-322) against the oracle restatement, plus a directional
    finite-difference check of J (the reference's own kind of test: bayesian_inference/gradient_fd_test.py)."""
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    from bayesianinferencedl_amd.fem import Function
    for m, S in ((4, 70), (12, 66)):
        prob = problems(m); V = spaces(m)
        fo = O.FinOracle(prob)
        fin = Fin(V)
        rng = np.random.default_rng(12)
        K = np.exp(0.3 * rng.standard_normal((S, prob.n)))
        data = rng.uniform(0.1, 0.6, 9)
        res = fin.gradient_batch(K, data)
        assert (res["info"] == 0).all()
        for s in (0, S - 1):
            g_ref = fo.gradient(K[s], data)
            assert np.linalg.norm(res["grad"][s] - g_ref) < 1e-9 * np.linalg.norm(g_ref)
            J_ref = 0.5 * np.sum((fo.qoi_operator(fo.forward(K[s])) - data) ** 2)
            assert abs(res["J"][s] - J_ref) < 1e-10 * J_ref
        # finite differences of the device J along a random direction
        d = rng.standard_normal(prob.n); d /= np.linalg.norm(d)
        h = 1e-5
        Jp = fin.gradient_batch((K[0] + h * d)[None, :], data)["J"][0]
        Jm = fin.gradient_batch((K[0] - h * d)[None, :], data)["J"][0]
        assert abs((Jp - Jm) / (2 * h) - res["grad"][0] @ d) < 1e-6 * abs(res["grad"][0] @ d) + 1e-12
        # scalar call surface
        z = Function(V); z.vector().set_local(K[1])
        assert np.linalg.norm(fin.gradient(z, data) - res["grad"][1]) < 1e-13 * np.linalg.norm(res["grad"][1])
    # per-fin parametrisation: chain rule through the interpolation
    fin = Fin(spaces(12)); prob = problems(12); fo = O.FinOracle(prob)
    k9 = np.random.default_rng(13).uniform(0.5, 3.0, (3, 9))
    data = np.full(9, 0.3)
    r9 = fin.gradient_batch(k9, data, params="nine")
    g_field = fo.gradient(fo.nine_param_to_function(k9[2]), data)
    N9 = spaces(12).operators().N9
    assert np.linalg.norm(r9["grad"][2] - g_field @ N9) < 1e-9 * np.linalg.norm(g_field @ N9)


@pytest.mark.parametrize("S", [0, 1, 63, 64, 65, 129])
def test_edge_batch_sizes(problems, spaces, S):
    """Empty, single and ragged batches (the kernels work on blocks of 64 samples; the tail block replicates
    the last sample and must not write past the outputs)."""
    from bayesianinferencedl_amd.pairs import FinPairSolver
    m = 4
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, 8)
    fo = O.FinOracle(prob); ro = O.AffineROMOracle(prob, phi)
    rng = np.random.default_rng(11)
    X = rng.uniform(0.1, 10.0, (S, 5))
    res = FinPairSolver(V, phi, params="five").solve_pairs(X)
    assert np.asarray(res["qoi"]).shape == (S, 9) and np.asarray(res["qoi_r"]).shape == (S, 9)
    if S == 0:
        return
    assert (np.asarray(res["info"]) == 0).all()
    for i in {0, S - 1}:
        k = fo.five_param_to_function(X[i])
        q = fo.B_obs @ fo.forward(k)
        qr = ro.qoi_reduced(ro.forward_reduced(k))
        assert np.linalg.norm(np.asarray(res["qoi"])[i] - q) < TOL * np.linalg.norm(q)
        assert np.linalg.norm(np.asarray(res["qoi_r"])[i] - qr) < TOL * np.linalg.norm(qr)


@pytest.mark.parametrize("chunk,cache,m,params", [(16, 5, 4, "nine"), (8, 17, 12, "five"), (8, 13, 12, "nine")])
def test_fwd_chunk_16_stream_gives_the_same_solution(problems, spaces, monkeypatch, chunk, cache, m, params):
    """The interpreter is instantiated for 8- and 16-op prefetch chunks (finrom_fom_desc.fwd_chunk); small row caches
    exercise the LDX / FMAX ops (rows longer than the cache) and the wrap-around of the ring allocation, with the affine
    assembly fused into the stream (17 slots, five parameters) and as a pre-pass (5 and 13 slots, nine parameters)."""
    import bayesianinferencedl_amd.engine as E
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    from bayesianinferencedl_amd.fom.thermal_fin import get_space
    prob = problems(m)
    fo = O.FinOracle(prob)
    rng = np.random.default_rng(4)
    X = rng.uniform(0.1, 10.0, (70, 9 if params == "nine" else 5))
    monkeypatch.setattr(E, "FWD_CHUNK", chunk)
    monkeypatch.setattr(E, "ROW_CACHE_SLOTS", cache)
    monkeypatch.setattr(E, "USE_BAND", False)              # this test is about fom_vm_kernel<8|16>: no band sweep,
    monkeypatch.setattr(E, "SMALL_MAX", 0)                 # no small-batch schedule (whatever the fixture chose)
    fin = Fin(get_space(40, m=m))
    res = fin.forward_batch(X, want_w=True, params=params)
    eng = fin._engine(params)
    assert eng.last_path() == "interpreter"
    assert eng.fused == (cache == 17) and eng.cache_slots == (12 if cache == 17 else cache)
    lift = fo.nine_param_to_function if params == "nine" else fo.five_param_to_function
    W = np.array([fo.forward(lift(X[i])) for i in range(8)])
    assert rel(np.asarray(res["w"])[:8], W) < TOL


@pytest.mark.parametrize("m,r", [(4, 8), (12, 17), (12, 33), (12, 50), (12, 64), (12, 80)])
def test_rom_qoi_only_epilogue_parity(problems, spaces, m, r, fom_schedule):
    """r <= 80 with nothing but the reduced QoI wanted (what finrom_solve_pairs asks for): the single-wave kernel's MFMA-form
    epilogue (rom_proj_device.h::fused_solve_sw -- diagonal tiles by shuffle steps, panels and the extra column
    [B_r | (B_obs Phi)^T] by MFMA, qoi_r = Z[:, 1:]^T Z[:, 0], no backward substitution) against the oracle and against the
    kernel's own solve-based epilogue (want_w=True: chol_tiles + solve_tiles); batch tail included (130 = 32 x 4 + 2)."""
    if fom_schedule != "throughput schedule":
        pytest.skip("ROM only: one run")
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, r)
    ro = O.AffineROMOracle(prob, phi)
    rng = np.random.default_rng(40 + r)
    TH = rng.uniform(0.1, 10.0, (130, 9))
    rom = AffineROMFin(V, None, phi)
    a = rom.forward_nine_param_reduced_batch(TH, want_w=False)          # QoI only: the MFMA-form epilogue
    b = rom.forward_nine_param_reduced_batch(TH, want_w=True)           # with w_r: factorisation + two substitutions
    assert "w_r" not in a and (a["info"] == 0).all() and (b["info"] == 0).all()
    assert rel(a["qoi_r"], b["qoi_r"]) < 1e-11
    Q = np.array([ro.qoi_reduced(ro.forward_nine_param_reduced(TH[i])) for i in (0, 1, 2, 3, 4, 63, 64, 127, 128, 129)])
    assert rel(a["qoi_r"][[0, 1, 2, 3, 4, 63, 64, 127, 128, 129]], Q) < TOL
    # an indefinite reduced operator cannot occur (A_r = psi^T psi), a singular one can: theta = 0 kills every conduction term
    TH[5] = 0.0; TH[129] = np.nan
    c = rom.forward_nine_param_reduced_batch(TH, want_w=False)
    assert c["info"][129] != 0 and np.isnan(c["qoi_r"][129]).all()
    good = np.setdiff1d(np.arange(130), [5, 129])
    assert np.array_equal(c["qoi_r"][good], a["qoi_r"][good]) and (c["info"][good] == 0).all()


@pytest.mark.parametrize("r", [33, 80, 81])
def test_rom_batch_size_thresholds_change_the_kernels_not_the_results(problems, spaces, r, fom_schedule):
    """Batches of <= 64 samples take the one-sample kernels (rom_onesample.hip: contraction split over several workgroups,
    MFMA-form left-looking solve), larger ones the throughput kernels (other summation order): the same samples must agree to
    round-off across the threshold -- which is why shards of a multi-rank run are bit-identical only when every shard is on the
    same side of the thresholds -- and a batch is reproducible bit for bit on its own side."""
    if fom_schedule != "throughput schedule":
        pytest.skip("ROM only: one run")
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    prob = problems(12); V = spaces(12)
    rom = AffineROMFin(V, None, oracle_basis(prob, r))
    TH = np.random.default_rng(r).uniform(0.1, 10.0, (65, 9))
    small = rom.forward_nine_param_reduced_batch(TH[:64])
    large = rom.forward_nine_param_reduced_batch(TH)
    # (two summation orders of psi^T psi and two factorisation orders at cond(A_r) ~ 1e7: 1e-11 measured; the parity metric)
    assert rel(small["qoi_r"], large["qoi_r"][:64]) < TOL
    assert rel(small["w_r"] @ rom.phi.T, large["w_r"][:64] @ rom.phi.T) < TOL
    again = rom.forward_nine_param_reduced_batch(TH[:64])
    assert np.array_equal(again["qoi_r"], small["qoi_r"]) and np.array_equal(again["w_r"], small["w_r"])
    one = rom.forward_nine_param_reduced_batch(TH[7:8])          # a sample alone = the same sample inside a small batch
    assert np.array_equal(one["qoi_r"][0], small["qoi_r"][7])


def test_interpreter_forward_path_on_a_mesh_without_band_plan_sizes(problems, fom_schedule):
    """m = 32 (n = 10017): the library has no window sizes for this mesh (finrom_fom_set_band answers UNSUPPORTED), so the
    throughput path is the schedule interpreter fom_vm_kernel + fom_bwd_kernel -- the one forward-path test on that kernel
    for a mesh where it is the PRODUCT's path; field, nine and five inputs, batch tail, against the oracle."""
    if fom_schedule != "throughput schedule":
        pytest.skip("one run is enough: the test removes the small-batch schedule itself")
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    from bayesianinferencedl_amd.fom.thermal_fin import get_space
    m = 32
    prob = problems(m)
    fo = O.FinOracle(prob)
    fin = Fin(get_space(None, m=m))
    rng = np.random.default_rng(24)
    S = 150
    for params, dim in (("field", prob.n), ("nine", 9), ("five", 5)):
        X = np.exp(0.5 * rng.standard_normal((S, dim))) if params == "field" else rng.uniform(0.1, 10.0, (S, dim))
        eng = fin._engine(params)
        assert eng.band is None
        res = fin.forward_batch(X, want_w=True, params=None if params == "field" else params)
        assert eng.last_path() == "interpreter"
        assert (res["info"] == 0).all()
        lift = {"field": lambda x: x, "nine": fo.nine_param_to_function, "five": fo.five_param_to_function}[params]
        for s_ in (0, 63, 64, 127, 128, S - 1):
            w = fo.forward(lift(X[s_]))
            assert np.linalg.norm(res["w"][s_] - w) < TOL * np.linalg.norm(w), (params, s_)
            q = fo.qoi_operator(w)
            assert np.linalg.norm(res["qoi"][s_] - q) < TOL * np.linalg.norm(q), (params, s_)


def test_create_rejects_corrupt_descriptors(spaces):
    """Descriptor indices are validated on the host: a bad index is an error code, never a GPU fault."""
    import ctypes as C
    from bayesianinferencedl_amd import _ffi
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    lib = _ffi.lib()
    real = lib.finrom_fom_create
    seen = {}

    def spy(dref, href):                 # the descriptor's arrays are alive only during the engine's own call
        d = dref._obj
        h = C.c_void_p()
        old = d.fwd_a[5]
        d.fwd_a[5] = d.nnzL + 3 * d.n + 7                     # load index outside the value vector
        seen["bad_index"] = (real(C.byref(d), C.byref(h)), lib.finrom_last_error())
        d.fwd_a[5] = old
        old_chunk = d.fwd_chunk
        d.fwd_chunk = 12
        seen["bad_chunk"] = real(C.byref(d), C.byref(h))
        d.fwd_chunk = old_chunk
        # a both-operands-in-LDS multiply-add (kind 11) whose slot no FINOFF has written yet, and one beyond the cache
        kinds = np.ctypeslib.as_array(d.fwd_kind, shape=(d.nops_fwd,))
        t11 = int(np.nonzero(kinds == 11)[0][0]) if (kinds == 11).any() else None
        seen["has_fmall"] = t11 is not None
        if t11 is not None:
            old = d.fwd_d[t11]
            d.fwd_d[t11] = d.cache_slots                      # beyond the row cache
            seen["bad_slot"] = real(C.byref(d), C.byref(h))
            d.fwd_d[t11] = old
            first_off = int(np.nonzero(kinds == 5)[0][0])     # before the first FINOFF nothing is in the cache
            ok, oa, ob, od = d.fwd_kind[0], d.fwd_a[0], d.fwd_b[0], d.fwd_d[0]
            assert first_off > 0
            d.fwd_kind[0], d.fwd_a[0], d.fwd_b[0], d.fwd_d[0] = 11, -1, 0, 0
            seen["unwritten_slot"] = real(C.byref(d), C.byref(h))
            d.fwd_kind[0], d.fwd_a[0], d.fwd_b[0], d.fwd_d[0] = ok, oa, ob, od
        old = d.perm[0]
        d.perm[0] = d.perm[1]                                 # not a permutation
        seen["bad_perm"] = real(C.byref(d), C.byref(h))
        d.perm[0] = old
        return real(dref, href)
    try:
        lib.finrom_fom_create = spy
        res = Fin(spaces(4)).forward_batch(np.full((3, 9), 1.0), params="nine")
    finally:
        lib.finrom_fom_create = real
    assert seen["bad_index"][0] != 0 and b"invalid" in seen["bad_index"][1]
    assert seen["bad_chunk"] != 0 and seen["bad_perm"] != 0
    assert seen["has_fmall"] and seen["bad_slot"] != 0 and seen["unwritten_slot"] != 0
    assert (np.asarray(res["info"]) == 0).all()


def test_fom_sensitivity_and_regulariser(problems, spaces):
    """Fin.sensitivity (fom/forward_solve.py:324-342): Jacobian of the observables by n_obs adjoint solves, checked against
    the oracle and a finite difference; Tikhonov reg / grad_reg (:186-191) against the oracle's cell loops."""
    from bayesianinferencedl_amd.fem import Function
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    m = 4
    prob = problems(m); V = spaces(m)
    fo = O.FinOracle(prob)
    rng = np.random.default_rng(8)
    k = np.exp(0.4 * rng.standard_normal(prob.n))
    fin = Fin(V)
    J = fin.sensitivity(Function(V, k))
    assert J.shape == (9, prob.n)
    Jo = fo.sensitivity(k)
    assert np.linalg.norm(J - Jo) < 1e-9 * np.linalg.norm(Jo)
    dk = rng.standard_normal(prob.n)
    eps = 1e-6
    qp = fo.B_obs @ fo.forward(k + eps * dk); qm = fo.B_obs @ fo.forward(k - eps * dk)
    fd = (qp - qm) / (2 * eps)
    assert np.linalg.norm(J @ dk - fd) < 1e-6 * np.linalg.norm(fd)
    # per-fin parameters: chain rule through the interpolation
    kap = rng.uniform(0.5, 5.0, (3, 9))
    Jb = fin.sensitivity_batch(kap, params="nine")
    assert Jb.shape == (3, 9, 9)
    N9 = V.operators().N9
    for s in range(3):
        Js = fo.sensitivity(fo.nine_param_to_function(kap[s])) @ N9
        assert np.linalg.norm(Jb[s] - Js) < 1e-9 * np.linalg.norm(Js)
    # Gauss-Newton Hessian action = J^T J u; also the derivative of the gradient along u where the residual vanishes
    u = rng.standard_normal(prob.n)
    Hu = fin.GN_hessian_action(Function(V, k), Function(V, u))
    assert np.linalg.norm(Hu - Jo.T @ (Jo @ u)) < 1e-8 * np.linalg.norm(Hu)
    data0 = fo.B_obs @ fo.forward(k)
    fd_g = (fo.gradient(k + eps * u, data0) - fo.gradient(k - eps * u, data0)) / (2 * eps)
    assert np.linalg.norm(Hu - fd_g) < 1e-5 * np.linalg.norm(fd_g)
    fin._k.assign(Function(V, k))
    assert abs(fin.reg - fo.reg(k)) < 1e-12 * abs(fo.reg(k))
    assert np.linalg.norm(fin.grad_reg - fo.grad_reg(k)) < 1e-12 * np.linalg.norm(fo.grad_reg(k))


def test_grad_romml_parity(problems, spaces, tmp_path):
    """AffineROMFin.grad_romml (rom/averaged_affine_ROM.py:358-396): ROM + learned-error loss and gradient; the device ROM
    adjoint with shifted data plus the error model's vector-Jacobian product, against the oracle's dense restatement."""
    from bayesianinferencedl_amd.deep_learning.dl_model import ResBnFcModel
    from bayesianinferencedl_amd.fem import Function
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    m = 4
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, 8)
    model = ResBnFcModel(n_in=prob.n, n_out=9, n_layers=2, n_weights=12, seed=5)
    model.save(tmp_path / "err_model.npz")
    model = ResBnFcModel.load(tmp_path / "err_model.npz")
    rng = np.random.default_rng(9)
    data = rng.uniform(0.2, 1.0, 9)
    rom = AffineROMFin(V, model, phi); rom.set_data(data)
    ro = O.AffineROMOracle(prob, phi); ro.set_data(data)
    for _ in range(3):
        k = np.exp(0.3 * rng.standard_normal(prob.n))
        go, lo = O.grad_romml_oracle(ro, model, k)
        rom.set_dl_model(model, device=False)              # network on the host: the oracle's own fp32 GEMV order -> tight
        g, loss = rom.grad_romml(Function(V, k))
        assert abs(loss - lo) < 1e-9 * lo
        assert np.linalg.norm(g - go) < 1e-7 * np.linalg.norm(go)
        rom.set_dl_model(model)                             # network on the device (the default): fp32 in another order
        g, loss = rom.grad_romml(Function(V, k))
        assert abs(loss - lo) < 2e-5 * lo and np.linalg.norm(g - go) < 1e-5 * np.linalg.norm(go)
    # batched: the fp32 network sums in a different order for a batch (BLAS GEMM vs GEMV) -> fp32-level agreement
    K = np.exp(0.3 * rng.standard_normal((5, prob.n)))
    res = rom.grad_romml_batch(K)
    for s in range(5):
        go, lo = O.grad_romml_oracle(ro, model, K[s])
        assert abs(res["loss"][s] - lo) < 1e-5 * lo and np.linalg.norm(res["grad"][s] - go) < 1e-4 * np.linalg.norm(go)


@pytest.mark.parametrize("r", [8, 33, 81])
def test_grad_romml_one_sample_form_at_survey_mesh(problems, spaces, r):
    """The one-sample form of finrom_romml_grad on the survey's mesh (m = 12, n = 1597) for bases of 1, 3 and 6 tiles: the error
    model's forward pass rides in the ROM's contraction kernel and keeps its input in that kernel's dynamic LDS -- at r <= 16 the
    contraction's own three partial triangles (6 KB) are smaller than the input (6.4 KB).  Against the oracle, and a sample
    alone = the same sample inside a batch of four."""
    import torch
    from bayesianinferencedl_amd.deep_learning.dl_model import ResBnFcModel
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    m = 12
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, r)
    model = ResBnFcModel(n_in=prob.n, n_out=9, n_layers=2, n_weights=16, seed=7)
    data = np.linspace(0.2, 1.0, 9)
    rom = AffineROMFin(V, model, phi); rom.set_data(data)
    ro = O.AffineROMOracle(prob, phi); ro.set_data(data)
    K = np.exp(0.3 * np.random.default_rng(r).standard_normal((4, prob.n)))
    res = rom.grad_romml_batch(torch.from_numpy(K).cuda())
    assert (res["info"].cpu().numpy() == 0).all()
    g, loss = res["grad"].cpu().numpy(), res["loss"].cpu().numpy()
    for s_ in (0, 3):
        go, lo = O.grad_romml_oracle(ro, model, K[s_])
        assert abs(loss[s_] - lo) <= 2e-5 * abs(lo) and np.linalg.norm(g[s_] - go) <= 1e-4 * np.linalg.norm(go), (r, s_)
    one = rom.grad_romml_batch(torch.from_numpy(K[2:3]).cuda())
    assert np.array_equal(one["grad"].cpu().numpy()[0], g[2]) and one["loss"].cpu().numpy()[0] == loss[2]


@pytest.mark.parametrize("S", [3, 100])
def test_grad_romml_info_is_this_calls_alone(problems, spaces, S):
    """finrom_romml_grad OVERWRITES info (include/finrom.h): the Python side hands it an uninitialised array.  A field with a NaN
    makes the reduced operator's first pivot NaN -> flag 2 and NaN outputs for that sample only; the next call on clean fields
    reports zeros everywhere -- on the one-sample form (S = 3: the solve kernel stores the flag) and on the batched form
    (S = 100: cleared by the library, flags or-ed in)."""
    import torch
    from bayesianinferencedl_amd.deep_learning.dl_model import ResBnFcModel
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    m = 4
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, 8)
    model = ResBnFcModel(n_in=prob.n, n_out=9, n_layers=2, n_weights=12, seed=5)
    rom = AffineROMFin(V, model, phi); rom.set_data(np.linspace(0.2, 1.0, 9))
    rng = np.random.default_rng(3)
    K = np.exp(0.3 * rng.standard_normal((S, prob.n)))
    clean = rom.grad_romml_batch(torch.from_numpy(K).cuda())
    assert (clean["info"].cpu().numpy() == 0).all() and np.isfinite(clean["grad"].cpu().numpy()).all()
    Kb = K.copy(); Kb[1, 7] = np.nan
    dirty = rom.grad_romml_batch(torch.from_numpy(Kb).cuda())
    info = dirty["info"].cpu().numpy()
    assert info[1] != 0 and (np.delete(info, 1) == 0).all(), info
    g = dirty["grad"].cpu().numpy()
    assert np.array_equal(np.delete(g, 1, axis=0), np.delete(clean["grad"].cpu().numpy(), 1, axis=0))
    for _ in range(3):                                     # fresh, uninitialised info arrays (the allocator hands the old blocks back)
        again = rom.grad_romml_batch(torch.from_numpy(K).cuda())
        assert (again["info"].cpu().numpy() == 0).all()
        assert np.array_equal(again["grad"].cpu().numpy(), clean["grad"].cpu().numpy())


def test_unsupported_basis_size_is_an_error_not_a_crash(spaces):
    """r > 208 (13 blocks of 16) is outside the projection kernels' range: the ROM handle is created, the solve reports
    FINROM_ERR_UNSUPPORTED through the Python layer."""
    from bayesianinferencedl_amd._ffi import FinromError
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    V = spaces(4)
    rng = np.random.default_rng(0)
    phi = np.linalg.qr(rng.standard_normal((V.dim(), 209)))[0]
    with pytest.raises(FinromError):
        AffineROMFin(V, None, phi).forward_nine_param_reduced_batch(np.ones((2, 9)))


def test_small_schedule_rejects_corrupt_levels(spaces):
    """finrom_fom_set_small validates structure, pairs and the level property on the host (a row may only read rows of
    lower levels): a corrupted schedule is an error code."""
    import ctypes as C
    from bayesianinferencedl_amd import _ffi
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    import bayesianinferencedl_amd.engine as E
    if E.SMALL_MAX == 0:
        pytest.skip("small-batch schedule disabled in this parametrisation")
    lib = _ffi.lib()
    real = lib.finrom_fom_set_small
    seen = {}

    def spy(h, dref):
        d = dref._obj
        n_lev = d.nlev_f
        a, b = d.lev_rows_f[0], d.lev_rows_f[d.lev_ptr_f[n_lev] - 1]      # a leaf row and the root row
        d.lev_rows_f[0], d.lev_rows_f[d.lev_ptr_f[n_lev] - 1] = b, a       # root in level 0: reads rows of higher levels
        seen["levels"] = (real(h, C.byref(d)), lib.finrom_last_error())
        d.lev_rows_f[0], d.lev_rows_f[d.lev_ptr_f[n_lev] - 1] = a, b
        old = d.pair_b[0]
        d.pair_b[0] = -5
        seen["pair"] = real(h, C.byref(d))
        d.pair_b[0] = old
        return real(h, dref)
    try:
        lib.finrom_fom_set_small = spy
        res = Fin(spaces(4)).forward_batch(np.full((3, 9), 1.0), params="nine")
    finally:
        lib.finrom_fom_set_small = real
    assert seen["levels"][0] != 0 and b"levels" in seen["levels"][1]
    assert seen["pair"] != 0
    assert (np.asarray(res["info"]) == 0).all()


def test_sq_error_ops_value_and_gradient(problems, spaces):
    """bayesian_inference/pymc_func_bayes_inverse.py:29-167: the misfit value-and-gradient callables behind the reference's
    Theano operators, for the FOM, the ROM and the ROM + learned error, against the oracle."""
    from bayesianinferencedl_amd.bayesian_inference.pymc_func_bayes_inverse import SqErrorOpFOM, SqErrorOpROM, SqErrorOpROMML
    from bayesianinferencedl_amd.deep_learning.dl_model import ResBnFcModel
    m = 4
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, 8)
    rng = np.random.default_rng(12)
    k_true = np.exp(0.3 * rng.standard_normal(prob.n))
    model = ResBnFcModel(n_in=prob.n, n_out=9, n_layers=2, n_weights=10, seed=2)
    fo = O.FinOracle(prob); ro = O.AffineROMOracle(prob, phi)
    data = fo.B_obs @ fo.forward(k_true)
    ro.set_data(data)
    k = np.exp(0.3 * rng.standard_normal(prob.n))
    for Op, want in ((SqErrorOpFOM, (0.5 * np.sum((fo.B_obs @ fo.forward(k) - data) ** 2), fo.gradient(k, data))),
                     (SqErrorOpROM, tuple(reversed(ro.grad_reduced(k)))),
                     (SqErrorOpROMML, tuple(reversed(O.grad_romml_oracle(ro, model, k))))):
        op = Op(V, None, False, phi=phi, k_true=k_true, err_model=model)
        assert np.linalg.norm(op._error_op.obs_data - data) < 1e-10 * np.linalg.norm(data)
        out = [[None], [None]]
        op.perform(None, [k], out)
        # (ROM+ML: the fp32 network runs on the device in another summation order than the oracle's NumPy copy)
        vtol, gtol = (2e-5, 1e-5) if Op is SqErrorOpROMML else (1e-8, 1e-6)
        assert abs(float(out[0][0]) - want[0]) < vtol * abs(want[0])
        assert np.linalg.norm(out[1][0] - want[1]) < gtol * np.linalg.norm(want[1])
        assert np.allclose(op.grad([k], [2.0])[0], 2.0 * out[1][0])


def test_device_philox_sampler_matches_the_oracle_and_ignores_the_shard_cut(problems, spaces):
    """finrom_sampler_draw_seeded: xi drawn on the device (Philox4x32-10 keyed by the global sample index) against the oracle's
    NumPy restatement (pinned by Random123's known-answer vectors), the fields k = exp(0.5 U^T xi) against the oracle's, and
    bit-identical rows for any cut of the sample range."""
    from bayesianinferencedl_amd.bayesian_inference.gaussian_field import make_cov_chol
    from bayesianinferencedl_amd.engine import FieldSampler
    m = 4
    prob, V = problems(m), spaces(m)
    chol = make_cov_chol(V, length=1.6)
    smp = FieldSampler(chol)
    k, xi = smp.draw(5, 1000, 300, want_xi=True)
    ref_xi = O.philox_normal(5, 1000, 300, prob.n)
    assert np.max(np.abs(xi - ref_xi)) < 1e-13
    assert rel(k, O.sample_fields(chol, ref_xi)) < 1e-12
    a, b = smp.draw(5, 1000, 111), smp.draw(5, 1111, 189)
    assert np.array_equal(np.concatenate([a, b]), k)
    assert not np.array_equal(smp.draw(6, 1000, 4), k[:4])


def test_device_error_model_matches_the_host_network(problems, spaces):
    """f3: the fp32 residual network on the device (finrom_mlp_predict / finrom_romml_grad) against its NumPy original
    (deep_learning/dl_model.py::ResBnFcModel, the stand-in for the reference's Keras model, dl_model.py:149-176) -- fp32
    arithmetic in a different summation order: 1e-5 -- and the fused value-and-gradient call against the oracle's dense
    restatement of rom/averaged_affine_ROM.py:358-396 at m = 12, r = 81."""
    import bench
    from bayesianinferencedl_amd.engine import DeviceErrorModel
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    from bayesianinferencedl_amd.rom.basis import pod_basis
    m, r = 12, 81
    prob, V = problems(m), spaces(m)
    model = bench.hmc_error_model(prob.n)
    rng = np.random.default_rng(3)
    K = np.exp(0.3 * rng.standard_normal((7, prob.n)))
    dev = DeviceErrorModel(model)
    e_dev, e_host = dev.predict(K), model.predict(K).astype(np.float64)
    assert np.max(np.abs(e_dev - e_host)) <= 1e-5 * np.max(np.abs(e_host))
    phi = pod_basis(Fin(V), r, n_snapshots=200, low=0.1, high=10.0, params="nine", seed=1)
    data = rng.uniform(0.2, 1.0, 9)
    ro = O.AffineROMOracle(prob, phi); ro.set_data(data)
    for projection in ("direct", "offline_online"):
        rom = AffineROMFin(V, model, phi, projection=projection); rom.set_data(data)
        assert rom._dev_model is not None
        res = rom.grad_romml_batch(K)
        rom.set_dl_model(model, device=False)               # the same evaluation with the network on the host
        ref = rom.grad_romml_batch(K)
        assert (res["info"] == 0).all()
        for s in range(len(K)):
            assert abs(res["loss"][s] - ref["loss"][s]) <= 2e-5 * abs(ref["loss"][s])
            assert np.linalg.norm(res["grad"][s] - ref["grad"][s]) <= 1e-5 * np.linalg.norm(ref["grad"][s])
        go, lo = O.grad_romml_oracle(ro, model, K[2])
        assert abs(res["loss"][2] - lo) <= 2e-5 * abs(lo) and np.linalg.norm(res["grad"][2] - go) <= 1e-5 * np.linalg.norm(go)


def test_hessian_action_matches_differences_of_the_device_gradient(spaces, fom_schedule):
    """Full Hessian action (four solves on the band sweep's stored factor, finrom_fom_solve_rhs) against central differences of the
    DEVICE adjoint gradient, and its Gauss-Newton part against `GN_hessian_action` (device Jacobian) when the data are reproduced
    exactly (zero residual)."""
    import warnings
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    fin = Fin(spaces(8))
    rng = np.random.default_rng(1)
    k = np.exp(0.3 * rng.standard_normal(fin.dofs)); u = rng.standard_normal(fin.dofs)
    d = rng.uniform(0.1, 1.0, fin.n_obs)
    if fom_schedule != "throughput schedule":          # no band plan in this fixture mode: the host fallback, announced
        with pytest.warns(RuntimeWarning, match="on the host"):
            H = fin.hessian_action(k, u, d)
        warnings.filterwarnings("ignore", category=RuntimeWarning, message="Fin.hessian_action")
    else:
        H = fin.hessian_action(k, u, d)
    eps = 1e-4
    fd = (fin.gradient(k + eps * u, d) - fin.gradient(k - eps * u, d)) / (2 * eps)
    assert np.linalg.norm(H - fd) < 1e-6 * np.linalg.norm(fd)
    q = fin.qoi_operator(fin.forward(k)[0])
    assert np.linalg.norm(fin.hessian_action(k, u, q) - fin.GN_hessian_action(k, u)) < 1e-8 * np.linalg.norm(H)


@pytest.mark.parametrize("r", [120, 200])
def test_wide_basis_fused_qoi_with_forty_observations(problems, spaces, r):
    """The QoI-only path of bases wider than 96 carries [B_r | (B_obs Phi)^T] as extra tile columns: 1 + n_obs = 41 columns are
    three tile columns (the default observation operator has nine: one).  Same numbers as the path that returns w_r."""
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    prob = problems(12); V = spaces(12)
    rom = AffineROMFin(V, None, oracle_basis(prob, r), external_obs=True)
    assert rom.n_obs == 40
    TH = np.random.default_rng(8).uniform(0.1, 3.5, (130, 9))
    full = rom.forward_nine_param_reduced_batch(TH)
    qonly = rom.forward_nine_param_reduced_batch(TH, want_w=False)
    assert (full["info"] == 0).all() and (qonly["info"] == 0).all()
    assert rel(qonly["qoi_r"], full["qoi_r"]) < TOL
    assert rel(full["qoi_r"], full["w_r"] @ rom.B_obs_phi.T) < TOL


@pytest.mark.parametrize("m,r", [(12, 81), (12, 120), (20, 200)])
def test_forty_point_observations_against_the_oracle(problems, spaces, m, r, fom_schedule):
    """The 40 point observations of `external_obs=True` (fom/forward_solve.py:215-228, rom/averaged_affine_ROM.py:192-212;
    SURVEY 8(d) cfg 4 quotes n_obs = 9 AND 40) against the oracle's own observation operator: the same boundary dofs, FOM and
    reduced observables <= 1e-10 on samples from full blocks and from the tail block, in the full form (w / w_r returned) and
    in the QoI-only forms -- for the FOM the full sweep with 40 rows of B_obs (two observations on one fin: the fins-as-functionals
    form does not apply and must not be picked), for bases wider than 96 the three-tile-column epilogue [B_r | (B_obs Phi)^T]."""
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    from bayesianinferencedl_amd.pairs import FinPairSolver
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    if m == 20 and fom_schedule != "throughput schedule":
        pytest.skip("m = 20 without the band plan is the interpreter's case: covered at m = 12")
    prob, V = problems(m), spaces(m)
    fo = O.FinOracle(prob, external_obs=True)
    assert fo.n_obs == 40 and len(np.unique(np.nonzero(fo.B_obs)[1])) > 30
    phi = oracle_basis(prob, r, n_snap=r + 40)
    ro = O.AffineROMOracle(prob, phi, B_obs=fo.B_obs)
    fin = Fin(V, external_obs=True)
    rom = AffineROMFin(V, None, phi, external_obs=True)
    assert fin.n_obs == rom.n_obs == 40
    assert np.array_equal(np.asarray(fin.B_obs), fo.B_obs) and np.array_equal(np.asarray(rom.B_obs), fo.B_obs)
    assert np.max(np.abs(rom.B_obs_phi - ro.B_obs_phi)) == 0.0
    S = 130                                                # two full blocks of 64 and a tail of two
    idx = np.array([0, 1, 31, 63, 64, 100, 127, 128, 129])
    K = np.exp(0.5 * np.random.default_rng(40).standard_normal((S, prob.n)))
    W = np.array([fo.forward(K[i]) for i in idx])
    Q = W @ fo.B_obs.T
    WR = np.array([ro.forward_reduced(K[i]) for i in idx])
    QR = WR @ ro.B_obs_phi.T
    # FOM: with w and without (the pair path's request)
    full = fin.forward_batch(K, want_w=True)
    qo = fin.forward_batch(K, want_w=False)
    eng = fin._engine("field")
    if fom_schedule == "throughput schedule":
        assert not eng.band_qoi_only, "40 point observations do not split by fins: the QoI-only sweep must not be installed"
        assert eng.last_path() in ("band_registers", "band_lds_4wave")
    assert (full["info"] == 0).all() and (qo["info"] == 0).all()
    assert rel(full["w"][idx], W) < TOL and rel(full["qoi"][idx], Q) < TOL and rel(qo["qoi"][idx], Q) < TOL
    assert np.array_equal(np.asarray(full["qoi"]), np.asarray(qo["qoi"]))
    # ROM: w_r returned / only the reduced QoI (bases wider than 96: fused_solve_mw with three extra tile columns)
    rw = rom.forward_reduced_batch(K)
    rq = rom.forward_reduced_batch(K, want_w=False)
    assert (rw["info"] == 0).all() and (rq["info"] == 0).all()
    assert rel(rw["qoi_r"][idx], QR) < TOL and rel(rq["qoi_r"][idx], QR) < TOL
    assert rel(rw["w_r"][idx] @ phi.T, WR @ phi.T) < TOL
    # the sample-pair call: both halves + the error, 40 columns each
    res = FinPairSolver(V, phi, True, "field", fin, rom).solve_pairs(K)
    assert res["qoi"].shape == res["qoi_r"].shape == res["err"].shape == (S, 40) and (res["info"] == 0).all()
    assert rel(res["qoi"][idx], Q) < TOL and rel(res["qoi_r"][idx], QR) < TOL
    assert np.max(np.abs(res["err"] - (res["qoi"] - res["qoi_r"]))) == 0.0
    # one-sample call patterns (batches <= 64 take other kernels): the scalar surface, and the per-fin parameter form
    z = fin.forward(K[129])[0]
    assert np.linalg.norm(fin.qoi_operator(z) - Q[-1]) < TOL * np.linalg.norm(Q[-1])
    assert np.linalg.norm(rom.qoi_reduced(rom.forward_reduced(K[129])) - QR[-1]) < TOL * np.linalg.norm(QR[-1])
    X9 = np.random.default_rng(41).uniform(0.1, 3.5, (70, 9))
    p9 = FinPairSolver(V, phi, True, "nine", fin, rom).solve_pairs(X9)
    for i in (0, 63, 64, 69):
        k = fo.nine_param_to_function(X9[i])
        q = fo.B_obs @ fo.forward(k); qr = ro.B_obs_phi @ ro.forward_reduced(k)
        assert np.linalg.norm(p9["qoi"][i] - q) < TOL * np.linalg.norm(q) and np.linalg.norm(p9["qoi_r"][i] - qr) < TOL * np.linalg.norm(qr)
