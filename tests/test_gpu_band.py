"""The frontal band sweep (csrc/fom_band.hip) against the oracle and against the schedule interpreter it replaces on the
throughput path: w and QoI <= 1e-10 relative for nodal fields and for five / nine fin conductivities, on every mesh the library
has window sizes for -- the front in registers (m = 4, 8, 12: fom_band_kernel) and the post's window over four waves with LDS as
the exchange (m = 16, 20, 24, 28: fom_band_ldsw_kernel; from m = 24 on some of the extras' rows live in the workspace), each in its full form (w wanted) and in its QoI-only form -- failure flags for
indefinite operators, batch tails.

Every case ASSERTS WHICH KERNEL RAN (finrom_fom_last_path): the small-batch schedule takes batches of <= 512 samples (<= 4096
when the value vector does not fit LDS, i.e. m >= 16), so a test that only sizes its batch "large" can silently compare
fom_small_kernel with itself -- round 2's m = 16 / 20 cases did.  Here the threshold is moved out of the way on the handle
(finrom_fom_set_small_max) and the path is checked after every call."""
import numpy as np
import pytest

from oracle import fin_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-10
BAND_PATH = {4: "band_registers", 8: "band_registers", 12: "band_registers", 16: "band_lds_4wave", 20: "band_lds_4wave",
             24: "band_lds_4wave", 28: "band_lds_4wave"}


def _rel(a, b):
    return np.max(np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1))


def _throughput_engines(V, kinds=("field", "nine", "five")):
    """(Fin on the band sweep, Fin on the interpreter), both with the small-batch schedule out of the way."""
    import bayesianinferencedl_amd.engine as E
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    fin = Fin(V)
    old = E.USE_BAND
    try:
        E.USE_BAND = False
        fin_i = Fin(V)                                    # interpreter-only engines (created lazily: force them now)
        for params in kinds:
            assert fin_i._engine(params).band is None
            fin_i._engine(params).set_small_max(0)
    finally:
        E.USE_BAND = old
    for params in kinds:
        eng = fin._engine(params)
        assert eng.band is not None, "band sweep not installed"
        assert eng.last_path() == "none"
        eng.set_small_max(0)
    return fin, fin_i


@pytest.mark.parametrize("m", [4, 8, 12, 16, 20, 24, 28])
def test_band_sweep_matches_oracle_and_interpreter(problems, spaces, m):
    prob, V = problems(m), spaces(m)
    fo = O.FinOracle(prob)
    rng = np.random.default_rng(10 + m)
    S = 700                                               # 11 blocks of 64 lanes, the last one with 60 live lanes
    fin, fin_i = _throughput_engines(V)
    # samples checked against the oracle: first / last lane of a block, mid lanes of several blocks, the tail block
    picks = (0, 1, 63, 64, 130, 257, 389, 511, 640, 699)
    for params, dim in (("field", prob.n), ("nine", 9), ("five", 5)):
        X = np.exp(0.5 * rng.standard_normal((S, dim))) if params == "field" else rng.uniform(0.1, 10.0, (S, dim))
        kw = dict(want_w=True, params=None if params == "field" else params)
        res = fin.forward_batch(X, **kw)
        assert fin._engine(params).last_path() == BAND_PATH[m], fin._engine(params).last_path()
        ref = fin_i.forward_batch(X, **kw)
        assert fin_i._engine(params).last_path() == "interpreter"
        assert (res["info"] == 0).all() and (ref["info"] == 0).all()
        assert _rel(res["w"], ref["w"]) < TOL and _rel(res["qoi"], ref["qoi"]) < TOL
        lift = {"field": lambda x: x, "nine": fo.nine_param_to_function, "five": fo.five_param_to_function}[params]
        for s in picks:
            w = fo.forward(lift(X[s]))
            assert np.linalg.norm(res["w"][s] - w) < TOL * np.linalg.norm(w), (params, s)
            q = fo.qoi_operator(w)
            assert np.linalg.norm(res["qoi"][s] - q) < TOL * np.linalg.norm(q), (params, s)
        # QoI-only calls (the sample-pair path asks for no w) take the kernel's QoI-only form: the fins ride through their
        # forward sweeps as functionals of the interface values -- no factor, y or backward sweep for them (fom_band.hip)
        res_q = fin.forward_batch(X, want_w=False, params=kw["params"])
        assert fin._engine(params).last_path() == BAND_PATH[m] + "_qoi"
        assert (res_q["info"] == 0).all()
        assert _rel(res_q["qoi"], res["qoi"]) < 1e-11
        for s in picks:
            q = fo.qoi_operator(fo.forward(lift(X[s])))
            assert np.linalg.norm(res_q["qoi"][s] - q) < TOL * np.linalg.norm(q), (params, s, "qoi-only")


@pytest.mark.parametrize("m", [12, 16, 20, 24, 28])
def test_band_sweep_flags_indefinite_samples(spaces, m):
    """A negative conductivity in every fin (the failure is seen by a FIN's sweep -- in the four-wave kernel by whichever wave
    swept that fin) and one in the centre post only (seen by the post's sweep): both samples are flagged and NaN, nobody else
    is, including their neighbours in the same wave; the second one sits in the tail block."""
    fin, _ = _throughput_engines(spaces(m), kinds=("nine",))
    rng = np.random.default_rng(3)
    X = rng.uniform(0.5, 2.0, (600, 9))
    X[17] = -X[17]; X[599, 4] = -5.0
    X[130, 7] = -3.0                                      # one fin only: in the four-wave kernel ONE wave sees this failure
    res = fin.forward_batch(X, want_w=True, params="nine")
    assert fin._engine("nine").last_path() == BAND_PATH[m]
    bad = np.nonzero(res["info"])[0].tolist()
    assert bad == [17, 130, 599]
    for s in bad:
        assert np.isnan(res["qoi"][s]).all() and np.isnan(res["w"][s]).all()
    good = np.setdiff1d(np.arange(600), bad)
    assert np.isfinite(res["qoi"][good]).all() and np.isfinite(res["w"][good]).all()
    # the same through the QoI-only form
    res_q = fin.forward_batch(X, want_w=False, params="nine")
    assert fin._engine("nine").last_path() == BAND_PATH[m] + "_qoi"
    assert np.nonzero(res_q["info"])[0].tolist() == bad
    assert np.isnan(res_q["qoi"][bad]).all() and np.isfinite(res_q["qoi"][good]).all()
    assert _rel(res_q["qoi"][good], res["qoi"][good]) < 1e-11


@pytest.mark.parametrize("m,small_path,small_max", [(12, "small_lds", 512), (20, "small_global", 4096)])
def test_dispatch_by_batch_size(spaces, m, small_path, small_max, monkeypatch):
    """The dispatch, asserted.  FORWARD solves take the band sweep at every batch size once the plan is installed (a lone wave of
    it beats the level-scheduled small-batch kernel: 1.7 against 2.7 ms at m = 12, 4.2 against 17.9 ms at m = 20); the GRADIENT of
    a small batch takes the small-batch kernel (value vector in LDS at m = 12, in the workspace at m = 20), of a larger one the
    band adjoint.  Without a band plan the small-batch kernel is the forward path too.  Results agree to round-off."""
    import bayesianinferencedl_amd.engine as E
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    fin = Fin(spaces(m))
    eng = fin._engine("nine")
    rng = np.random.default_rng(8)
    X = rng.uniform(0.1, 10.0, (small_max + 1, 9))
    one = fin.forward_batch(X[:1], want_w=True, params="nine")
    assert eng.last_path() == BAND_PATH[m]
    a = fin.forward_batch(X[:64], want_w=False, params="nine")
    assert eng.last_path() == BAND_PATH[m] + "_qoi"
    b = fin.forward_batch(X, want_w=False, params="nine")
    assert eng.last_path() == BAND_PATH[m] + "_qoi"
    assert np.array_equal(a["qoi"], b["qoi"][:64]) and _rel(one["qoi"], b["qoi"][:1]) < 1e-11     # same kernel: same numbers
    data = np.full(9, 0.3)
    n_small = 40
    gs = fin.gradient_batch(X[:n_small], data, params="nine")
    assert eng.last_path() == small_path
    gl = fin.gradient_batch(X, data, params="nine")
    assert eng.last_path() == BAND_PATH[m]
    assert _rel(gs["grad"], gl["grad"][:n_small]) < 1e-9
    monkeypatch.setattr(E, "USE_BAND", False)
    fin_nb = Fin(spaces(m))
    c = fin_nb.forward_batch(X[:64], want_w=False, params="nine")
    assert fin_nb._engine("nine").last_path() == small_path
    # two schedules of the same factorisation (other elimination order, other summation order): round-off times the operator's
    # condition number (kappa in [0.1, 10]: 2e-12 measured at m = 20), an order below the parity tolerance
    assert _rel(c["qoi"], a["qoi"]) < 1e-11


@pytest.mark.parametrize("m", [12, 16, 20, 24, 28])
def test_adjoint_gradient_on_the_band_layout(problems, spaces, m):
    """finrom_fom_gradient for batches beyond the small-batch schedule (Fin.gradient, fom/forward_solve.py:293-322): the full
    band sweep leaves the factor and w in the workspace, fom_band_adjoint_kernel solves the adjoint with the stored columns
    (forward substitution + one more backward sweep) and contracts the gradient there -- against the oracle, against the
    interpreter's stored-factor path on every sample, with shared and per-sample data, batch tail, a flagged sample."""
    prob, V = problems(m), spaces(m)
    fo = O.FinOracle(prob)
    rng = np.random.default_rng(50 + m)
    S = 200                                               # 3 blocks + a tail of 8 lanes
    fin, fin_i = _throughput_engines(V)
    data = rng.uniform(0.1, 0.6, 9)
    for params, dim in (("field", prob.n), ("nine", 9), ("five", 5)):
        X = np.exp(0.3 * rng.standard_normal((S, dim))) if params == "field" else rng.uniform(0.3, 5.0, (S, dim))
        kw = dict(params=None if params == "field" else params)
        res = fin.gradient_batch(X, data, **kw)
        assert fin._engine(params).last_path() == BAND_PATH[m]
        ref = fin_i.gradient_batch(X, data, **kw)
        assert fin_i._engine(params).last_path() == "interpreter"
        assert (res["info"] == 0).all()
        assert _rel(res["grad"], ref["grad"]) < 1e-9 and np.max(np.abs(res["J"] - ref["J"]) / ref["J"]) < 1e-10
        lift = {"field": lambda x: x, "nine": fo.nine_param_to_function, "five": fo.five_param_to_function}[params]
        ops = V.operators()
        chain = {"field": None, "nine": ops.N9, "five": ops.N9 @ ops.E59}[params]
        for s in (0, 63, 64, S - 1):
            g = fo.gradient(lift(X[s]), data)
            g = g if chain is None else g @ chain
            assert np.linalg.norm(res["grad"][s] - g) < 1e-9 * np.linalg.norm(g), (params, s)
            J = 0.5 * np.sum((fo.qoi_operator(fo.forward(lift(X[s]))) - data) ** 2)
            assert abs(res["J"][s] - J) < 1e-10 * J
    # per-sample data and a sample whose operator is indefinite
    X = rng.uniform(0.3, 5.0, (S, 9))
    X[77, 2] = -4.0
    D2 = rng.uniform(0.1, 0.6, (S, 9))
    res = fin.gradient_batch(X, D2, params="nine")
    assert fin._engine("nine").last_path() == BAND_PATH[m]
    assert np.nonzero(res["info"])[0].tolist() == [77] and np.isnan(res["grad"][77]).all()
    good = np.setdiff1d(np.arange(S), [77])
    ref = fin_i.gradient_batch(X[good], D2[good], params="nine")
    assert _rel(res["grad"][good], ref["grad"]) < 1e-9


@pytest.mark.parametrize("m", [12, 20, 24])
def test_general_right_hand_sides_on_the_stored_factor(spaces, m):
    """finrom_fom_solve_rhs (the incremental solves of Fin.hessian_action, fom/forward_solve.py:344-368): out = A(x_s)^-1 rhs for
    three right-hand sides per sample -- dense random ones, so every fin's contribution to the post's right-hand side and every
    extra's collected right-hand side is exercised -- against SciPy's sparse LU, batch tail included; then the batched Hessian
    action against central differences of the device gradient."""
    import scipy.sparse.linalg as spl
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    V = spaces(m)
    fin = Fin(V)
    ops = V.operators()
    eng = fin._engine("field")
    rng = np.random.default_rng(60 + m)
    S, nrhs = 70, 3
    K = np.exp(0.4 * rng.standard_normal((S, ops.n)))
    R = rng.standard_normal((S, nrhs, ops.n))
    R[:, 2, :] = 0.0; R[:, 2, rng.integers(0, ops.n, 5)] = 1.0          # a sparse one too
    res = eng.solve_rhs(K, R)
    assert eng.last_path() == BAND_PATH[m] and (res["info"] == 0).all()
    for s in (0, 1, 63, 64, S - 1):
        lu = spl.splu(ops.csr(ops.robin_vals + ops.W_field @ K[s]).tocsc())
        for k in range(nrhs):
            ref = lu.solve(R[s, k])
            assert np.linalg.norm(res["out"][s, k] - ref) < 1e-10 * np.linalg.norm(ref), (s, k)
    # Hessian action: symmetric, and the derivative of the device gradient
    U = rng.standard_normal((4, ops.n)); d = rng.uniform(0.1, 1.0, fin.n_obs)
    H = fin.hessian_action_batch(K[:4], U, d)
    eps = 1e-4
    for s in range(2):
        fd = (fin.gradient(K[s] + eps * U[s], d) - fin.gradient(K[s] - eps * U[s], d)) / (2 * eps)
        assert np.linalg.norm(H[s] - fd) < 1e-6 * np.linalg.norm(fd)
    H01 = fin.hessian_action(K[0], U[1], d)
    assert abs(U[1] @ H[0] - U[0] @ H01) < 1e-9 * np.linalg.norm(H[0]) * np.linalg.norm(U[1])
