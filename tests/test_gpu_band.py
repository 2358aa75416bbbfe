"""The frontal band sweep (csrc/fom_band.hip, front in registers) against the oracle and against the schedule interpreter it
replaces on the throughput path: w and QoI <= 1e-10 relative for nodal fields and for five / nine fin conductivities, on
every mesh the library has window sizes for, failure flags for indefinite operators, batch tails."""
import os

import numpy as np
import pytest

from oracle import fin_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _rel(a, b):
    return np.max(np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1))


@pytest.mark.parametrize("m", [4, 8, 12, 16, 20])      # 16, 20: the variant with the post's window in LDS
def test_band_sweep_matches_oracle_and_interpreter(problems, spaces, m):
    import bayesianinferencedl_amd.engine as E
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    prob, V = problems(m), spaces(m)
    fo = O.FinOracle(prob)
    rng = np.random.default_rng(10 + m)
    S = 700                                               # beyond the small-batch schedule; not a multiple of 64
    fin = Fin(V)
    old = E.USE_BAND
    try:
        E.USE_BAND = False
        fin_i = Fin(V)                                    # interpreter-only engines (created lazily: force them now)
        for params in ("field", "nine", "five"):
            assert fin_i._engine(params).band is None
    finally:
        E.USE_BAND = old
    for params, dim in (("field", prob.n), ("nine", 9), ("five", 5)):
        X = np.exp(0.5 * rng.standard_normal((S, dim))) if params == "field" else rng.uniform(0.1, 10.0, (S, dim))
        eng = fin._engine(params)
        assert eng.band is not None, "band sweep not installed"
        res = fin.forward_batch(X, want_w=True, params=None if params == "field" else params)
        ref = fin_i.forward_batch(X, want_w=True, params=None if params == "field" else params)
        assert fin_i._engine(params).band is None
        assert (res["info"] == 0).all()
        assert _rel(res["w"], ref["w"]) < TOL and _rel(res["qoi"], ref["qoi"]) < TOL
        lift = {"field": lambda x: x, "nine": fo.nine_param_to_function, "five": fo.five_param_to_function}[params]
        for s in (0, 1, 63, 64, S - 1):
            w = fo.forward(lift(X[s]))
            assert np.linalg.norm(res["w"][s] - w) < TOL * np.linalg.norm(w)
            q = fo.qoi_operator(w)
            assert np.linalg.norm(res["qoi"][s] - q) < TOL * np.linalg.norm(q)


@pytest.mark.parametrize("m", [12, 20])
def test_band_sweep_flags_indefinite_samples(spaces, m):
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    V = spaces(m)
    fin = Fin(V)
    rng = np.random.default_rng(3)
    X = rng.uniform(0.5, 2.0, (600, 9))
    X[17] = -X[17]; X[599, 4] = -5.0
    res = fin.forward_batch(X, want_w=True, params="nine")
    assert fin._engine("nine").band is not None
    bad = np.nonzero(res["info"])[0].tolist()
    assert bad == [17, 599]
    assert np.isnan(res["qoi"][17]).all() and np.isnan(res["w"][599]).all()
    good = np.setdiff1d(np.arange(600), bad)
    assert np.isfinite(res["qoi"][good]).all() and np.isfinite(res["w"][good]).all()
