"""Committed golden vectors (tests/golden/fin_m4_r8.npz, oracle-generated: see make_golden.py).
CPU: the oracle still reproduces them (guards the oracle against drift).
GPU: the HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

from oracle import fin_oracle as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fin_m4_r8.npz"))
TOL = 1e-10


def rel(a, b):
    return np.max(np.linalg.norm(np.asarray(a) - b, axis=-1) / np.linalg.norm(b, axis=-1))


def test_oracle_reproduces_golden(problems):
    prob = problems(int(G["m"]))
    fo = O.FinOracle(prob); ro = O.AffineROMOracle(prob, G["phi"])
    assert rel([fo.forward(fo.five_param_to_function(x)) for x in G["k5"][:4]], G["w_five"][:4]) < 1e-12
    assert rel([ro.forward_reduced(x) @ ro.B_obs_phi.T for x in G["fields"][:4]], G["qoi_r_field"][:4]) < 1e-11
    assert rel(O.sample_fields(O.make_cov_chol(prob.coords, 'm52', 1.6), G["xi"]), G["fields"]) < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["small-batch schedule", "throughput schedule"])
@pytest.mark.parametrize("kind,key", [("five", "k5"), ("nine", "k9"), ("field", "fields")])
def test_gpu_reproduces_golden(spaces, kind, key, schedule, monkeypatch):
    from bayesianinferencedl_amd.pairs import FinPairSolver
    import bayesianinferencedl_amd.engine as E
    if schedule == "small-batch schedule":         # fom_small_kernel: the forward path where no band plan is installed
        monkeypatch.setattr(E, "USE_BAND", False)
    V = spaces(int(G["m"]))
    res = FinPairSolver(V, G["phi"], params=kind).solve_pairs(G[key], want_w=True, want_w_r=True)
    assert (res["info"] == 0).all()
    assert rel(res["w"], G[f"w_{kind}"]) < TOL
    assert rel(res["qoi"], G[f"qoi_{kind}"]) < TOL
    assert rel(res["qoi_r"], G[f"qoi_r_{kind}"]) < TOL
    assert rel(res["w_r"] @ G["phi"].T, G[f"w_r_{kind}"] @ G["phi"].T) < TOL
    assert rel(res["theta"], G[f"theta_{kind}"]) < 1e-13
    assert np.max(np.abs(res["err"] - (G[f"qoi_{kind}"] - G[f"qoi_r_{kind}"]))) < 1e-10


@pytest.mark.gpu
def test_gpu_sampler_reproduces_golden_fields(spaces):
    from bayesianinferencedl_amd.bayesian_inference.gaussian_field import make_cov_chol
    from bayesianinferencedl_amd.engine import FieldSampler
    V = spaces(int(G["m"]))
    assert rel(FieldSampler(make_cov_chol(V, length=1.6))(G["xi"]), G["fields"]) < 1e-12
