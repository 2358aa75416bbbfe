"""Offline/online form of the reduced operator (finrom_rom_set_gram / finrom_rom_set_projection): A_r assembled from
precomputed blocks G_pq instead of the per-sample psi^T psi contraction.  Same contract as the direct form, checked against
the same oracle restatement of rom/averaged_affine_ROM.py:278-310 (which contracts directly, like the reference)."""
import numpy as np
import pytest

from oracle import fin_oracle as O
from test_gpu_parity import oracle_basis, rel, TOL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,r", [(4, 8), (12, 80), (12, 81), (12, 33), (12, 120), (12, 200), (4, 150)])
def test_rom_parity_offline_online(problems, spaces, m, r):
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, r)
    ro = O.AffineROMOracle(prob, phi)
    rng = np.random.default_rng(4)
    TH = rng.uniform(0.1, 3.5, (70, 9))          # 70: a ragged last group for the 32-sample store kernel and the 4-wave one
    rom = AffineROMFin(V, None, phi, projection="offline_online")
    assert rom.projection == "offline_online" and rom._rom.gram_pairs <= 55
    res = rom.forward_nine_param_reduced_batch(TH, want_state=True)
    assert (res["info"] == 0).all()
    n_check = 12
    idx = list(range(n_check - 2)) + [68, 69]
    WR, AR, BR = [], [], []
    for i in idx:
        w_r, A_r, B_r, _ = ro.forward_nine_param_reduced(TH[i], return_parts=True)
        WR.append(w_r); AR.append(A_r); BR.append(B_r)
    WR = np.array(WR); AR = np.array(AR); BR = np.array(BR)
    assert rel(res["A_r"][idx].reshape(n_check, -1), AR.reshape(n_check, -1)) < 1e-12
    assert rel(res["B_r"][idx], BR) < 1e-12
    assert rel(res["qoi_r"][idx], WR @ ro.B_obs_phi.T) < TOL
    assert rel(res["w_r"][idx] @ phi.T, WR @ phi.T) < TOL
    fused = rom.forward_nine_param_reduced_batch(TH)          # factorisation (and for r <= 80 the solve) in registers
    assert (fused["info"] == 0).all()
    assert rel(fused["qoi_r"][idx], WR @ ro.B_obs_phi.T) < TOL
    assert rel(fused["w_r"][idx] @ phi.T, WR @ phi.T) < TOL
    # switching back gives the direct contraction again, same answers
    rom.set_projection("direct")
    direct = rom.forward_nine_param_reduced_batch(TH)
    assert rel(fused["qoi_r"], direct["qoi_r"]) < TOL and rel(fused["w_r"] @ phi.T, direct["w_r"] @ phi.T) < TOL


@pytest.mark.parametrize("m,r", [(4, 8), (12, 80), (12, 120)])
def test_rom_gradient_offline_online(problems, spaces, m, r):
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, r)
    ro = O.AffineROMOracle(prob, phi)
    rom = AffineROMFin(V, None, phi, projection="offline_online")
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


@pytest.mark.parametrize("params,S", [("five", 300), ("field", 40), ("five", 1000)])
def test_pairs_offline_online(problems, spaces, params, S):
    """The dataset loop body (generate_fin_dataset.py:83-100) with the offline/online ROM half, against the oracle."""
    from bayesianinferencedl_amd.pairs import FinPairSolver
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    m = 12 if params == "five" else 4
    prob = problems(m); V = spaces(m)
    phi = oracle_basis(prob, 80 if m == 12 else 8)
    rng = np.random.default_rng(12)
    X = rng.uniform(0.1, 1.0, (S, 5)) if params == "five" else np.exp(0.4 * rng.standard_normal((S, prob.n)))
    ps = FinPairSolver(V, phi, params=params, solver_r=AffineROMFin(V, None, phi, projection="offline_online"))
    res = ps.solve_pairs(X)
    assert (res["info"] == 0).all()
    fo = O.FinOracle(prob); ro = O.AffineROMOracle(prob, phi)
    for i in [0, 1, S // 2, S - 1]:
        k = fo.five_param_to_function(X[i]) if params == "five" else X[i]
        q = fo.qoi_operator(fo.forward(k))
        qr = ro.qoi_reduced(ro.forward_reduced(k))
        assert rel(res["qoi"][i][None], q[None]) < TOL
        assert rel(res["qoi_r"][i][None], qr[None]) < TOL
        assert np.max(np.abs(res["err"][i] - (q - qr))) < 1e-10 * np.max(np.abs(q))


def test_gram_descriptor_errors(problems, spaces):
    """Misuse of the offline/online entry points comes back as an error status, not a crash."""
    from bayesianinferencedl_amd._ffi import FinromError
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    prob = problems(4); V = spaces(4)
    phi = oracle_basis(prob, 8)
    rom = AffineROMFin(V, None, phi)
    with pytest.raises(FinromError, match="set_gram has not been called"):
        rom._rom.set_projection("offline_online")
    G = np.eye(8)[None]
    with pytest.raises(FinromError, match="invalid pair"):
        rom._rom.set_gram_blocks([(3, 2)], G)
    with pytest.raises(FinromError, match="invalid pair"):
        rom._rom.set_gram_blocks([(0, 10)], G)
    with pytest.raises(FinromError, match="duplicate"):
        rom._rom.set_gram_blocks([(1, 2), (1, 2)], np.concatenate([G, G]))
    bad = np.eye(8); bad[0, 3] = 1.0
    with pytest.raises(FinromError, match="not symmetric"):
        rom._rom.set_gram_blocks([(0, 0)], bad[None])
    with pytest.raises(FinromError, match="more than 64"):
        rom._rom.set_gram_blocks([(0, 0)] * 65, np.repeat(G, 65, 0))
    with pytest.raises(ValueError):
        rom.set_projection("galerkin")
    # and the handle still works afterwards
    assert rom.forward_nine_param_reduced_batch(np.ones((3, 9)))["info"].tolist() == [0, 0, 0]
