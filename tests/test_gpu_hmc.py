"""BASELINE configs[4] on the HIP path: HMC chains of sequential, dependent one-sample ROM + learned-error value-and-gradient
calls (reference bayesian_inference/pymc_func_bayes_inverse.py:92-104,148-167 -> rom/averaged_affine_ROM.py:358-396) at the
survey's sizes (m = 12, r = 81).  Every recorded evaluation -- its input is the product of the chain's own earlier gradients
-- is re-evaluated by the oracle's dense restatement."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import fin_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def setup(problems, spaces):
    sys.path.insert(0, ROOT)
    import bench
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    from bayesianinferencedl_amd.rom.basis import pod_basis
    m, r = 12, 81
    prob, V = problems(m), spaces(m)
    solver = Fin(V)
    phi = pod_basis(solver, r, n_snapshots=200, low=0.1, high=10.0, params="nine", seed=1)
    model = bench.hmc_error_model(V.dim())
    k_true = np.exp(0.25 * np.random.default_rng(11).standard_normal(V.dim()))
    data = solver.qoi_operator(solver.forward(k_true)[0])
    ro = O.AffineROMOracle(prob, phi); ro.set_data(data)
    return V, phi, model, data, ro


@pytest.mark.parametrize("projection", ["direct", "offline_online"])
def test_hmc_chains_value_and_gradient_match_the_oracle(setup, projection):
    from bayesianinferencedl_amd.bayesian_inference import hmc
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    V, phi, model, data, ro = setup
    rom = AffineROMFin(V, model, phi, projection=projection); rom.set_data(data)
    chains = [0, 1, 2, 3]
    K0 = np.stack([np.exp(0.1 * np.random.default_rng(6 + c).standard_normal(V.dim())) for c in chains])
    want = {0, 1, 10, 37, 80}
    res = hmc.run_chains(hmc.romml_value_and_grad(rom), K0, 81, seeds=[6 + c for c in chains], record=want)
    assert res.n_evals == 81 and res.proposals == 8 and len(res.recorded) == len(want)
    assert res.accept.sum() > 0, "no proposal accepted: the chains did not move"
    moved = np.linalg.norm(res.K - K0, axis=1)
    assert (moved > 0).any()
    for ev, K, loss, grad in res.recorded:
        for c in range(len(chains)):
            go, lo = O.grad_romml_oracle(ro, model, K[c])
            # value: fp64 ROM + fp32 network output (1e-7 absolute) entering a residual of order 0.1: ~5e-6 relative on the loss
            assert abs(loss[c] - lo) <= 2e-5 * abs(lo), (ev, c, loss[c], lo)
            assert np.linalg.norm(grad[c] - go) <= 1e-5 * np.linalg.norm(go), (ev, c)
    # a chain alone walks the same path as the same chain advanced in lockstep with others (fp32 network: GEMV vs GEMM order)
    solo = hmc.run_chains(hmc.romml_value_and_grad(rom), K0[1:2], 21, seeds=[7])
    lock = hmc.run_chains(hmc.romml_value_and_grad(rom), K0[:2], 21, seeds=[6, 7])
    assert np.linalg.norm(solo.K[0] - lock.K[1]) <= 1e-6 * np.linalg.norm(lock.K[1])


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("graph", [True, False])
def test_device_resident_chains_walk_the_host_chains_path(setup, graph, fused):
    """hmc.run_chains_device -- positions, momenta, gradients, the Metropolis test and the accept counters stay on the device, one
    captured HIP graph per PROPOSAL (graph=True) or the same launches in stream order -- must walk the path of the host recursion
    with the same seeds (the elementwise updates round differently: 1e-9), accept the same proposals, and its recorded
    evaluations must match the oracle like the host chain's do.  The proposals that hold a recorded evaluation (3 of 12) run step
    by step, the others through the graph.  eps large enough that proposals ARE rejected (a chain that accepts everything does
    not exercise the accept / reject bookkeeping)."""
    from bayesianinferencedl_amd.bayesian_inference import hmc
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    V, phi, model, data, ro = setup
    rom = AffineROMFin(V, model, phi); rom.set_data(data)
    chains = [0, 1, 2, 3]
    K0 = np.stack([np.exp(0.1 * np.random.default_rng(6 + c).standard_normal(V.dim())) for c in chains])
    kw = dict(seeds=[100 + c for c in chains], eps=3e-2, n_leapfrog=10)
    want = {0, 1, 10, 55, 120}
    host = hmc.run_chains(hmc.romml_value_and_grad(rom), K0, 121, record=want, **kw)
    # fused: the trajectory's arithmetic inside the library (finrom_hmc_begin / _leapfrog / _end, round 4); not fused: round 3's
    # form, finrom_romml_grad between torch elementwise kernels
    dev = hmc.run_chains_device(rom, K0, 121, record=want, graph=graph, fused=fused, **kw)
    assert dev.fused == fused
    assert dev.graph == graph, "HIP graph capture of the leapfrog step failed" if graph else "graph not requested"
    assert dev.n_evals == host.n_evals == 121 and dev.proposals == host.proposals == 12
    assert 0 < host.accept.sum() < 4 * 12, host.accept           # some accepted, some rejected
    assert np.array_equal(dev.accept, host.accept)
    assert np.linalg.norm(dev.K - host.K) <= 1e-9 * np.linalg.norm(host.K)
    assert [e for e, *_ in dev.recorded] == [e for e, *_ in host.recorded]
    for (ev, K, loss, grad), (_, Kh, lossh, gradh) in zip(dev.recorded, host.recorded):
        assert np.linalg.norm(K - Kh) <= 1e-9 * np.linalg.norm(Kh), ev
        assert np.linalg.norm(grad - gradh) <= 1e-6 * np.linalg.norm(gradh), ev
        go, lo = O.grad_romml_oracle(ro, model, K[2])
        assert abs(loss[2] - lo) <= 2e-5 * abs(lo) and np.linalg.norm(grad[2] - go) <= 1e-5 * np.linalg.norm(go), ev


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("graph,block", [(True, 32), (True, 5), (False, 7)])
def test_device_chains_without_host_round_trips_reproduce_the_host_chains(setup, graph, block, fused):
    """Nothing recorded: every proposal is one graph replay (or the same kernels in stream order) and the host sees the chain
    only at its end.  Random numbers drawn `block` proposals ahead and uploaded block by block (block = 5: blocks of 5, 5, 4;
    the device-side slice counter restarts per block): same accepted proposals, same end points, same per-proposal trace as the
    host recursion."""
    from bayesianinferencedl_amd.bayesian_inference import hmc
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    V, phi, model, data, ro = setup
    rom = AffineROMFin(V, model, phi); rom.set_data(data)
    chains = [0, 1, 2, 3]
    K0 = np.stack([np.exp(0.1 * np.random.default_rng(6 + c).standard_normal(V.dim())) for c in chains])
    kw = dict(seeds=[100 + c for c in chains], eps=3e-2, n_leapfrog=10, keep_trace=True)
    host = hmc.run_chains(hmc.romml_value_and_grad(rom), K0, 141, **kw)
    dev = hmc.run_chains_device(rom, K0, 141, graph=graph, block=block, fused=fused, **kw)
    assert dev.fused == fused and dev.graph == graph and dev.n_evals == host.n_evals == 141 and dev.proposals == host.proposals == 14
    assert 0 < host.accept.sum() < 4 * 14, host.accept
    assert np.array_equal(dev.accept, host.accept)
    assert dev.trace.shape == host.trace.shape == (15, 4, V.dim())
    assert np.max(np.abs(dev.trace - host.trace)) <= 1e-9 * np.max(np.abs(host.trace))
    assert np.linalg.norm(dev.K - host.K) <= 1e-9 * np.linalg.norm(host.K)


def test_bench_hmc_mode_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "hmc", "--steps", "40", "--warmup", "10",
                        "--cpu-samples", "4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["unit"] == "evals/s" and d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["chains"] == 4
    assert d["config"]["evals_per_chain"] == 41 and d["config"]["r"] == 81
    assert d["single_chain_latency_ms_per_call"] > 0 and d["cpu_baseline"]["kind"] == "port"
    assert d["roofline"] is None and d["critical_path"]["bound"] == "latency" and d["critical_path"]["sum_us"] > 0
    assert d["config"]["fused_leapfrog"] is True and d["config"]["hip_graph"] is True and d["chains_gathered"] == 4 and len(d["chains_sha256"]) == 64
    assert abs(d["value"] - 4 * 41 / (d["ms_per_step"] * 41e-3)) < 1e-6 * d["value"]
