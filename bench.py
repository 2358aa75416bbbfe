#!/usr/bin/env python3
"""bench.py -- FOM+ROM forward-solve sample pairs / second on MI355X (BASELINE.json metric).

A *step* is one pass of the hot path (the body of deep_learning/generate_fin_dataset.py:83-100,
batched) over one batch of synthetic conductivity samples already resident in HBM:
    FOM sparse solve + QoI  ->  sub-fin averages  ->  LSPG ROM solve + QoI  ->  error.
Workload (BASELINE.json configs[1]): five-parameter fin, lattice mesh m=12 (n = 1597 DoF),
orthonormal POD basis r = 80, 100 000 samples per GPU, fp64.  With --gpus N > 1 every rank
runs its own 100k-sample shard (weak scaling, no data-path collective) and the QoIs are
gathered on rank 0 with one RCCL all_gather per step (the "gather at the end" of north_star).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel, measured live with HIP events on the launch stream
  cpu_baseline  the NumPy/SciPy oracle (oracle/fin_oracle.py, kind="port") timed on one host core
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X dense fp64 matrix peak (vendor; = 256 CU * 4 SIMD * 32 FLOP/clk * 2.4 GHz)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=100_000, help="samples per GPU per step")
    ap.add_argument("--m", type=int, default=12, help="lattice divisor (12 -> 1597 DoF)")
    ap.add_argument("--r", type=int, default=80)
    ap.add_argument("--params", default="five", choices=["five", "nine", "field"])
    ap.add_argument("--cpu-samples", type=int, default=2000, help="oracle samples for cpu_baseline (0 = skip)")
    ap.add_argument("--projection", default="direct", choices=["direct", "offline_online"],
                    help="how the timed region forms A_r: 'direct' = per-sample psi^T psi on MFMA (what the reference executes, the "
                         "headline); 'offline_online' = precomputed Gram blocks (same results, ~50x fewer ROM flops)")
    ap.add_argument("--no-other", action="store_true", help="skip the extra (untimed) pass with the other --projection")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-host-io", action="store_true", help="skip the extra (untimed) host-buffer pass")
    return ap.parse_args()


def flops_per_pair(ops, plan, rom, n_obs, P):
    """SURVEY 8(d) formula, direct LSPG with the symmetric half of psi^T psi."""
    n, r = ops.n, rom.n_r
    cc = np.diff(plan.col_ptr).astype(np.int64) + 1
    return {
        "assembly": 2 * (P + 1) * ops.nnz,
        "cholesky": int((cc * cc).sum()),
        "trisolves": 4 * plan.nnzL,
        "psi": 2 * ops.nnz * r,
        "syrk_sym": n * r * (r + 1),
        "rhs": 2 * n * r,
        "reduced_solve": r ** 3 // 3 + 2 * r * r,
        "qoi": 2 * n_obs * (n + r),
    }


# ---- all-core CPU baseline: worker processes, started BEFORE anything touches the GPU ----------------
_W = {}


def _w_init(m, params, phi):
    os.environ["OMP_NUM_THREADS"] = "1"
    try:
        from threadpoolctl import threadpool_limits
        _W["lim"] = threadpool_limits(limits=1)
    except Exception:
        pass
    from oracle import fin_oracle as O
    prob = O.FinProblem(m)
    _W["fo"], _W["ro"] = O.FinOracle(prob), O.AffineROMOracle(prob, phi)
    _W["lift"] = _W["fo"].five_param_to_function if params == "five" else _W["fo"].nine_param_to_function


def _w_run(X):
    fo, ro, lift = _W["fo"], _W["ro"], _W["lift"]
    acc = 0.0
    for x in X:
        k = lift(x)
        acc += float(fo.qoi_operator(fo.forward(k))[0] + ro.qoi_reduced(ro.forward_reduced(k))[0])
    return len(X), acc


def cpu_baseline_all_cores(args):
    """The same one-sample-at-a-time oracle loop in os.cpu_count() worker processes over sample shards (one thread each)."""
    import multiprocessing as mp
    from oracle import fin_oracle as O
    n = max(1, min(os.cpu_count() or 1, 16))          # the one-GPU box grants 16 CPUs
    prob = O.FinProblem(args.m)
    fo = O.FinOracle(prob)
    rng = np.random.default_rng(1)
    lift, dim = (fo.five_param_to_function, 5) if args.params == "five" else (fo.nine_param_to_function, 9)
    Y = np.array([fo.forward(lift(rng.uniform(0.1, 3.5, dim))) for _ in range(max(2 * args.r, 100))])
    phi = O.pod_basis(Y, args.r)
    per = max(50, args.cpu_samples // 2)
    X = np.random.default_rng(3).uniform(0.1, 10.0, (n * per, dim))
    with mp.get_context("spawn").Pool(n, initializer=_w_init, initargs=(args.m, args.params, phi)) as pool:
        pool.map(_w_run, [X[i:i + 2] for i in range(0, 2 * n, 2)])                  # every worker built its operators
        t0 = time.perf_counter()
        done = sum(c for c, _ in pool.map(_w_run, [X[i * per:(i + 1) * per] for i in range(n)], chunksize=1))
        dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "pairs/s", "cores": n, "kind": "port",
            "sample": f"{done} samples of the same distribution in {n} worker processes x 1 thread, oracle/fin_oracle.py loop"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_all = None
    if world == 1 and args.cpu_samples > 0 and args.params != "field":
        try:
            cpu_all = cpu_baseline_all_cores(args)        # worker processes must be gone before the GPU is initialised
        except Exception as e:                             # context only: never fail the bench for it
            cpu_all = {"error": repr(e)}
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from bayesianinferencedl_amd import _ffi
    from bayesianinferencedl_amd.fom.thermal_fin import get_space
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    from bayesianinferencedl_amd.rom.basis import pod_basis
    from bayesianinferencedl_amd.pairs import FinPairSolver
    _ffi.check(_ffi.lib().finrom_set_device(local_rank))

    V = get_space(None, m=args.m)
    solver = Fin(V)
    bparams = "five" if args.params == "five" else "nine"
    phi = pod_basis(solver, args.r, n_snapshots=400, low=0.1, high=10.0, params=bparams, seed=1)
    solver_r = AffineROMFin(V, None, phi, projection=args.projection)
    pairs = FinPairSolver(V, phi, False, args.params, solver, solver_r)

    S = args.samples
    # inputs are keyed by the GLOBAL sample index, so rank g's shard is rows [g*S, (g+1)*S) of
    # the same stream whatever the GPU count (results independent of G)
    rng = np.random.default_rng(3)
    if args.params == "field":
        from bayesianinferencedl_amd.bayesian_inference.gaussian_field import make_cov_chol
        from bayesianinferencedl_amd.engine import FieldSampler
        rng = np.random.default_rng(3 + rank)
        xi = torch.from_numpy(rng.standard_normal((S, V.dim()))).to(dev)
        X = FieldSampler(make_cov_chol(V, length=1.6))(xi)
        del xi
    else:
        Xg = rng.uniform(0.1, 10.0, (world * S, pairs.xdim))
        X = torch.from_numpy(Xg[rank * S:(rank + 1) * S]).to(dev)
        del Xg
    from bayesianinferencedl_amd.distributed import gather_rows

    def step():
        res = pairs.solve_pairs(X)
        if world > 1:   # the one exchange step: QoI pairs of every shard (RCCL all_gather over xGMI)
            res["gathered"] = gather_rows(torch.cat([res["qoi"], res["qoi_r"]], dim=1), world)
        return res

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    fence()
    L = _ffi.lib()
    # context only (NOT the timed region): one pass with the two halves serialised on one stream gives
    # each kernel's stand-alone duration; in the timed region the FOM and ROM kernels overlap
    serial_ms = None
    if not args.no_profile:
        L.finrom_profile_reset(); L.finrom_profile_enable(1)
        L.finrom_set_overlap(0)
        step(); fence()
        L.finrom_set_overlap(1)
        L.finrom_profile_enable(0)
        serial_ms = {k: round(v[1] / v[0], 4) for k, v in _ffi.profile_read().items() if v[0]}
    # context only: the same step fed from / returned to HOST memory through the library's own copies (the NumPy-facing
    # boundary; parameters in, QoI pairs + errors + w_r + theta + info out), i.e. the PCIe-inclusive rate.  Never `value`.
    host_io = None
    if not args.no_profile and not args.no_host_io and world == 1:
        Xh = X.cpu().numpy()
        pairs.solve_pairs(Xh)
        t0 = time.perf_counter()
        pairs.solve_pairs(Xh)
        host_io = S / (time.perf_counter() - t0)
        del Xh
    L.finrom_profile_reset()
    L.finrom_profile_enable(0 if args.no_profile else 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    dt = time.perf_counter() - t0
    L.finrom_profile_enable(0)
    prof = _ffi.profile_read()
    n_bad = int((res["info"] != 0).sum().item())

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # context only, after the timed region: the same steps with the OTHER form of the reduced operator (see --projection)
    other = None
    if not args.no_profile and not args.no_other:
        other_mode = "offline_online" if args.projection == "direct" else "direct"
        solver_r.set_projection(other_mode)
        step(); fence()
        L.finrom_profile_reset(); L.finrom_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res_o = step()
        fence()
        dto = time.perf_counter() - t0
        L.finrom_profile_enable(0)
        tmax = torch.tensor([dto], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dto = float(tmax.item())
        dq = float((torch.linalg.norm(res_o["qoi_r"] - res["qoi_r"], dim=1) / torch.linalg.norm(res["qoi_r"], dim=1)).max().item())
        other = {"projection": other_mode, "value": world * S * args.steps / dto, "unit": "pairs/s", "ms_per_step": 1e3 * dto / args.steps,
                 "kernels_avg_ms": {k: round(v[1] / v[0], 4) for k, v in _ffi.profile_read().items() if v[0]},
                 "max_rel_diff_qoi_r_vs_timed_region": dq,
                 "note": "same inputs and outputs as the timed region; A_r assembled from precomputed blocks G_pq = Psi_p^T Psi_q "
                         "instead of the per-sample psi^T psi contraction" if other_mode == "offline_online" else
                         "per-sample psi^T psi contraction on fp64 MFMA (what the reference executes)"}
        solver_r.set_projection(args.projection)

    if rank == 0:
        ops, plan = V.operators(), solver._plan
        fl = flops_per_pair(ops, plan, solver_r, pairs.n_obs, pairs.xdim)
        total_pairs = world * S * args.steps
        ms = {k: (v[1] / v[0] if v[0] else 0.0) for k, v in prof.items()}       # avg ms per launch
        # dominant kernel by measured time
        cand = {k: v[1] for k, v in prof.items()}
        dom = max(cand, key=cand.get) if any(cand.values()) else "rom_proj_mfma"
        roof = None
        traffic = measured_traffic(dom, f"{args.params}/m{args.m}/r{args.r}/S{S}")
        if args.projection == "offline_online":
            npairs = solver_r._rom.gram_pairs
            fl["syrk_sym"] = npairs * args.r * (args.r + 1)          # multiply-adds of the block sum, symmetric half
            fl["psi"] = 0
            fl["rhs"] = 2 * 10 * args.r
            if dom == "rom_proj_mfma":                               # (only if the FOM half is not the longer one)
                dom = "fom_chol_solve" if ms.get("fom_chol_solve", 0) > 0 else dom
        if dom == "rom_proj_mfma" and ms[dom] > 0:
            # what the projection kernel computes: psi rows from the sparse tables, the symmetric half of psi^T psi, psi^T F and --
            # for r <= 80, where the reduced system is factored and solved in the same kernel -- r^3/3 + 2 r^2 + the reduced QoI
            alg = S * (fl["syrk_sym"] + 2 * solver_r._rom.nterms * args.r + fl["rhs"]
                       + (fl["reduced_solve"] + 2 * pairs.n_obs * args.r if args.r <= 80 else 0))
            ach = alg / (ms[dom] * 1e-3) / 1e12
            roof = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                    "algorithmic_flops_per_launch": alg, "avg_launch_ms": ms[dom]}
        elif ms.get(dom, 0) > 0:
            # FOM interpreter: per sample one 8-B operand per multiply-add of the schedule
            # (the other operand sits in LDS), L / 1/L_ii / y / w each written once, x read once
            st = solver._engine("field" if args.params == "field" else args.params)._streams
            nload_f = int((st["fwd"][1] >= 0).sum())                 # ops that fetch a global operand
            nload_b = int((st["bwd"][0] == 1).sum() * 2 + (st["bwd"][0] > 1).sum())
            per_sample = 8 * (nload_f + nload_b + 2 * plan.nnzL + 4 * ops.n + pairs.xdim + pairs.n_obs)
            alg = S * per_sample
            ach = alg / (ms[dom] * 1e-3) / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": ach / PEAK_HBM_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms[dom]}

        cpu = None
        if world == 1 and args.cpu_samples > 0:
            cpu = cpu_baseline(args, phi, X[: args.cpu_samples].cpu().numpy(), res, pairs)

        out = {
            "metric": "FOM+ROM forward-solve sample pairs/sec (five-param fin)",
            "value": total_pairs / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.params}_param thermal fin, lattice m={args.m} (n={ops.n} DoF, nnz={ops.nnz}, "
                                   f"nnz(L)={plan.nnzL}), POD basis r={args.r}, {S} samples per GPU, "
                                   "FOM sparse Cholesky + LSPG ROM + QoIs + error per sample",
                       "samples_per_gpu": S, "n_dof": ops.n, "r": args.r, "params": args.params,
                       "projection": args.projection,
                       "flops_per_pair": int(sum(fl.values())), "failed_samples": n_bad},
            "roofline": roof,
            "cpu_baseline": cpu,
            "kernels_avg_ms": {k: round(v, 4) for k, v in ms.items() if v > 0},
            "kernels_serial_ms": serial_ms,
            "host_io_pairs_per_s": host_io,
            "cpu_baseline_all_cores": cpu_all,
            "other_projection": other,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def measured_traffic(kernel, workload_key):
    """HBM bytes per launch of `kernel` from the rocprofv3 --pmc passes of this same command
    (FETCH_SIZE and WRITE_SIZE collected in separate runs, profiles/r01_pmc_summary.json); None if the
    summary is absent or was taken on another workload."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("workload_key") != workload_key:
            return None
        return d["hbm_bytes_per_launch"].get(kernel)
    except Exception:
        return None


def cpu_baseline(args, phi, Xs, res, pairs):
    """The oracle's one-sample-at-a-time loop (what the reference does, minus FEniCS form
    assembly overhead) on ONE host core, on the first `cpu_samples` inputs of the GPU batch."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:
        limiter = None
    from oracle import fin_oracle as O
    prob = O.FinProblem(args.m)
    fo = O.FinOracle(prob)
    ro = O.AffineROMOracle(prob, phi)
    lift = {"five": fo.five_param_to_function, "nine": fo.nine_param_to_function, "field": lambda x: x}[args.params]
    budget_s = 25.0
    t0 = time.perf_counter()
    done = 0
    q = np.zeros((len(Xs), 9)); qr = np.zeros((len(Xs), 9))
    for i in range(len(Xs)):
        k = lift(Xs[i])
        w = fo.forward(k)
        w_r = ro.forward_reduced(k)
        q[i] = fo.qoi_operator(w); qr[i] = ro.qoi_reduced(w_r)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits() if hasattr(limiter, "restore_original_limits") else None
    gq = res["qoi"][:done].cpu().numpy(); gqr = res["qoi_r"][:done].cpu().numpy()
    dev = float(max(np.max(np.linalg.norm(gq - q[:done], axis=1) / np.linalg.norm(q[:done], axis=1)),
                    np.max(np.linalg.norm(gqr - qr[:done], axis=1) / np.linalg.norm(qr[:done], axis=1))))
    return {"value": done / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": f"first {done} samples of the GPU batch, oracle/fin_oracle.py loop (SciPy SuperLU FOM + NumPy LSPG ROM), "
                      f"1 thread; max rel QoI deviation GPU vs oracle on them = {dev:.2e}"}


if __name__ == "__main__":
    main()
