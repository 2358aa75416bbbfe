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
HMC_EPS = 7e-2                   # leapfrog step of the configs[4] rehearsal: acceptance 0.95 at 0.05, 0.65 at 0.07 (tools/hmc_eps_sweep.sh; erratic beyond: leapfrog resonances of the 1597-dimensional Gaussian prior)


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
    ap.add_argument("--stream", choices=("default", "own"), default="own",
                    help="pairs workload: the steps on a torch stream of their own (non-blocking; the ROM half of finrom_solve_pairs "
                         "then stays on it, no cross-stream wait per step) or on torch's default (null) stream")
    ap.add_argument("--no-other", action="store_true", help="skip the extra (untimed) pass with the other --projection")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-host-io", action="store_true", help="skip the extra (untimed) host-buffer pass")
    ap.add_argument("--workload", default="pairs", choices=["pairs", "hmc"],
                    help="pairs: the FOM+ROM dataset loop (BASELINE metric, configs[1..3]); hmc: BASELINE configs[4], chains of "
                         "sequential dependent one-sample ROM + learned-error value-and-gradient calls (steps = calls per chain)")
    ap.add_argument("--chains", type=int, default=4, help="hmc: number of independent chains (sharded over the GPUs)")
    ap.add_argument("--hmc-mode", default="device", choices=["device", "device-torch", "host"],
                    help="hmc: 'device' keeps the chains in HBM and runs a whole proposal as library launches (finrom_hmc_begin / "
                         "_leapfrog / _end: the position update in front of the contraction, the momentum update behind the "
                         "gradient), one captured HIP graph per proposal (hmc.run_chains_fused); 'device-torch' is round 3's form: "
                         "finrom_romml_grad between torch elementwise kernels in the same graph; 'host' is round 2's NumPy recursion "
                         "around one library call per evaluation (three copies and a synchronisation each)")
    ap.add_argument("--hmc-trace", action="store_true", help="hmc: keep and gather the per-proposal trace of every chain")
    ap.add_argument("--hmc-eps", type=float, default=None, help="hmc: leapfrog step size (default: tuned for 60-90 %% acceptance)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 run: nccl (= RCCL over xGMI, one rank per GPU) or gloo (host "
                         "gather; ranks may then share a GPU: rehearsal of the N > 1 path on a one-GPU box)")
    ap.add_argument("--sync-gather", action="store_true",
                    help="(RCCL) every step waits for its own all-gather before the next step's kernels start (default: the gather "
                         "of step i runs beside the kernels of step i + 1; all joined before the clock stops)")
    ap.add_argument("--force-pg", action="store_true",
                    help="create the torch.distributed process group and run the per-step gather even with ONE rank (the "
                         "collective then really executes: RCCL communicator init + all_gather_into_tensor on device tensors at "
                         "world = 1) -- rehearsal of the N > 1 code path on a one-GPU box")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / sharding rehearsal without a GPU: ranks start, build their shard of the globally keyed "
                         "inputs, gather a stand-in over gloo and rank 0 prints the JSON line with value = null")
    # rank processes get their arguments through the environment: torch.distributed.run's own parser chokes on script
    # options that are prefixes of its own (--m)
    return ap.parse_args(json.loads(os.environ["BENCH_ARGV"]) if "BENCH_ARGV" in os.environ else None)


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no rank environment: THIS process only launches.  It starts N fresh rank
    processes (torch.distributed.run, rendezvous on 127.0.0.1) BEFORE anything here has touched the GPU -- no torch import,
    no library load, no exec of an initialised process -- and relays rank 0's JSON line (the children inherit stdout)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BENCH_ARGV=json.dumps(sys.argv[1:]))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def global_uniform(seed, lo, hi, dim, a=0.1, b=10.0, block=4096):
    """Rows [lo, hi) of the ONE global input stream: block j of `block` rows comes from default_rng([seed, j]), so a rank's
    shard is the same numbers whatever the GPU count and nobody materialises the other ranks' rows."""
    out = np.empty((hi - lo, dim))
    for j in range(lo // block, (hi + block - 1) // block if hi > lo else lo // block):
        rows = np.random.default_rng([seed, j]).uniform(a, b, (block, dim))
        g0, g1 = max(lo, j * block), min(hi, (j + 1) * block)
        out[g0 - lo:g1 - lo] = rows[g0 - j * block:g1 - j * block]
    return out


def global_normal(seed, lo, hi, dim, block=1024):
    """As global_uniform, standard normals (the xi of the Gaussian-field sampler when drawn on the host)."""
    out = np.empty((hi - lo, dim))
    for j in range(lo // block, (hi + block - 1) // block if hi > lo else lo // block):
        rows = np.random.default_rng([seed, j]).standard_normal((block, dim))
        g0, g1 = max(lo, j * block), min(hi, (j + 1) * block)
        out[g0 - lo:g1 - lo] = rows[g0 - j * block:g1 - j * block]
    return out


def flops_per_pair(ops, plan, rom, n_obs, P, field=False):
    """SURVEY 8(d) formula, direct LSPG with the symmetric half of psi^T psi.  Assembly: 2 (P + 1) nnz for a parameter vector of P
    entries (affine sum of P + 1 value tables), 21 ncells for a nodal field (per cell: mean of three nodal values, nine scaled
    element entries added)."""
    n, r = ops.n, rom.n_r
    cc = np.diff(plan.col_ptr).astype(np.int64) + 1
    return {
        "assembly": 21 * len(ops.mesh.cells) if field else 2 * (P + 1) * ops.nnz,
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
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            launch_ranks(args)                             # never returns
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_run:
        return dry_run_hmc(args, rank, world) if args.workload == "hmc" else dry_run(args, rank, world)
    if args.workload == "hmc":
        return run_hmc(args, rank, local_rank, world)
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
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == "nccl":
        raise SystemExit(f"bench.py: rank {rank} has no GPU of its own ({ndev} visible); RCCL needs one GPU per rank "
                         "(--backend gloo lets ranks share a GPU for a rehearsal)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_pg = world > 1 or args.force_pg                   # process group + per-step gather (always for N > 1)
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:               # plain `python bench.py --force-pg`: a one-rank rendezvous of our own
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from bayesianinferencedl_amd import _ffi
    from bayesianinferencedl_amd.fom.thermal_fin import get_space
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    from bayesianinferencedl_amd.rom.basis import pod_basis
    from bayesianinferencedl_amd.pairs import FinPairSolver
    _ffi.check(_ffi.lib().finrom_set_device(dev_index))

    V = get_space(None, m=args.m)
    solver = Fin(V)
    bparams = "five" if args.params == "five" else "nine"
    phi = pod_basis(solver, args.r, n_snapshots=400, low=0.1, high=10.0, params=bparams, seed=1)
    solver_r = AffineROMFin(V, None, phi, projection=args.projection)
    pairs = FinPairSolver(V, phi, False, args.params, solver, solver_r)

    S = args.samples
    # inputs are keyed by the GLOBAL sample index, so rank g's shard is rows [g*S, (g+1)*S) of
    # the same stream whatever the GPU count (results independent of G)
    if args.params == "field":
        from bayesianinferencedl_amd.bayesian_inference.gaussian_field import make_cov_chol
        from bayesianinferencedl_amd.engine import FieldSampler
        sampler = FieldSampler(make_cov_chol(V, length=1.6))
        # xi on the device: Philox keyed by the GLOBAL sample index (seed 5), so rank g's shard is rows [g S, (g + 1) S) of one
        # stream whatever the GPU count -- no host draw, no upload (finrom_sampler_draw_seeded)
        X = sampler.draw(5, rank * S, S, like=torch.empty(0, device=dev, dtype=torch.float64))
    else:
        X = torch.from_numpy(global_uniform(3, rank * S, (rank + 1) * S, pairs.xdim)).to(dev)
    from bayesianinferencedl_amd.distributed import gather_rows

    if args.stream == "own":
        # the steps run on a stream of their own (a torch stream is a non-blocking HIP stream): beside the null stream the
        # library's CU-masked FOM stream -- a blocking stream, hipExtStreamCreateWithCUMask takes no flags -- cannot let the ROM
        # half stay on the caller's stream (finrom_solve_pairs)
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.Stream(dev))

    pending = []      # (RCCL) the steps' all-gathers in flight: each runs on the communicator's stream behind its own step's kernels
                      # and beside the NEXT step's -- the path has no dependence on the gathered rows; joined at the fence

    def step():
        res = pairs.solve_pairs(X)
        if use_pg:      # the one exchange step: QoI pairs of every shard (RCCL all_gather over xGMI)
            loc = torch.cat([res["qoi"], res["qoi_r"]], dim=1)
            if args.backend == "nccl" and not args.sync_gather:
                res["gathered"], work = gather_rows(loc, world, force=True, async_op=True)
                pending.append((work, loc, res["gathered"]))
            else:
                res["gathered"] = gather_rows(loc if args.backend == "nccl" else loc.cpu(), world, force=True)
        return res

    def fence():
        for work, _, _ in pending:                       # the current stream waits for every gather of the region
            if work is not None:
                work.wait()
        pending.clear()
        if use_pg:
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
    fence()                                               # barrier + device synchronisation on both sides of the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    dt = time.perf_counter() - t0
    L.finrom_profile_enable(0)
    prof = _ffi.profile_read()
    n_bad = int((res["info"] != 0).sum().item())

    cdev = dev if args.backend == "nccl" else torch.device("cpu")        # where the collectives' tensors live
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if use_pg:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # checksum of the gathered QoI pairs in GLOBAL sample order (not timed): equal for every GPU count on the same total
    import hashlib
    full = res["gathered"] if use_pg else torch.cat([res["qoi"], res["qoi_r"]], dim=1)
    gathered_sha = hashlib.sha256(full.cpu().numpy().tobytes()).hexdigest() if rank == 0 else None
    del full

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
        tmax = torch.tensor([dto], dtype=torch.float64, device=cdev)
        if use_pg:
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

    # context only (Gaussian-field inputs): what drawing a fresh shard costs on top of the pair step -- xi by Philox on the device
    # + k = exp(0.5 xi U) (finrom_sampler_draw_seeded); the timed region above works on a resident shard, as the contract asks
    sampler_ms = None
    if args.params == "field" and not args.no_profile:
        like = torch.empty(0, device=dev, dtype=torch.float64)
        sampler.draw(5, rank * S, S, like=like)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            sampler.draw(5, rank * S, S, like=like)
        e1.record()
        torch.cuda.synchronize()
        sampler_ms = e0.elapsed_time(e1) / 3

    if rank == 0:
        ops, plan = V.operators(), solver._plan
        fl = flops_per_pair(ops, plan, solver_r, pairs.n_obs, pairs.xdim, field=args.params == "field")
        total_pairs = world * S * args.steps
        ms = {k: (v[1] / v[0] if v[0] else 0.0) for k, v in prof.items()}       # avg ms per launch
        launches = {k: v[0] for k, v in prof.items()}
        # dominant kernel by measured time
        cand = {k: v[1] for k, v in prof.items()}
        dom = max(cand, key=cand.get) if any(cand.values()) else "rom_proj_mfma"
        if args.projection == "offline_online":
            npairs = solver_r._rom.gram_pairs
            fl["syrk_sym"] = npairs * args.r * (args.r + 1)          # multiply-adds of the block sum, symmetric half
            fl["psi"] = 0
            fl["rhs"] = 2 * 10 * args.r
        roof = roofline(dom, ms, launches, S, args, ops, plan, solver, solver_r, pairs, fl)

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
                       "projection": args.projection, "stream": args.stream,
                       "flops_per_pair": int(sum(fl.values())), "failed_samples": n_bad},
            "roofline": roof,
            "cpu_baseline": cpu,
            "kernels_avg_ms": {k: round(v, 4) for k, v in ms.items() if v > 0},
            "gathered_sha256": gathered_sha, "backend": args.backend if use_pg else None,
            "process_group": {"backend": dist.get_backend(), "world": dist.get_world_size(), "forced": bool(args.force_pg and world == 1)} if use_pg else None,
            "kernels_serial_ms": serial_ms,
            "host_io_pairs_per_s": host_io,
            "sampler_ms_per_step": sampler_ms,
            "value_incl_sampler": None if sampler_ms is None else world * S / (dt / args.steps + sampler_ms * 1e-3),
            "cpu_baseline_all_cores": cpu_all,
            "other_projection": other,
        }
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.destroy_process_group()


def run_hmc(args, rank, local_rank, world):
    """BASELINE configs[4]: `--chains` independent HMC chains (seed 6 + chain), each a sequence of `--steps` DEPENDENT
    one-sample evaluations of AffineROMFin.grad_romml (ROM adjoint + error-model value and input gradient; reference
    bayesian_inference/pymc_func_bayes_inverse.py:92-104,148-167, rom/averaged_affine_ROM.py:358-396) at m = 12, r = 81.
    Rank g owns chains g, g + N, ...; the chains of one rank advance in lockstep (one device call evaluates the current
    leapfrog point of each).  A step = one value-and-gradient evaluation of every chain; value = evaluations / s."""
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == "nccl":
        raise SystemExit(f"bench.py: rank {rank} has no GPU of its own ({ndev} visible)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))
    from bayesianinferencedl_amd import _ffi
    from bayesianinferencedl_amd.fom.thermal_fin import get_space
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
    from bayesianinferencedl_amd.rom.basis import pod_basis
    from bayesianinferencedl_amd.deep_learning.dl_model import ResBnFcModel
    from bayesianinferencedl_amd.bayesian_inference import hmc
    _ffi.check(_ffi.lib().finrom_set_device(dev_index))
    r = 81 if args.r == 80 else args.r                      # configs[4] is quoted at r = 81 (SURVEY 8(d) cfg 5)
    V = get_space(None, m=args.m)
    solver = Fin(V)
    phi = pod_basis(solver, r, n_snapshots=400, low=0.1, high=10.0, params="nine", seed=1)
    model = hmc_error_model(V.dim())
    solver_r = AffineROMFin(V, model, phi, projection=args.projection)
    k_true = np.exp(0.25 * global_normal(11, 0, 1, V.dim())[0])
    solver_r.set_data(solver.qoi_operator(solver.forward(k_true)[0]))
    mine = [c for c in range(args.chains) if c % world == rank]
    K0 = np.stack([np.exp(0.1 * np.random.default_rng(6 + c).standard_normal(V.dim())) for c in mine]) if mine else np.zeros((0, V.dim()))
    f = hmc.romml_value_and_grad(solver_r)
    L = 10
    steps = args.steps if args.steps >= L else 10000        # (a step = one evaluation here; fewer than one trajectory: configs[4]'s 10k)
    n_evals = 1 + steps // L * L                            # evaluation 0 (the start point) + whole trajectories
    eps = args.hmc_eps if args.hmc_eps is not None else HMC_EPS

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    seeds = [1000 + c for c in mine]                       # (K0 came from default_rng(6 + c): the chains' streams are others)

    def run(n, k0=None, sd=None):
        k0 = K0 if k0 is None else k0
        sd = seeds if sd is None else sd
        if not mine:
            return None
        if args.hmc_mode != "host":
            return hmc.run_chains_device(solver_r, k0, n, seeds=sd, n_leapfrog=L, eps=eps, fused=args.hmc_mode == "device",
                                         keep_trace=args.hmc_trace)
        return hmc.run_chains(f, k0, n, seeds=sd, n_leapfrog=L, eps=eps, keep_trace=args.hmc_trace)
    run(1 + max(L, args.warmup // L * L))
    fence()
    Lb = _ffi.lib()
    Lb.finrom_profile_reset(); Lb.finrom_profile_enable(0 if args.no_profile else 1)
    t0 = time.perf_counter()
    res = run(n_evals)
    fence()
    dt = time.perf_counter() - t0
    Lb.finrom_profile_enable(0)
    if args.hmc_mode != "host" and not args.no_profile and mine:
        # kernels replayed from a HIP graph cannot be bracketed with events: per-kernel times come from a short pass of the same
        # chains with the launches in stream order (not timed)
        Lb.finrom_profile_reset(); Lb.finrom_profile_enable(1)
        hmc.run_chains_device(solver_r, K0, 1 + 20 * L, seeds=seeds, n_leapfrog=L, eps=eps, graph=False, fused=args.hmc_mode == "device")
        torch.cuda.synchronize()
        Lb.finrom_profile_enable(0)
    prof = _ffi.profile_read()
    cdev = dev if args.backend == "nccl" else torch.device("cpu")
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # the gather at the end ("cfg5: one chain per GPU, no communication until the final trace gather";
    # bayesian_inference/inference.py:165-169 keeps the trace): end states, accept counts and -- with --hmc-trace -- the
    # per-proposal traces of every rank's chains, assembled on rank 0 in chain order
    chains = gather_chains(res, mine, world, dist if world > 1 else None)
    # one chain alone, one sample per call: the per-call latency a single PyMC chain would see (not timed above)
    lat = None
    if rank == 0 and mine:
        n1 = 1 + min(1000, steps) // L * L
        run(1 + L, K0[:1], seeds[:1])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(n1, K0[:1], seeds[:1])
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t1) / n1
    if rank == 0:
        total = args.chains * n_evals
        ms = {k: (v[1] / v[0] if v[0] else 0.0) for k, v in prof.items()}
        dom = max(prof, key=lambda k: prof[k][1]) if any(v[1] for v in prof.values()) else None
        roof = None                                       # (no throughput roofline for a latency chain: see critical_path)
        # a chain of DEPENDENT one-sample calls has no throughput roofline: what bounds it is the critical path of a leapfrog step --
        # the step's launches in order, each waiting for the previous one (durations from the stream-order pass above)
        lib_order = ["rom_proj_mfma", "rom_reduced_solve", "misc"]
        crit = {k: round(ms[k] * 1e3 * (prof[k][0] / max(prof["rom_proj_mfma"][0], 1)), 2) for k in lib_order if ms.get(k, 0) > 0} if prof.get("rom_proj_mfma", (0, 0))[0] else None
        critical_path = None if not crit else {
            "bound": "latency", "per_step_us": crit, "sum_us": round(sum(crit.values()), 2),
            "measured_step_us": round(1e6 * dt / n_evals, 2),
            "note": "one leapfrog step = dependent launches: contraction + theta + error-model forward (rom_proj_mfma), factor + "
                    "forward / adjoint solves and the gradient contraction (rom_reduced_solve: two launches), error-model backward + "
                    "momentum update (misc: also the proposal's begin / end kernels); kernel durations from a stream-order pass, "
                    "the measured step adds the graph's per-node launch gaps"}
        cpu = None
        if world == 1 and args.cpu_samples > 0:
            cpu = hmc_cpu_baseline(args, phi, model, solver_r.data, K0, res)
        print(json.dumps({
            "metric": "ROM+DL value-and-gradient evaluations/sec (HMC chains, BASELINE configs[4])",
            "value": total / dt, "unit": "evals/s", "n_gpus": world, "steps": n_evals - 1, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / n_evals, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64 (ROM) + f32 (error model)", "data": "synthetic",
            "config": {"workload": f"HMC: {args.chains} chains x {n_evals} dependent one-sample grad_romml calls, lattice m={args.m} "
                                   f"(n={V.dim()}), r={r}, res_bn_fc error model 5 x 50, {L} leapfrog steps per proposal, "
                                   f"{len(mine)} chains per call on rank 0", "chains": args.chains, "evals_per_chain": n_evals,
                       "r": r, "projection": args.projection, "accepted": res["accept"].tolist() if res else None,
                       "accepted_all_chains": chains["accept"],
                       "proposals": res["proposals"] if res else None, "eps": eps, "mode": args.hmc_mode,
                       "fused_leapfrog": bool(res.get("fused")) if res else None,
                       "hip_graph": bool(res.get("graph")) if res else None,
                       "acceptance": float(np.mean(res["accept"]) / max(res["proposals"], 1)) if res else None},
            "roofline": roof, "critical_path": critical_path, "cpu_baseline": cpu,
            "chains_sha256": chains["sha256"], "chains_gathered": chains["n"], "trace_sha256": chains["trace_sha256"],
            "single_chain_latency_ms_per_call": None if lat is None else 1e3 * lat,
            "kernels_avg_ms": {k: round(v, 4) for k, v in ms.items() if v > 0}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def gather_chains(res, mine, world, dist):
    """End states [chains x n], accept counts and optional traces of every rank's chains on rank 0, in chain order, + checksums.
    The payload is small (a few fields per chain), so it travels as Python objects (gather_object: gloo or RCCL alike)."""
    import hashlib
    part = {c: (res["K"][i], int(res["accept"][i]), None if res.get("trace") is None else res["trace"][:, i]) for i, c in enumerate(mine)} if res else {}
    parts = [part]
    if dist is not None:
        parts = [None] * world
        dist.all_gather_object(parts, part)
    allc = {}
    for p_ in parts:
        allc.update(p_)
    order = sorted(allc)
    h = hashlib.sha256()
    for c in order:
        h.update(np.ascontiguousarray(allc[c][0]).tobytes()); h.update(np.int64(allc[c][1]).tobytes())
    ht = None
    if order and allc[order[0]][2] is not None:
        ht = hashlib.sha256()
        for c in order:
            ht.update(np.ascontiguousarray(allc[c][2]).tobytes())
        ht = ht.hexdigest()
    return {"n": len(order), "sha256": h.hexdigest(), "accept": [allc[c][1] for c in order], "trace_sha256": ht}


def dry_run_hmc(args, rank, world):
    """No GPU: the chain deal (rank g owns chains g, g + N, ...), a stand-in for the chains (a deterministic walk seeded per chain)
    and the gather of end states / accept counts / traces over gloo; rank 0 checks the assembled result against the one-process
    one and prints the JSON line with value = null."""
    import torch.distributed as dist
    n, L = 64, 10
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")

    def walk(chain_ids):
        K = np.stack([np.cumsum(np.random.default_rng(1000 + c).standard_normal((args.steps // L + 1, n)), axis=0) for c in chain_ids], axis=1) \
            if chain_ids else np.zeros((args.steps // L + 1, 0, n))
        return {"K": K[-1], "accept": np.array([int(abs(K[-1, i]).sum()) % 7 for i in range(len(chain_ids))]), "trace": K}
    mine = [c for c in range(args.chains) if c % world == rank]
    t0 = time.perf_counter()
    got = gather_chains(walk(mine), mine, world, dist if world > 1 else None)
    dt = time.perf_counter() - t0
    if rank == 0:
        every = list(range(args.chains))
        ref = gather_chains(walk(every), every, 1, None)
        assert got == ref, "gathered chains differ from the one-process run"
        print(json.dumps({"metric": "ROM+DL value-and-gradient evaluations/sec (HMC chains, BASELINE configs[4])", "value": None,
                          "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "config": {"workload": "dry run: chain deal + gloo gather of end states / accept counts / traces only",
                                     "chains": args.chains}, "dry_run": True, "gather_s": dt, "chains_gathered": got["n"],
                          "chains_sha256": got["sha256"], "trace_sha256": got["trace_sha256"]}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def hmc_error_model(n):
    """res_bn_fc error model of the reference's load_bn_model(randobs=False) shape (5 units x 50, deep_learning/dl_model.py:
    225), random weights and non-trivial batch-norm statistics (no checkpoint can be read here), small output scale."""
    from bayesianinferencedl_amd.deep_learning.dl_model import ResBnFcModel
    rng = np.random.default_rng(7)
    model = ResBnFcModel(n, 9, 5, 50, seed=7)
    for u in model.units + [model.head]:
        u["gamma"] = rng.uniform(0.5, 1.5, u["gamma"].shape).astype(np.float32)
        u["beta"] = rng.normal(0, 0.2, u["beta"].shape).astype(np.float32)
        u["mean"] = rng.normal(0, 0.2, u["mean"].shape).astype(np.float32)
        u["var"] = rng.uniform(0.5, 2.0, u["var"].shape).astype(np.float32)
    model.head["W"] *= np.float32(0.02)
    return model


def hmc_cpu_baseline(args, phi, model, data, K0, res):
    """oracle.grad_romml_oracle (dense NumPy restatement of rom/averaged_affine_ROM.py:358-396) on one core, a bounded
    number of evaluations at the chains' start points."""
    from oracle import fin_oracle as O
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(limits=1)
    except Exception:
        lim = None
    ro = O.AffineROMOracle(O.FinProblem(args.m), phi)
    ro.set_data(data)
    t0 = time.perf_counter(); done = 0
    while done < args.cpu_samples and time.perf_counter() - t0 < 20.0:
        O.grad_romml_oracle(ro, model, K0[done % len(K0)])
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": f"{done} evaluations of oracle.grad_romml_oracle at the chains' start points, 1 thread"}


def roofline(dom, ms, launches, S, args, ops, plan, solver, solver_r, pairs, fl):
    """Roofline object of the dominant kernel `dom` (a profile slot of the library), chosen BY KERNEL: what it computes or
    must move per launch over its measured average launch time.  A slot that is launched several times per step
    (workspace pieces) processes S / pieces samples per launch."""
    if not ms.get(dom, 0) > 0:
        return None
    r, n, n_obs = args.r, ops.n, pairs.n_obs
    rp = (r + 15) // 16 * 16
    per_launch = S * args.steps / max(launches.get(dom, args.steps), 1)          # samples one launch processes
    t = ms[dom] * 1e-3
    key = f"{args.params}/m{args.m}/r{args.r}/S{S}" + ("" if args.projection == "direct" else "/" + args.projection)

    def hbm(bytes_per_sample, model):
        alg = per_launch * bytes_per_sample
        traffic, src = measured_traffic(dom, key)
        ach = alg / t / 1e9
        return {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms[dom], "model": model}

    def mfma(flops_per_sample, model):
        alg = per_launch * flops_per_sample
        ach = alg / t / 1e12
        traffic, src = measured_traffic(dom, key)
        return {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_source": src, "algorithmic_flops_per_launch": alg,
                "avg_launch_ms": ms[dom], "model": model}

    if dom == "rom_proj_mfma" and args.projection == "direct":
        # psi rows from the sparse tables, the symmetric half of psi^T psi, psi^T F on the ROOT rows only (F = 0 elsewhere) and --
        # where the reduced system is factored and solved in the same kernel (r <= 80, r > 96) -- r^3/3 + 2 r^2 + the reduced QoI
        nroot = int(np.count_nonzero(ops.F))
        f = fl["syrk_sym"] + 2 * solver_r._rom.nterms * r + 2 * nroot * r
        if r <= 80 or r > 96:       # (96 < r: fused across the sample's waves when w_r is not asked for -- it is not, here)
            f += fl["reduced_solve"] + 2 * n_obs * r
        return mfma(f, "n r (r+1) + 2 nterms r + 2 nroot r (+ r^3/3 + 2 r^2 + 2 n_obs r for r <= 80 and r > 96)")
    if dom == "rom_proj_mfma":
        # offline/online form: the block sum streams tile images out of L2; what HAS to cross HBM is theta in, w_r + qoi_r out
        return hbm(8 * (9 + r + n_obs), "compulsory bytes only (theta in; w_r, qoi_r out): the block images are L2 traffic")
    if dom == "fom_chol_solve":
        eng = solver._engine("field" if args.params == "field" else args.params)
        if eng.band is not None:
            bp = eng.band
            # frontal band sweep: value slots read once, every column of L (+ extras) written once and read once, y written and
            # read, w written, QoI out -- nothing else leaves the registers
            # (value slots: the physical ones -- slots with the same affine record are shared and mostly served by L2)
            if getattr(eng, "band_qoi_only", False):
                # QoI-only form (the pair path asks for no w): only the POST's columns, y and w travel; a fin leaves nif doubles
                nLp = bp.npost * bp.NSP
                return hbm(8 * (eng.band_slots + 2 * nLp + 2 * bp.nLx + 3 * bp.npost + 2 * bp.nfins * (bp.q + 1) + n_obs),
                           "QoI-only form: value slots + 2 (L_post + Lx) + 3 n_post + 2 nfins nif + n_obs doubles per sample")
            return hbm(8 * (eng.band_slots + 2 * bp.nL + 2 * bp.nLx + 3 * n + n_obs), "value slots + 2 (L + Lx) + 3 n + n_obs doubles per sample")
        # interpreter: lower bound -- L written once and read once, y / w, parameters; its operand re-fetches come on top
        return hbm(8 * (2 * plan.nnzL + 4 * n + pairs.xdim + n_obs), "lower bound: 2 nnz(L) + 4 n + xdim + n_obs doubles per sample")
    if dom == "rom_reduced_solve":
        # wide bases: the packed A_r is read and its factor written by the blocked Cholesky, the factor is read again by the
        # substitutions (at least once), B_r in, w_r and qoi_r out
        npk = rp * (rp + 1) // 2
        return hbm(8 * (3 * npk + rp + r + n_obs), "lower bound: 3 packed triangles + B_r + w_r + qoi_r per sample")
    if dom == "fom_assemble":
        eng = solver._engine("field" if args.params == "field" else args.params)
        nval = eng.band_slots if eng.band is not None else len(eng._streams["a_list"])
        return hbm(8 * (nval + pairs.xdim), "value slots written + parameters read")
    if dom == "sampler_gemm_exp":
        return mfma(n * n, "n^2 (triangular half of the dense GEMM)")
    return hbm(8 * (pairs.xdim + 2 * n_obs), "compulsory bytes")


def dry_run(args, rank, world):
    """No GPU: the launcher, the rank environment, the global keying of the inputs and the gather, over gloo.  Each rank
    'solves' its shard with a stand-in (row sums); rank 0 checks the gathered array against the single-process one and
    prints the JSON line of the contract with value = null."""
    import hashlib
    import torch
    import torch.distributed as dist
    from bayesianinferencedl_amd.distributed import gather_rows
    S, dim = args.samples, {"five": 5, "nine": 9, "field": 16}[args.params]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    gen = global_normal if args.params == "field" else global_uniform
    X = gen(3, rank * S, (rank + 1) * S, dim)
    local = torch.from_numpy(np.stack([X.sum(1), (X * X).sum(1)], 1))
    t0 = time.perf_counter()
    full = gather_rows(local, world) if world > 1 else local
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        Xall = gen(3, 0, world * S, dim)
        ref = np.stack([Xall.sum(1), (Xall * Xall).sum(1)], 1)
        assert np.array_equal(full.numpy(), ref), "gathered shards differ from the single-process stream"
        print(json.dumps({"metric": "FOM+ROM forward-solve sample pairs/sec (five-param fin)", "value": None, "unit": "pairs/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "config": {"workload": "dry run: launcher + sharding + gloo gather only", "samples_per_gpu": S},
                          "dry_run": True, "gather_s": dt,
                          "gathered_sha256": hashlib.sha256(full.numpy().tobytes()).hexdigest()}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def measured_traffic(kernel, workload_key):
    """HBM bytes per launch of `kernel` from the rocprofv3 --pmc passes of this same command on this workload (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, tools/collect_profiles.sh -> profiles/rNN_pmc_summary_<workload>.json; the newest
    round's summary for the workload wins) -> (bytes or None, source or None).  The counters cannot be collected inside a timed
    run (rocprofv3 serialises the kernels), so the figure is a RECORD of the same command on the builder's box, not a measurement
    of this run: `source` names the file and the commit it was taken at, and any other workload (sample count, sizes) gets None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary_*.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get("workload_key") == workload_key and d["hbm_bytes_per_launch"].get(kernel) is not None:
                return d["hbm_bytes_per_launch"][kernel], {"file": os.path.relpath(path, ROOT), "commit": d.get("commit"),
                                                           "note": "rocprofv3 --pmc record of the same command (2 FETCH + WRITE, per launch); not collected in this run"}
        except Exception:
            continue
    return None, None


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
