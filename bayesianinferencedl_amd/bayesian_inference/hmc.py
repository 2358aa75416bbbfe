"""Minimal Hamiltonian Monte Carlo driver for BASELINE configs[4]: chains of sequential, DEPENDENT one-sample
value-and-gradient calls of the ROM + learned-error misfit (`AffineROMFin.grad_romml`), the call pattern PyMC3's NUTS
drives through the reference's Theano op (bayesian_inference/pymc_func_bayes_inverse.py:92-104 `err_grad_ROMML`,
:148-167 `SqErrorOpROMML.perform`, model :186-203: potential = misfit / sigma^2 on a latent Gaussian field).

PyMC3 / Theano are control plane and out of scope; what the hot path needs from them is the leapfrog recursion
that makes every evaluation's input depend on the previous evaluation's gradient.  This module is that recursion and
nothing more: fixed-length leapfrog trajectories with a Metropolis test, an i.i.d. Gaussian prior around a mean field (the
reference's Matern-5/2 latent GP would add a dense n x n triangular solve per step on the HOST, which is not the path
measured here), independent chains seeded `seed + chain`.

Chains are independent, so C chains may advance in LOCKSTEP: one device call evaluates the current leapfrog point of every
chain (a batch of C samples), each chain keeping its own momentum, random stream and accept/reject decision.  On N GPUs a
rank owns chains `rank, rank + N, ...` (one chain per GPU at N = C) and there is no communication until the traces are
gathered."""
from __future__ import annotations

import numpy as np


class HmcResult(dict):
    __getattr__ = dict.__getitem__


def potential(loss, grad, K, mean, sigma, tau):
    """U(k) = loss(k) / sigma^2 + |k - mean|^2 / (2 tau^2) and its gradient, row-wise for a batch K [C, n]."""
    d = K - mean
    U = np.asarray(loss, dtype=np.float64) / sigma ** 2 + 0.5 * np.einsum("cn,cn->c", d, d) / tau ** 2
    return U, np.asarray(grad, dtype=np.float64) / sigma ** 2 + d / tau ** 2


def run_chains(value_and_grad, K0, n_evals, *, seeds, eps=2e-3, n_leapfrog=10, sigma=0.05, tau=0.5, mean=None, record=None,
               keep_trace=False):
    """Advance C = len(K0) chains in lockstep for `n_evals` value-and-gradient evaluations per chain.

    value_and_grad(K [C, n]) -> (loss [C], grad [C, n], bad [C] bool): ONE device call per leapfrog point; `bad` marks
    samples whose reduced operator was not positive definite (treated as infinite potential: the proposal is rejected).
    record: optional set of evaluation indices whose (input, loss, gradient) are kept for parity checks.
    Returns HmcResult(K [C, n] final states, accept [C] accepted proposals, proposals, n_evals (per chain),
    trace [proposals + 1, C, n] if keep_trace, recorded = list of (eval index, K copy, loss, grad) for parity checks)."""
    K = np.array(K0, dtype=np.float64, copy=True)
    C, n = K.shape
    mean = K.copy() if mean is None else np.broadcast_to(np.asarray(mean, dtype=np.float64), K.shape)
    rngs = [np.random.default_rng(s) for s in seeds]
    assert len(rngs) == C
    recorded = []
    evals = 0

    def evaluate(Kq):
        nonlocal evals
        loss, grad, bad = value_and_grad(Kq)
        if record is not None and evals in record:
            recorded.append((evals, Kq.copy(), np.array(loss, copy=True), np.array(grad, copy=True)))
        evals += 1
        U, dU = potential(loss, grad, Kq, mean, sigma, tau)
        bad = np.asarray(bad, dtype=bool) | ~np.isfinite(U)
        U = np.where(bad, np.inf, U)
        dU = np.where(bad[:, None], 0.0, dU)
        return U, dU

    U, dU = evaluate(K)                                              # evaluation 0: the starting point
    if not np.all(np.isfinite(U)):
        raise ValueError("HMC start point has an indefinite reduced operator")
    trace = [K.copy()] if keep_trace else None
    accept = np.zeros(C, np.int64)
    proposals = 0
    while evals + n_leapfrog <= n_evals:
        P = np.stack([r.standard_normal(n) for r in rngs])
        H0 = U + 0.5 * np.einsum("cn,cn->c", P, P)
        Kq, Pq, Uq, dUq = K.copy(), P.copy(), U, dU
        for _ in range(n_leapfrog):                                  # each step's input depends on the previous gradient
            Pq = Pq - 0.5 * eps * dUq
            Kq = Kq + eps * Pq
            Uq, dUq = evaluate(Kq)
            Pq = Pq - 0.5 * eps * dUq
        H1 = Uq + 0.5 * np.einsum("cn,cn->c", Pq, Pq)
        u = np.array([r.uniform() for r in rngs])
        with np.errstate(over="ignore", invalid="ignore"):
            ok = np.isfinite(H1) & (np.log(u) < H0 - H1)
        K = np.where(ok[:, None], Kq, K); U = np.where(ok, Uq, U); dU = np.where(ok[:, None], dUq, dU)
        accept += ok
        proposals += 1
        if keep_trace:
            trace.append(K.copy())
    return HmcResult(K=K, accept=accept, proposals=proposals, n_evals=evals, recorded=recorded,
                     trace=np.stack(trace) if keep_trace else None)


def romml_value_and_grad(solver_r):
    """The evaluation the reference's SqErrorOpROMML performs, batched over chains: AffineROMFin.grad_romml_batch."""
    def f(K):
        res = solver_r.grad_romml_batch(K)
        return np.asarray(res["loss"]), np.asarray(res["grad"]), np.asarray(res["info"]) != 0
    return f


def run_chains_fused(solver_r, K0, n_evals, *, seeds, eps=2e-3, n_leapfrog=10, sigma=0.05, tau=0.5, mean=None, record=None,
                     keep_trace=False, graph=True, data=None, block=32):
    """`run_chains_device` with the trajectory's arithmetic INSIDE the library (round 4: finrom_hmc_begin / _leapfrog / _end,
    include/finrom.h): a leapfrog step is the four launches of finrom_romml_grad and nothing else -- the position update rides in
    front of the contraction and the error model's forward pass, the momentum update behind the gradient -- and a proposal is
    1 + 4 n_leapfrog + 2 launches, captured once and replayed.  Same random numbers, same order of evaluations and the same chains
    as `run_chains` (the host recursion) up to the rounding of fused multiply-adds; same return value as `run_chains_device`.
    Raises _ffi.FinromError (FINROM_ERR_UNSUPPORTED) where the library has no one-sample form for the model -- callers fall back to
    `run_chains_device(..., fused=False)`."""
    import ctypes as C
    import torch
    from .. import _ffi
    L = _ffi.lib()
    rom, mlp, Sop = solver_r._rom, solver_r._dev_model, solver_r._avg._S
    if mlp is None:
        raise _ffi.FinromError("run_chains_fused needs the error model on the device (a ResBnFcModel)")
    solver_r._ensure_gradient()
    dev = torch.device("cuda", torch.cuda.current_device())
    f64 = dict(dtype=torch.float64, device=dev)
    i64 = dict(dtype=torch.int64, device=dev)
    K = torch.as_tensor(np.ascontiguousarray(K0, dtype=np.float64), **f64).clone()
    Cn, n = K.shape
    mean_t = K.clone() if mean is None else torch.as_tensor(np.broadcast_to(np.asarray(mean, dtype=np.float64), (Cn, n)).copy(), **f64)
    data_np = np.ascontiguousarray(solver_r.data if data is None else data, dtype=np.float64)
    data_t = torch.as_tensor(data_np, **f64)
    per_sample = 1 if data_np.ndim == 2 else 0
    rngs = [np.random.default_rng(s) for s in seeds]
    assert len(rngs) == Cn
    c_lik, c_pri = 1.0 / sigma ** 2, 1.0 / tau ** 2
    n_prop = max(0, (n_evals - 1) // n_leapfrog)
    B = max(1, min(block, n_prop))
    Kq = [torch.empty_like(K), torch.empty_like(K)]
    P, dUq, dU = torch.zeros_like(K), torch.zeros_like(K), torch.zeros_like(K)
    U, H0, loss = torch.zeros(Cn, **f64), torch.zeros(Cn, **f64), torch.zeros(Cn, **f64)
    info = torch.zeros(Cn, dtype=torch.int32, device=dev)
    P_dev, lu_dev = torch.zeros(B, Cn, n, **f64), torch.zeros(B, Cn, **f64)
    jt, pt, acc = torch.zeros(1, **i64), torch.zeros(1, **i64), torch.zeros(Cn, **i64)
    trace = torch.zeros(n_prop + 1, Cn, n, **f64) if keep_trace else None
    grad_rec = torch.zeros_like(K)                                   # raw misfit gradient, written only for recorded evaluations
    st = _ffi.HmcState(C=Cn, n=n, eps=eps, c_lik=c_lik, c_pri=c_pri, mean=mean_t.data_ptr(), K=K.data_ptr(), U=U.data_ptr(),
                       dU=dU.data_ptr(), Kq=(C.c_void_p * 2)(Kq[0].data_ptr(), Kq[1].data_ptr()), P=P.data_ptr(), dUq=dUq.data_ptr(),
                       H0=H0.data_ptr(), P_block=P_dev.data_ptr(), lu_block=lu_dev.data_ptr(), jt=jt.data_ptr(), pt=pt.data_ptr(),
                       accept=acc.data_ptr(), trace=trace.data_ptr() if trace is not None else None, loss=loss.data_ptr(),
                       info=info.data_ptr())
    st0 = _ffi.HmcState.from_buffer_copy(st)                         # evaluation 0: a "step" of length zero from K itself
    st0.eps = 0.0

    def stream():
        return torch.cuda.current_stream().cuda_stream

    def leap(state, step, want_grad=False):
        _ffi.check(L.finrom_hmc_leapfrog(rom._h, mlp._h, Sop.ptr, C.byref(state), step, data_t.data_ptr(), per_sample,
                                         grad_rec.data_ptr() if want_grad else None, None, None, stream()), "finrom_hmc_leapfrog")
        Sop.used_on(stream())

    recorded, evals = [], 0

    def note(step):
        nonlocal evals
        if record is not None and evals in record:
            recorded.append((evals, Kq[(step + 1) & 1].cpu().numpy().copy(), loss.cpu().numpy().copy(), grad_rec.cpu().numpy().copy()))
        evals += 1

    def proposal(rec=False):
        _ffi.check(L.finrom_hmc_begin(C.byref(st), stream()), "finrom_hmc_begin")
        for i in range(n_leapfrog):                                  # each step's input depends on the previous gradient
            leap(st, i, want_grad=rec)
            if rec:
                note(i)
        _ffi.check(L.finrom_hmc_end(C.byref(st), n_leapfrog, stream()), "finrom_hmc_end")

    # evaluation 0: the starting point (also warms the library up: workspaces, function attributes)
    Kq[0].copy_(K)
    leap(st0, 0, want_grad=True)                                     # Kq[1] = K + 0 * P, dUq = grad U / c_pri at K
    note(0)
    D = K - mean_t
    Uv = torch.linalg.vecdot(D, D).mul_(0.5 * c_pri).add_(loss, alpha=c_lik)
    U.copy_(torch.nan_to_num_(Uv, nan=float("inf"), posinf=float("inf"), neginf=float("inf")).masked_fill_(info.ne(0), float("inf")))
    if not bool(torch.isfinite(U).all()):
        raise ValueError("HMC start point has an indefinite reduced operator")
    dU.copy_(dUq)
    if trace is not None:
        trace[0].copy_(K)
    g = None
    if graph and n_prop > 0:
        state = [t.clone() for t in (K, U, dU, acc, jt, pt)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                               # warm-up on a side stream, as torch's graph recipe asks
            proposal()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            proposal()
        for t, t0 in zip((K, U, dU, acc, jt, pt), state):           # (the warm-up moved the state; the capture does not run)
            t.copy_(t0)
        if trace is not None:
            trace[1].zero_()
    done = 0
    while done < n_prop:
        nb = min(B, n_prop - done)
        P_host, lu_host = np.zeros((B, Cn, n)), np.zeros((B, Cn))
        for j in range(nb):                                         # the host chain's draws, in its order
            for c_, r in enumerate(rngs):
                P_host[j, c_] = r.standard_normal(n)
            with np.errstate(divide="ignore"):
                lu_host[j] = np.log(np.array([r.uniform() for r in rngs]))
        P_dev.copy_(torch.from_numpy(P_host)); lu_dev.copy_(torch.from_numpy(lu_host))
        jt.zero_()
        for j in range(nb):
            first = 1 + (done + j) * n_leapfrog
            if record is not None and any(first <= e < first + n_leapfrog for e in record):
                assert evals == first
                proposal(rec=True)
            else:
                g.replay() if g is not None else proposal()
                evals += n_leapfrog
        done += nb
    return HmcResult(K=K.cpu().numpy(), accept=acc.cpu().numpy(), proposals=n_prop, n_evals=evals, recorded=recorded,
                     trace=trace.cpu().numpy() if trace is not None else None, graph=g is not None, fused=True)


def run_chains_device(solver_r, K0, n_evals, *, seeds, eps=2e-3, n_leapfrog=10, sigma=0.05, tau=0.5, mean=None, record=None,
                      keep_trace=False, graph=True, data=None, block=32, fused=None):
    """`run_chains` with the chains RESIDENT ON THE DEVICE (torch tensors on the current CUDA device): positions, momenta,
    potentials, the Metropolis test and the accept counters never visit the host.  A whole PROPOSAL -- momentum in, n_leapfrog
    steps of (a few elementwise kernels around ONE library call, finrom_romml_grad on the tensors in place), Hamiltonians,
    accept / reject, state update -- is captured once in a HIP graph (torch.cuda.CUDAGraph: the library launches on torch's capture
    stream) and replayed: one graph launch per proposal, no synchronisation until the chain ends.  The random numbers are the
    host chain's: every chain's NumPy generator is drawn in the same order (n normals for the momentum, then one uniform), `block`
    proposals ahead -- the draws do not depend on the state -- while the device works on the previous block, and uploaded
    block by block; the graph picks its proposal's slice by a device-side counter.  (First version: one graph per leapfrog step
    and one synchronisation per proposal; the ~25 small launches and two copies around each trajectory were a quarter of
    the time, tools/graph_call_cost.py.)

    Proposals that contain an evaluation index listed in `record` run the same operations in stream order, step by step, so that
    the evaluation's input, loss and gradient can be copied out.  graph=False: everything in stream order.

    Same chains as run_chains(romml_value_and_grad(solver_r), ...) up to the rounding of the elementwise updates.
    Returns HmcResult(K [C, n] (NumPy), accept, proposals, n_evals, recorded, trace, graph: whether a graph was replayed)."""
    import torch
    if fused is None or fused:
        # the trajectory's arithmetic inside the library (round 4); fused=None: fall back to the torch-op form below where the
        # library has no one-sample form for this model (FINROM_ERR_UNSUPPORTED)
        from .. import _ffi
        try:
            return run_chains_fused(solver_r, K0, n_evals, seeds=seeds, eps=eps, n_leapfrog=n_leapfrog, sigma=sigma, tau=tau,
                                    mean=mean, record=record, keep_trace=keep_trace, graph=graph, data=data, block=block)
        except _ffi.FinromError:
            if fused:
                raise
    dev = torch.device("cuda", torch.cuda.current_device())
    f64 = dict(dtype=torch.float64, device=dev)
    K = torch.as_tensor(np.ascontiguousarray(K0, dtype=np.float64), **f64).clone()
    C, n = K.shape
    mean_t = K.clone() if mean is None else torch.as_tensor(np.broadcast_to(np.asarray(mean, dtype=np.float64), (C, n)).copy(), **f64)
    data_t = torch.as_tensor(np.ascontiguousarray(solver_r.data if data is None else data, dtype=np.float64), **f64)
    rngs = [np.random.default_rng(s) for s in seeds]
    assert len(rngs) == C
    c_lik, c_pri = 1.0 / sigma ** 2, 1.0 / tau ** 2
    n_prop = max(0, (n_evals - 1) // n_leapfrog)
    B = max(1, min(block, n_prop))
    # static tensors (the graph's operands): state K, U, dU | work Kq, Pq, D, dUq | inputs P_dev, lu_dev | counters
    Kq, Pq, D, dUq = (torch.empty_like(K) for _ in range(4))
    U, dU = torch.zeros(C, **f64), torch.zeros_like(K)
    P_dev, lu_dev = torch.zeros(B, C, n, **f64), torch.zeros(B, C, **f64)
    P0, lu = torch.zeros(1, C, n, **f64), torch.zeros(1, C, **f64)
    jt = torch.zeros(1, dtype=torch.int64, device=dev)                # proposal inside the uploaded block
    pt = torch.zeros(1, dtype=torch.int64, device=dev)                # proposal of the chain (trace row)
    acc = torch.zeros(C, dtype=torch.int64, device=dev)
    trace = torch.zeros(n_prop + 1, C, n, **f64) if keep_trace else None
    out = {}

    def evaluate():
        """dUq (= grad U / c_pri), out <- value and gradient at Kq (all static tensors: the same buffers at every call once captured)."""
        res = solver_r.grad_romml_batch(Kq, data=data_t)
        torch.sub(Kq, mean_t, out=D)
        torch.add(D, res["grad"], alpha=c_lik / c_pri, out=dUq)     # dU / c_pri (one kernel; c_pri rides in the momentum updates' alpha)
        dUq.masked_fill_(res["info"].ne(0)[:, None], 0.0)           # an indefinite reduced operator: no force, rejected below
        out["loss"], out["grad"], out["info"] = res["loss"], res["grad"], res["info"]

    def step():
        Kq.add_(Pq, alpha=eps)
        evaluate()
        Pq.add_(dUq, alpha=-eps * c_pri)                           # two half steps; the ends of a trajectory correct by +- eps/2

    inf = float("inf")

    def potential_now():
        # (as few kernels as possible: each is a node of the proposal's graph, ~1.6 us)
        Uv = torch.add(torch.linalg.vecdot(D, D).mul_(0.5 * c_pri), out["loss"], alpha=c_lik)
        return torch.nan_to_num_(Uv, nan=inf, posinf=inf, neginf=inf).masked_fill_(out["info"].ne(0), inf)

    recorded, evals = [], 0

    def note():
        nonlocal evals
        if record is not None and evals in record:
            recorded.append((evals, Kq.cpu().numpy().copy(), out["loss"].cpu().numpy().copy(), out["grad"].cpu().numpy().copy()))
        evals += 1

    def proposal(hook=None):
        """One proposal on the static tensors: momentum and log u of slice jt, trajectory, Metropolis test, state update."""
        torch.index_select(P_dev, 0, jt, out=P0)
        torch.index_select(lu_dev, 0, jt, out=lu)
        H0 = torch.add(U, torch.linalg.vecdot(P0[0], P0[0]), alpha=0.5)
        Kq.copy_(K); dUq.copy_(dU)
        torch.add(P0[0], dUq, alpha=-0.5 * eps * c_pri, out=Pq)     # first half step
        for _ in range(n_leapfrog):                                 # each step's input depends on the previous gradient
            step()
            if hook is not None:
                hook()
        Pq.add_(dUq, alpha=0.5 * eps * c_pri)                       # the last update was a whole step: back to a half
        Uq = potential_now()
        H1 = torch.add(Uq, torch.linalg.vecdot(Pq, Pq), alpha=0.5)
        ok = lu[0] < H0 - H1                                        # (H1 = inf or nan compares false, as on the host: rejected)
        torch.where(ok[:, None], Kq, K, out=K); torch.where(ok, Uq, U, out=U); torch.where(ok[:, None], dUq, dU, out=dU)
        acc.add_(ok)
        jt.add_(1); pt.add_(1)
        if trace is not None:
            trace.index_copy_(0, pt, K[None])

    Kq.copy_(K); Pq.zero_()
    evaluate()                                                      # evaluation 0: the starting point (also warms the library up)
    note()
    U.copy_(potential_now())
    if not bool(torch.isfinite(U).all()):
        raise ValueError("HMC start point has an indefinite reduced operator")
    dU.copy_(dUq)
    if trace is not None:
        trace[0].copy_(K)
    g = None
    if graph and n_prop > 0:
        state = [t.clone() for t in (K, U, dU, acc, jt, pt)]
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                           # warm-up on a side stream, as torch's graph recipe asks
                proposal()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                proposal()
            for t, t0 in zip((K, U, dU, acc, jt, pt), state):       # (the warm-up moved the state; the capture does not run)
                t.copy_(t0)
        except Exception as exc:                                    # no graph support for this sequence: plain stream order
            import warnings
            warnings.warn(f"hmc: HIP graph capture failed ({exc!r}); the proposals are launched kernel by kernel")
            g = None
            for t, t0 in zip((K, U, dU, acc, jt, pt), state):
                t.copy_(t0)
    done = 0
    while done < n_prop:
        nb = min(B, n_prop - done)
        P_host, lu_host = np.zeros((B, C, n)), np.zeros((B, C))
        for j in range(nb):                                         # the host chain's draws, in its order (overlaps the device's
            for c_, r in enumerate(rngs):                           #  work on the previous block: nothing here waits for it)
                P_host[j, c_] = r.standard_normal(n)
            with np.errstate(divide="ignore"):
                lu_host[j] = np.log(np.array([r.uniform() for r in rngs]))
        P_dev.copy_(torch.from_numpy(P_host)); lu_dev.copy_(torch.from_numpy(lu_host))
        jt.zero_()
        for j in range(nb):
            first = 1 + (done + j) * n_leapfrog                     # evaluation indices of this proposal: first .. first + L - 1
            if record is not None and any(first <= e < first + n_leapfrog for e in record):
                assert evals == first
                proposal(hook=note)
            else:
                g.replay() if g is not None else proposal()
                evals += n_leapfrog
        done += nb
    return HmcResult(K=K.cpu().numpy(), accept=acc.cpu().numpy(), proposals=n_prop, n_evals=evals, recorded=recorded,
                     trace=trace.cpu().numpy() if trace is not None else None, graph=g is not None, fused=False)
