"""Python-side owners of the library handles.  Each ``solve`` is one C-ABI call that runs the
whole batch on the GPU; inputs may be NumPy arrays (copied to / from the device around the
call) or torch CUDA tensors (used in place, results returned as torch tensors on the same
device, launched on torch's current stream)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _ffi
from ._ffi import FomBandDesc, FomBandGradDesc, FomSmallDesc, DeviceBuffer, FomDesc, FomGradDesc, MlpDesc, RomDesc, check, f64, i32, lib


import os as _os
# LDS row cache of the FOM interpreter: (slots+2) x 512 B per wave.  42 -> 22 KiB, 7 waves per CU (154 KiB);
# 72 -> 37 KiB, 4 waves per CU.  FWD_CHUNK = ops per prefetch chunk of the forward stream (8 or 16).
ROW_CACHE_SLOTS = int(_os.environ.get("FINROM_ROW_CACHE", "40"))
FWD_CHUNK = int(_os.environ.get("FINROM_FWD_CHUNK", "8"))
# Parameter vectors of at most FUSED_X_MAX entries (the five / nine fin conductivities) sit in the interpreter's LDS and
# the op stream assembles A itself (no pre-pass); their slots come out of the row cache so that 7 waves still share a CU.
FUSED_X_MAX = int(_os.environ.get("FINROM_FUSED_X_MAX", "16"))
# Batches of at most SMALL_MAX samples use the latency-oriented schedule: one workgroup per sample, 16 lanes per row of L
# (finrom_fom_set_small) -- for GRADIENTS on every mesh, for forward solves only where no band plan is installed (m >= 32):
# since round 3 the band sweep serves small forward batches too (a lone wave of it: 1.7-2.3 ms at m = 12, 4.2-4.4 ms at m = 20,
# against 2.7 and 17.9 ms here).  0 disables it.  Measured cross-over against the interpreter: ~700 samples at m = 12 (value
# vector in LDS, one workgroup per CU), several thousand at m = 20 (value vector in L2).
SMALL_MAX = int(_os.environ.get("FINROM_SMALL_MAX", "512"))
SMALL_MAX_GLOBAL = int(_os.environ.get("FINROM_SMALL_MAX_GLOBAL", "4096"))
# Batches beyond the small-batch schedule use the frontal band sweep (front in registers, csrc/fom_band.hip) when the mesh
# has a band plan and the library has its window sizes; FINROM_NO_BAND=1 keeps the schedule interpreter.
USE_BAND = _os.environ.get("FINROM_NO_BAND") is None


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


class DeviceArray:
    """A [S, d] fp64 array that already lives on the device in a library-owned buffer: the output of one call handed to the
    next without a round trip through the host (sub-fin averages -> reduced solve for NumPy callers)."""

    def __init__(self, buf, shape):
        self.buf, self.shape = buf, tuple(shape)


class _Batch:
    """Uniform view of a [S, d] fp64 batch living on the device."""

    def __init__(self, x, d):
        self.torch = _is_torch(x)
        if isinstance(x, DeviceArray):
            assert x.shape[-1] == d, (x.shape, d)
            self.S = int(np.prod(x.shape[:-1]))
            self.keep = x.buf
            self.ptr = x.buf.ptr
            self.stream = None
        elif self.torch:
            import torch
            if not x.is_cuda or x.dtype != torch.float64:
                raise TypeError("torch inputs must be float64 CUDA tensors")
            x = x.reshape(-1, d).contiguous()
            self.keep = x
            self.S = x.shape[0]
            self.ptr = x.data_ptr()
            self.device = x.device
            self.stream = torch.cuda.current_stream(x.device).cuda_stream
        else:
            a = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, d)
            self.S = a.shape[0]
            self.keep = DeviceBuffer.from_numpy(a)
            self.ptr = self.keep.ptr
            self.stream = None

    # NumPy callers: the outputs of one library call are carved from shared arenas (one zero-fill, ONE copy back for all of
    # them) -- a one-sample call is dominated by driver round trips, not by the kernels (the MAP / HMC pattern of configs[4]).
    _ARENA_MIN = 64 << 10

    def new(self, shape, dtype="f8", zero=True):
        """Allocate an output of the same kind, zero-filled unless the call overwrites all of it (zero=False: one fill kernel
        less per output on the torch path -- they add up in a one-sample call); returns (handle, device pointer)."""
        if self.torch:
            import torch
            t = (torch.zeros if zero else torch.empty)(shape, dtype=torch.float64 if dtype == "f8" else torch.int32, device=self.device)
            return t, t.data_ptr()
        n = int(np.prod(shape)) * (8 if dtype == "f8" else 4)
        arenas = self.__dict__.setdefault("_arenas", [])
        need = (n + 255) // 256 * 256
        if not arenas or arenas[-1][1] + need > arenas[-1][0].nbytes:
            buf = DeviceBuffer(max(need, self._ARENA_MIN if n <= self._ARENA_MIN else need))
            buf.zero()
            arenas.append([buf, 0, None])                  # [device buffer, bytes used, host copy]
        a = arenas[-1]
        off = a[1]
        a[1] += need
        return (len(arenas) - 1, off, n), a[0].ptr + off

    def out(self, obj, shape, dtype="f8"):
        if self.torch:
            return obj
        ia, off, n = obj
        a = self._arenas[ia]
        if a[2] is None:                                    # first read after the call: one copy of what the arena holds
            a[2] = a[0].to_numpy((max(a[1], 8),), np.uint8)
        return a[2][off:off + n].view(np.float64 if dtype == "f8" else np.int32).reshape(shape)


def _sync_if_mixed(b, other):
    """A torch-stream call that also consumed a NumPy operand staged through a pooled DeviceBuffer: the kernels may still be
    running on torch's (non-blocking) stream when that buffer goes back to the library's pool and its next owner writes it on the
    default stream.  The staging buffer remembers the launch stream: it is parked behind an event recorded there
    (finrom_free_async) and the next owner waits for that event -- no host synchronisation here."""
    if b.torch and not other.torch and isinstance(other.keep, DeviceBuffer):
        other.keep.used_on(b.stream)


def _csr_rows(M):
    M = sp.csr_matrix(M)
    M.sort_indices()
    return M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64)


class FomEngine:
    """Batched ``A(x) w = F`` + QoI (finrom_fom_*).  ``c0``/``W`` define the sparse-affine
    value map on the CSR pattern of A (see include/finrom.h)."""

    def __init__(self, plan, c0_csr, W_csr, rhs, B_obs, pattern=None, ops=None):
        self.plan = plan
        self.band = None
        self.n = plan.n
        W_csr = sp.csr_matrix(W_csr)
        self.xdim = W_csr.shape[1]
        self._W, self._pattern, self._grad = W_csr, pattern, False
        c0, aptr, aidx, aw = plan.entry_table(c0_csr, W_csr)
        Bp = sp.csr_matrix(np.asarray(B_obs)[:, plan.perm]) if not sp.issparse(B_obs) else sp.csr_matrix(B_obs)[:, plan.perm]
        self._Bp = Bp
        optr, oidx, ow = _csr_rows(Bp)
        self.n_obs = Bp.shape[0]
        keep = []

        def I(a):
            a, p = i32(a); keep.append(a); return p

        def D(a):
            a, p = f64(a); keep.append(a); return p

        fused = self.xdim <= min(FUSED_X_MAX, 16) and ROW_CACHE_SLOTS - self.xdim >= 8
        slots = ROW_CACHE_SLOTS - (self.xdim if fused else 0)
        streams = plan.op_streams(slots, np.asarray(rhs)[plan.perm], FWD_CHUNK, (c0, aptr, aidx, aw) if fused else None)
        self.cache_slots, self.fused, self._streams = slots, fused, streams
        fk, fa, fb, fd = streams["fwd"]; bk, ba, bb, bd = streams["bwd"]
        d = FomDesc(n=plan.n, nnzL=plan.nnzL, xdim=self.xdim, n_obs=self.n_obs, nasm=len(aidx),
                    n_alist=len(streams["a_list"]), cache_slots=slots, fwd_chunk=FWD_CHUNK, nops_fwd=len(fk), nops_bwd=len(bk),
                    a_list=I(streams["a_list"]), asm_c0=D(c0), asm_ptr=I(aptr), asm_idx=I(aidx), asm_w=D(aw),
                    rhs=D(np.asarray(rhs)[plan.perm]),
                    fwd_kind=I(fk), fwd_a=I(fa), fwd_b=I(fb), fwd_d=I(fd),
                    bwd_kind=I(bk), bwd_a=I(ba), bwd_b=I(bb), bwd_d=I(bd),
                    obs_ptr=I(optr), obs_idx=I(oidx), obs_w=D(ow), perm=I(plan.perm),
                    n_imm=len(streams["imm"]), imm=D(streams["imm"]))
        h = C.c_void_p()
        check(lib().finrom_fom_create(C.byref(d), C.byref(h)), "finrom_fom_create")
        self._h = h
        if SMALL_MAX > 0:
            lpf, lrf, lpb, lrb = plan.level_sets()
            in_lds = (plan.nnzL + 3 * plan.n + self.xdim) * 8 <= 156 * 1024          # the library's own criterion
            sd = FomSmallDesc(small_max=SMALL_MAX if in_lds else SMALL_MAX_GLOBAL, npairs=plan.npairs, nasm=len(aidx), nlev_f=len(lpf) - 1, nlev_b=len(lpb) - 1,
                              row_ptr=I(plan.row_ptr), ent_col=I(plan.ent_col),
                              pair_ptr=I(plan.pair_ptr), pair_a=I(plan.pair_a), pair_b=I(plan.pair_b),
                              asm_c0=D(c0), asm_ptr=I(aptr), asm_idx=I(aidx), asm_w=D(aw),
                              col_ptr=I(plan.col_ptr), col_ent=I(plan.col_ent), col_row=I(plan.col_row),
                              lev_ptr_f=I(lpf), lev_rows_f=I(lrf), lev_ptr_b=I(lpb), lev_rows_b=I(lrb))
            check(lib().finrom_fom_set_small(self._h, C.byref(sd)), "finrom_fom_set_small")
        if USE_BAND and ops is not None:
            self._enable_band(ops, c0_csr, W_csr, rhs, B_obs)

    @staticmethod
    def band_descriptor(bp, xdim, c0_csr, W_csr, rhs, B_obs):
        """finrom_fom_band_desc of band plan `bp` for one operator table -> (descriptor, arrays it borrows, physical slots)."""
        c0, ptr, idx, w = bp.ab_table(c0_csr, W_csr)
        abmap, c0p, ptrp, idxp, wp = bp.compact_slots(c0, ptr, idx, w)      # logical -> physical value slots (duplicates shared)
        F = np.asarray(rhs, dtype=np.float64)
        Fg = np.zeros(bp.G)
        for seg in bp.fin_segs + [bp.post_seg]:
            Fg[seg.g0:seg.g0 + seg.npiv] = F[bp.perm[seg.e0:seg.e0 + seg.npiv]]
        Bp = sp.csr_matrix(np.asarray(B_obs)[:, bp.perm]) if not sp.issparse(B_obs) else sp.csr_matrix(B_obs)[:, bp.perm]
        optr, oidx, ow = _csr_rows(Bp)
        nif = bp.q + 1
        schur = np.asarray([[abmap[off] for _, _, off in tg] for tg in bp.schur_target], np.int32)
        qo = FomEngine.qoi_only_tables(bp, Bp, Fg)
        keep = []

        def I(a):
            a, p = i32(a); keep.append(a); return p

        def D(a):
            a, p = f64(a); keep.append(a); return p
        d = FomBandDesc(NSF=bp.NSF, NSP=bp.NSP, NX=bp.NX, nfins=bp.nfins, npf=bp.npf, nif=nif, npost=bp.npost, nAB=len(c0p),
                        nterms=len(idxp), nLx=bp.nLx, ab_c0=D(c0p), ab_ptr=I(ptrp), ab_idx=I(idxp), ab_w=D(wp), abmap=I(abmap[:3 * bp.G]),
                        Fg=D(Fg), act=I(bp.act), lx_ptr=I(bp.lx_ptr), ent_extra=I(bp.ent_extra),
                        ecp_ptr=I(bp.ecp_ptr), ecp_slot=I(bp.ecp_slot), ecp_off=I(abmap[bp.ecp_off] if len(bp.ecp_off) else bp.ecp_off),
                        schur_off=I(schur), iface_elim=I(bp.iface_elim), perm=I(bp.perm),
                        obs_ptr=I(optr), obs_idx=I(oidx), obs_w=D(ow))
        if qo is not None:
            FgQ, row_fin, qptr, qidx, qw = qo
            d.qoi_FgQ, d.qoi_row_fin, d.qoi_obs_ptr, d.qoi_obs_idx, d.qoi_obs_w = D(FgQ), I(row_fin), I(qptr), I(qidx), D(qw)
        return d, keep, len(c0p)

    @staticmethod
    def qoi_only_tables(bp, Bp, Fg):
        """Tables of the band sweep's QoI-only form (include/finrom.h, finrom_fom_band_desc::qoi_*): every observation row is
        split into the weights on ONE fin's segment nodes (own + interface: they ride through that fin's forward sweep as its
        right-hand side) and a post-only remainder.  None when the observation operator does not split that way (a row with
        weights on the own nodes of two fins, two rows on one fin -- e.g. the 40 point observations of external_obs -- or a load
        on the fins): the full sweep then serves every call."""
        Bp = sp.csr_matrix(Bp)
        n_obs = Bp.shape[0]
        npf, nfins, nif = bp.npf, bp.nfins, bp.q + 1
        ntot, post_e0 = npf + nif, nfins * npf
        if np.any(Fg[:nfins * ntot] != 0.0):
            return None
        row_fin = np.full(n_obs, -1, np.int32)
        fin_row = np.full(nfins, -1, np.int64)
        for o in range(n_obs):
            idx = Bp.indices[Bp.indptr[o]:Bp.indptr[o + 1]]
            fins = np.unique(idx[idx < post_e0] // npf)
            if len(fins) > 1:
                return None
            if len(fins) == 1:
                if fin_row[fins[0]] >= 0:
                    return None
                row_fin[o], fin_row[fins[0]] = fins[0], o
        FgQ = np.array(Fg, dtype=np.float64, copy=True)
        qptr, qidx, qw = [0], [], []
        for o in range(n_obs):
            idx = Bp.indices[Bp.indptr[o]:Bp.indptr[o + 1]]; val = Bp.data[Bp.indptr[o]:Bp.indptr[o + 1]]
            f = int(row_fin[o])
            iface = {int(e): t for t, e in enumerate(bp.iface_elim[f])} if f >= 0 else {}
            for e, v in zip(idx, val):
                e = int(e)
                if e < post_e0:
                    FgQ[f * ntot + (e - f * npf)] = v
                elif e in iface:
                    FgQ[f * ntot + npf + iface[e]] = v
                else:
                    qidx.append(e); qw.append(v)
            qptr.append(len(qidx))
        return FgQ, row_fin, np.asarray(qptr, np.int32), np.asarray(qidx, np.int32), np.asarray(qw, np.float64)

    def _enable_band(self, ops, c0_csr, W_csr, rhs, B_obs):
        """Install the frontal band sweep (finrom_fom_set_band) when the mesh has a band plan and the library was built with
        its window sizes; otherwise the handle keeps the interpreter."""
        bp = ops.band_plan()
        if bp is None:
            return
        d, keep, nslots = self.band_descriptor(bp, self.xdim, c0_csr, W_csr, rhs, B_obs)
        rc = lib().finrom_fom_set_band(self._h, C.byref(d))
        if rc == -4:                                     # FINROM_ERR_UNSUPPORTED: window sizes not built in -> interpreter
            return
        check(rc, "finrom_fom_set_band")
        self.band = bp
        self._B_band = sp.csr_matrix(np.asarray(B_obs)[:, bp.perm]) if not sp.issparse(B_obs) else sp.csr_matrix(B_obs)[:, bp.perm]
        self.band_slots = nslots            # physical value slots per sample (bench: algorithmic bytes)
        self.band_qoi_only = bool(d.qoi_FgQ) and _os.environ.get("FINROM_BAND_NO_QOI_ONLY") is None      # calls without w: QoI-only form

    def solve(self, X, want_w=False):
        b = _Batch(X, self.xdim)
        S = b.S
        qoi, qp = b.new((S, self.n_obs))
        info, ip = b.new((S,), "i4")
        w, wp = (b.new((S, self.n)) if want_w else (None, None))
        check(lib().finrom_fom_solve(self._h, b.ptr, S, qp, wp, ip, b.stream), "finrom_fom_solve")
        return {"qoi": b.out(qoi, (S, self.n_obs)), "w": b.out(w, (S, self.n)) if want_w else None,
                "info": b.out(info, (S,), "i4")}

    def last_path(self):
        """Name of the schedule the most recent solve / gradient on this handle launched (finrom_fom_last_path):
        'small_lds', 'small_global', 'interpreter', 'band_registers', 'band_lds_4wave' ('none' before the first call)."""
        rc = lib().finrom_fom_last_path(self._h)
        if rc < 0:
            check(rc, "finrom_fom_last_path")
        return _ffi.FOM_PATHS[rc]

    def set_small_max(self, small_max):
        """Move the batch-size threshold of the small-batch schedule (0: every batch takes the throughput path)."""
        check(lib().finrom_fom_set_small_max(self._h, int(small_max)), "finrom_fom_set_small_max")

    def solve_rhs(self, X, rhs):
        """X [S, xdim], rhs [S, nrhs, n] (dof order) -> dict(out [S, nrhs, n] = A(x_s)^-1 rhs, info [S]) (finrom_fom_solve_rhs)."""
        b = _Batch(X, self.xdim)
        S = b.S
        rhs = rhs if _is_torch(rhs) else np.ascontiguousarray(rhs, dtype=np.float64)
        nrhs = int(rhs.shape[-2]) if rhs.ndim == 3 else 1
        rb = _Batch(rhs, nrhs * self.n)
        assert rb.S == S, (rb.S, S)
        out, op = b.new((S, nrhs, self.n), zero=False) if b.torch else b.new((S, nrhs, self.n))
        info, ip = b.new((S,), "i4")
        check(lib().finrom_fom_solve_rhs(self._h, b.ptr, S, rb.ptr, nrhs, op, ip, b.stream), "finrom_fom_solve_rhs")
        _sync_if_mixed(b, rb)
        return {"out": b.out(out, (S, nrhs, self.n)), "info": b.out(info, (S,), "i4")}

    def _enable_gradient(self):
        """One-time tables of the adjoint gradient (finrom_fom_set_gradient)."""
        if self._grad:
            return
        if self._pattern is None:
            raise ValueError("FomEngine needs the CSR pattern of A for gradients")
        from .symbolic import build_resolve_stream
        plan, n = self.plan, self.n
        rk, ra, rb, rd = build_resolve_stream(plan, plan.nnzL + 2 * n)
        Bt = sp.csr_matrix(self._Bp.T)                       # [n(permuted) x n_obs]
        Bt.sort_indices()
        indptr, indices = np.asarray(self._pattern[0]), np.asarray(self._pattern[1])
        e_row = np.repeat(np.arange(n), np.diff(indptr))
        Wt = sp.csc_matrix(self._W)                          # columns = parameters
        g_ptr = Wt.indptr.astype(np.int32)
        ent = Wt.indices
        g_a = plan.iperm[e_row[ent]].astype(np.int32)        # row of the entry, permuted
        g_b = plan.iperm[indices[ent]].astype(np.int32)      # column of the entry, permuted
        keep = []

        def I(a):
            a, p = i32(a); keep.append(a); return p

        def D(a):
            a, p = f64(a); keep.append(a); return p

        d = FomGradDesc(nops_res=len(rk), res_kind=I(rk), res_a=I(ra), res_b=I(rb), res_d=I(rd),
                        bt_ptr=I(Bt.indptr), bt_obs=I(Bt.indices), bt_w=D(Bt.data),
                        g_ptr=I(g_ptr), g_a=I(g_a), g_b=I(g_b), g_w=D(Wt.data))
        check(lib().finrom_fom_set_gradient(self._h, C.byref(d)), "finrom_fom_set_gradient")
        if self.band is not None:
            # the same tables over the band plan's elimination indices: large batches solve the adjoint on the band layout
            bp = self.band
            Btb = sp.csr_matrix(self._B_band.T)
            Btb.sort_indices()
            gb = FomBandGradDesc(bt_ptr=I(Btb.indptr), bt_obs=I(Btb.indices), bt_w=D(Btb.data), g_ptr=I(g_ptr),
                                 g_a=I(bp.iperm[e_row[ent]]), g_b=I(bp.iperm[indices[ent]]), g_w=D(Wt.data))
            check(lib().finrom_fom_set_band_gradient(self._h, C.byref(gb)), "finrom_fom_set_band_gradient")
        self._grad = True

    def gradient(self, X, data):
        """X [S, xdim], data [n_obs] or [S, n_obs] -> dict(grad [S, xdim], J [S], qoi, info)
        (Fin.gradient, fom/forward_solve.py:293-322, for a batch)."""
        self._enable_gradient()
        b = _Batch(X, self.xdim)
        S = b.S
        data = data if _is_torch(data) else np.ascontiguousarray(data, dtype=np.float64)
        per_sample = 1 if data.ndim == 2 else 0
        db = _Batch(data, self.n_obs)
        grad, gp = b.new((S, self.xdim)); J, Jp = b.new((S,)); qoi, qp = b.new((S, self.n_obs)); info, ip = b.new((S,), "i4")
        check(lib().finrom_fom_gradient(self._h, b.ptr, db.ptr, per_sample, S, gp, Jp, qp, ip, b.stream), "finrom_fom_gradient")
        _sync_if_mixed(b, db)
        return {"grad": b.out(grad, (S, self.xdim)), "J": b.out(J, (S,)), "qoi": b.out(qoi, (S, self.n_obs)),
                "info": b.out(info, (S,), "i4")}

    def close(self):
        if getattr(self, "_h", None):
            _ffi.destroy_handle("finrom_fom_destroy", self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RomEngine:
    """Batched LSPG reduced solve (finrom_rom_*).

    ``terms``: list of (theta_index, sparse-or-dense [n, r] matrix Psi_p = A_p Phi); theta_index 0 is the
    constant (Robin) term, 1..P the parameters.  Only the non-zero rows of each Psi_p are shipped."""

    @staticmethod
    def pack_terms(n, r, terms):
        """(row_ptr [n + 1], term_p [nterms], term_val [nterms, r]) of finrom_rom_desc: per row of psi its non-zero terms."""
        rows = [[] for _ in range(n)]
        for p, M in terms:
            M = np.asarray(M)
            nzr = np.nonzero(np.abs(M).sum(1))[0]
            for j in nzr:
                rows[j].append((p, M[j]))
        row_ptr = np.zeros(n + 1, np.int32)
        term_p, term_val = [], []
        for j in range(n):
            for p, v in rows[j]:
                term_p.append(p); term_val.append(v)
            row_ptr[j + 1] = len(term_p)
        return row_ptr, term_p, np.asarray(term_val, dtype=np.float64).reshape(len(term_p), r)

    def __init__(self, n, r, P, terms, rhs, obs_phi):
        self.n, self.r, self.P = n, r, P
        row_ptr, term_p, tv = self.pack_terms(n, r, terms)
        self.nterms = len(term_p)
        obs_phi = np.ascontiguousarray(obs_phi, dtype=np.float64)
        self.n_obs = obs_phi.shape[0]
        a1, p1 = i32(row_ptr); a2, p2 = i32(term_p); a3, p3 = f64(tv); a4, p4 = f64(rhs); a5, p5 = f64(obs_phi)
        d = RomDesc(n=n, r=r, P=P, n_obs=self.n_obs, nterms=self.nterms, row_ptr=p1, term_p=p2, term_val=p3,
                    rhs=p4, obs_phi=p5)
        h = C.c_void_p()
        check(lib().finrom_rom_create(C.byref(d), C.byref(h)), "finrom_rom_create")
        self._h = h

    def solve(self, theta, want_state=False, want_w=True):
        """want_w=False: only the reduced QoI comes back; bases wider than 96 then factor and solve inside the projection
        kernel's registers (no A_r in memory)."""
        b = _Batch(theta, self.P)
        S, r = b.S, self.r
        w_r, wp = b.new((S, r)) if want_w else (None, None)
        qoi, qp = b.new((S, self.n_obs))
        info, ip = b.new((S,), "i4")
        A_r, Ap = (b.new((S, r, r)) if want_state else (None, None))
        B_r, Bp = (b.new((S, r)) if want_state else (None, None))
        check(lib().finrom_rom_solve(self._h, b.ptr, S, wp, qp, Ap, Bp, ip, b.stream), "finrom_rom_solve")
        out = {"qoi_r": b.out(qoi, (S, self.n_obs)), "info": b.out(info, (S,), "i4")}
        if want_w:
            out["w_r"] = b.out(w_r, (S, r))
        if want_state:
            out["A_r"] = b.out(A_r, (S, r, r)); out["B_r"] = b.out(B_r, (S, r))
        return out

    def set_gradient_blocks(self, pairs, G):
        """pairs: list of (p, i); G [npairs, r, r] with G[t] = (A_p Phi)^T (A_i Phi)  (finrom_rom_set_gradient)."""
        pp, pi = i32([p for p, _ in pairs]), i32([i for _, i in pairs])
        Gt = np.ascontiguousarray(np.transpose(np.asarray(G, dtype=np.float64), (0, 2, 1)))   # column by column
        check(lib().finrom_rom_set_gradient(self._h, len(pairs), pp[1], pi[1], Gt.ctypes.data_as(_ffi.c_f64p)),
              "finrom_rom_set_gradient")
        self._has_grad = True

    def set_gram_blocks(self, pairs, G):
        """pairs: list of (p, q), 0 <= p <= q <= P; G [npairs, r, r] symmetric blocks Psi_p^T Psi_q (+ transpose for p != q)
        (finrom_rom_set_gram)."""
        pp, pq = i32([p for p, _ in pairs]), i32([q for _, q in pairs])
        G = np.ascontiguousarray(G, dtype=np.float64)
        check(lib().finrom_rom_set_gram(self._h, len(pairs), pp[1], pq[1], G.ctypes.data_as(_ffi.c_f64p)), "finrom_rom_set_gram")
        self.gram_pairs = len(pairs)

    def set_projection(self, mode):
        """'direct' (per-sample psi^T psi on MFMA, the reference's contraction) or 'offline_online' (precomputed blocks)."""
        code = {"direct": 0, "offline_online": 1}[mode]
        check(lib().finrom_rom_set_projection(self._h, code), "finrom_rom_set_projection")
        self.projection = mode

    def grad(self, theta, data):
        """theta [S, P], data [n_obs] or [S, n_obs] -> dict(J [S], g [S, P], w_r, qoi_r, info)."""
        b = _Batch(theta, self.P)
        S = b.S
        data = np.ascontiguousarray(data, dtype=np.float64) if not _is_torch(data) else data
        per_sample = 1 if data.ndim == 2 else 0
        db = _Batch(data, self.n_obs)
        J, Jp = b.new((S,)); g, gp = b.new((S, self.P)); w_r, wp = b.new((S, self.r))
        qoi, qp = b.new((S, self.n_obs)); info, ip = b.new((S,), "i4")
        check(lib().finrom_rom_grad(self._h, b.ptr, db.ptr, per_sample, S, Jp, gp, wp, qp, ip, b.stream), "finrom_rom_grad")
        _sync_if_mixed(b, db)
        return {"J": b.out(J, (S,)), "g": b.out(g, (S, self.P)), "w_r": b.out(w_r, (S, self.r)),
                "qoi_r": b.out(qoi, (S, self.n_obs)), "info": b.out(info, (S,), "i4")}

    def close(self):
        if getattr(self, "_h", None):
            _ffi.destroy_handle("finrom_rom_destroy", self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SubfinAverager:
    """theta = S k on the device (finrom_subfin_avg)."""

    def __init__(self, Sop):
        Sop = np.ascontiguousarray(Sop, dtype=np.float64)
        self.P, self.n = Sop.shape
        self._S = DeviceBuffer.from_numpy(Sop)

    def __call__(self, K):
        b = _Batch(K, self.n)
        th, tp = b.new((b.S, self.P))
        check(lib().finrom_subfin_avg(self._S.ptr, self.P, self.n, b.ptr, b.S, tp, b.stream), "finrom_subfin_avg")
        self._S.used_on(b.stream)
        return b.out(th, (b.S, self.P))

    def on_device(self, K):
        """NumPy K [S, n] -> theta as a DeviceArray (no copy back): the next library call consumes it in place."""
        b = _Batch(K, self.n)
        buf = DeviceBuffer(max(b.S * self.P * 8, 8))
        check(lib().finrom_subfin_avg(self._S.ptr, self.P, self.n, b.ptr, b.S, buf.ptr, None), "finrom_subfin_avg")
        out = DeviceArray(buf, (b.S, self.P))
        out._src = b.keep                                  # (the input's staging buffer lives until theta has been consumed)
        return out


class FieldSampler:
    """k = exp(0.5 * xi @ U) on the device (finrom_sampler_*)."""

    def __init__(self, U):
        U = np.ascontiguousarray(U, dtype=np.float64)
        self.n = U.shape[0]
        h = C.c_void_p()
        check(lib().finrom_sampler_create(U.ctypes.data_as(_ffi.c_f64p), self.n, C.byref(h)), "finrom_sampler_create")
        self._h = h

    def __call__(self, xi):
        b = _Batch(xi, self.n)
        k, kp = b.new((b.S, self.n))
        check(lib().finrom_sampler_draw(self._h, b.ptr, b.S, kp, b.stream), "finrom_sampler_draw")
        return b.out(k, (b.S, self.n))

    def draw(self, seed, first, S, like=None, want_xi=False):
        """S fields of the stream `seed` starting at GLOBAL sample index `first`, xi drawn on the device (Philox keyed by the
        global index: independent of how a dataset is sharded).  like: a torch CUDA tensor (-> torch outputs on its device /
        current stream) or None (-> NumPy).  Returns k [S, n] (and xi [S, n] if want_xi)."""
        b = _Batch(like if like is not None else np.zeros((0, self.n)), self.n)
        k, kp = b.new((S, self.n))
        xi, xp = b.new((S, self.n)) if want_xi else (None, None)
        check(lib().finrom_sampler_draw_seeded(self._h, int(seed), int(first), int(S), kp, xp, b.stream), "finrom_sampler_draw_seeded")
        out = b.out(k, (S, self.n))
        return (out, b.out(xi, (S, self.n))) if want_xi else out

    def close(self):
        if getattr(self, "_h", None):
            _ffi.destroy_handle("finrom_sampler_destroy", self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceErrorModel:
    """The learned error model on the device (finrom_mlp_*): the weights of a deep_learning/dl_model.py::ResBnFcModel (the
    stand-in for the reference's Keras res_bn_fc_model, dl_model.py:149-176), batch normalisation folded into scale / shift in
    fp32 exactly as the host model does."""

    def __init__(self, model):
        from .deep_learning.dl_model import BN_EPS
        layers = list(model.units) + [model.head]
        f32 = lambda a_: np.ascontiguousarray(a_, dtype=np.float32)
        scale = np.stack([u["gamma"] / np.sqrt(u["var"] + np.float32(BN_EPS)) for u in layers]).astype(np.float32)
        shift = np.stack([u["beta"] - u["mean"] * s_ for u, s_ in zip(layers, scale)]).astype(np.float32)
        L, nw = len(model.units), model.n_weights
        arrs = {"W0": f32(model.W0), "b0": f32(model.b0), "scale": f32(scale), "shift": f32(shift),
                "W": f32(np.stack([u["W"] for u in model.units]) if L else np.zeros((0, nw, nw))),
                "b": f32(np.stack([u["b"] for u in model.units]) if L else np.zeros((0, nw))),
                "Wh": f32(model.head["W"]), "bh": f32(model.head["b"])}
        self.n_in, self.n_out = model.n_in, model.n_out
        d = MlpDesc(n_in=model.n_in, n_w=nw, n_layers=L, n_out=model.n_out,
                    **{k: v.ctypes.data_as(_ffi.c_f32p) for k, v in arrs.items()})
        h = C.c_void_p()
        check(lib().finrom_mlp_create(C.byref(d), C.byref(h)), "finrom_mlp_create")
        self._h = h

    def predict(self, K):
        b = _Batch(K, self.n_in)
        e, ep = b.new((b.S, self.n_out))
        check(lib().finrom_mlp_predict(self._h, b.ptr, b.S, ep, b.stream), "finrom_mlp_predict")
        return b.out(e, (b.S, self.n_out))

    def close(self):
        if getattr(self, "_h", None):
            _ffi.destroy_handle("finrom_mlp_destroy", self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def romml_grad(rom, mlp, Sop_buf, K, data):
    """finrom_romml_grad: value and gradient of the ROM + learned-error misfit for a batch of nodal fields K [S, n] in ONE
    library call (sub-fin averages, network forward, ROM adjoint against data - e_NN, network backward + chain rule).
    -> dict(grad [S, n], loss [S], qoi_r, e_NN, info)."""
    n, n_obs = mlp.n_in, mlp.n_out
    b = _Batch(K, n)
    S = b.S
    data = data if _is_torch(data) else np.ascontiguousarray(data, dtype=np.float64)
    per_sample = 1 if data.ndim == 2 else 0
    db = _Batch(data, n_obs)
    grad, gp = b.new((S, n), zero=False); loss, lp = b.new((S,), zero=False); q, qp = b.new((S, n_obs), zero=False)
    e, ep = b.new((S, n_obs), zero=False); info, ip = b.new((S,), "i4", zero=False)      # (finrom_romml_grad overwrites info)
    check(lib().finrom_romml_grad(rom._h, mlp._h, Sop_buf.ptr, b.ptr, db.ptr, per_sample, S, gp, lp, qp, ep, ip, b.stream),
          "finrom_romml_grad")
    Sop_buf.used_on(b.stream)
    _sync_if_mixed(b, db)
    return {"grad": b.out(grad, (S, n)), "loss": b.out(loss, (S,)), "qoi_r": b.out(q, (S, n_obs)), "e_NN": b.out(e, (S, n_obs)),
            "info": b.out(info, (S,), "i4")}


def device_sub(a, b_):
    """a - b elementwise on the device (generate_fin_dataset.py:99)."""
    ba = _Batch(a, 1)
    bb = _Batch(b_, 1)
    out, op = ba.new((ba.S,))
    check(lib().finrom_sub(ba.ptr, bb.ptr, ba.S, op, ba.stream), "finrom_sub")
    res = ba.out(out, (ba.S,))
    return res.reshape(a.shape)
