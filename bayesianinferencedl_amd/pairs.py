"""Fused FOM + ROM pair evaluation for a batch of conductivity samples: the body of the
reference's dataset loop (deep_learning/generate_fin_dataset.py:83-100) as ONE C-ABI call
(finrom_solve_pairs): the sparse FOM solve runs on the caller's stream while the sub-fin
averaging and the LSPG reduced solve run concurrently on a library-owned stream."""
from __future__ import annotations

import numpy as np

from ._ffi import check, lib
from .engine import SubfinAverager, _Batch
from .fom.forward_solve import Fin
from .rom.averaged_affine_ROM import AffineROMFin


class FinPairSolver:
    """params: 'field' (x = nodal conductivity, n values -- the live reference loop),
    'nine' / 'five' (x = per-fin conductivities interpolated by nine_/five_param_to_function)."""

    def __init__(self, V, phi, external_obs=False, params="field", solver=None, solver_r=None):
        self.V = V
        self.params = params
        self.solver = solver if solver is not None else Fin(V, external_obs)
        self.solver_r = solver_r if solver_r is not None else AffineROMFin(V, None, phi, external_obs)
        ops = V.operators()
        lift = {"field": None, "nine": ops.N9, "five": ops.N9 @ ops.E59}[params]
        # theta = subfin_avg_op(k_h) with k_h = lift @ x   (rom/averaged_affine_ROM.py:272, 404-418)
        self._avg = SubfinAverager(ops.S if lift is None else ops.S @ lift)
        self.xdim = ops.n if lift is None else lift.shape[1]
        self.n_obs = self.solver.n_obs

    def solve_pairs(self, X, want_w=False, want_w_r=False):
        """want_w / want_w_r: also return the FOM state / the reduced coefficients (the dataset loop keeps neither:
        generate_fin_dataset.py:93-100 stores the parameters and qoi - qoi_r only; without w_r, bases wider than 96 factor and
        solve inside the projection kernel's registers)."""
        fom = self.solver._engine("field" if self.params == "field" else self.params)
        rom = self.solver_r._rom
        b = _Batch(X, self.xdim)
        S = b.S
        # (every output but the flags is written in full by the call -- failed samples as NaN: no fill kernels in front of it;
        # between two steps of the headline they were 65 us of the 0.33 ms the GPU waited)
        qoi, qp = b.new((S, self.n_obs), zero=False); qoi_r, qrp = b.new((S, self.n_obs), zero=False); err, ep = b.new((S, self.n_obs), zero=False)
        w_r, wrp = b.new((S, rom.r), zero=False) if want_w_r else (None, None); theta, tp = b.new((S, rom.P), zero=False); info, ip = b.new((S,), "i4")
        w, wp = (b.new((S, fom.n), zero=False) if want_w else (None, None))
        check(lib().finrom_solve_pairs(fom._h, rom._h, self._avg._S.ptr, b.ptr, S, qp, qrp, ep, wp, wrp, tp, ip, b.stream),
              "finrom_solve_pairs")
        self._avg._S.used_on(b.stream)
        return {"qoi": b.out(qoi, (S, self.n_obs)), "qoi_r": b.out(qoi_r, (S, self.n_obs)),
                "err": b.out(err, (S, self.n_obs)), "w": b.out(w, (S, fom.n)) if want_w else None,
                "w_r": b.out(w_r, (S, rom.r)) if want_w_r else None, "theta": b.out(theta, (S, rom.P)), "info": b.out(info, (S,), "i4")}
