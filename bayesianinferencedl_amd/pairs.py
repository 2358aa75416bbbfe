"""Fused FOM + ROM pair evaluation for a batch of conductivity samples: the body of the
reference's dataset loop (deep_learning/generate_fin_dataset.py:83-100) as four C-ABI calls
(finrom_fom_solve, finrom_subfin_avg, finrom_rom_solve, finrom_sub), all on the device."""
from __future__ import annotations

import numpy as np

from .engine import SubfinAverager, device_sub
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

    def solve_pairs(self, X, want_w=False):
        fom = self.solver.forward_batch(X, want_w=want_w, params=None if self.params == "field" else self.params)
        theta = self._avg(X)
        rom = self.solver_r.forward_nine_param_reduced_batch(theta)
        err = device_sub(fom["qoi"], rom["qoi_r"])
        return {"qoi": fom["qoi"], "qoi_r": rom["qoi_r"], "err": err, "w": fom["w"], "w_r": rom["w_r"],
                "theta": theta, "info": fom["info"] | rom["info"]}
