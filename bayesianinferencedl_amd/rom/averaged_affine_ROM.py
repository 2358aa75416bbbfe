"""MI355X-native ``AffineROMFin``: the sub-fin-averaged affine reduced-order model behind
the reference's call surface (rom/averaged_affine_ROM.py::AffineROMFin).

Reference methods mirrored (file rom/averaged_affine_ROM.py): __init__ :57-235 (forward-path
subset), forward :237-258, forward_reduced :260-276, forward_nine_param_reduced :278-310,
qoi :312-320, qoi_reduced :323-333, grad_reduced :335-356, grad_romml :358-396, set_data/set_dl_model :398-402, subfin_avg_op :404-418,
observation_operator :420-445.  The reduced system is least-squares Petrov-Galerkin
(psi = A Phi, A_r = psi^T psi, :295-297), NOT Galerkin."""
from __future__ import annotations

import os
import time

import numpy as np
import scipy.sparse as sp

from ..fem import BIOT, Function, as_nodal, deterministic_blas
from ..engine import DeviceErrorModel, FomEngine, RomEngine, SubfinAverager, romml_grad
from ..fom.forward_solve import _plan_for, external_observation_matrix


class AffineROMFin:
    def __init__(self, V, err_model, phi, external_obs=False, projection=None):
        with deterministic_blas():
            self._init(V, err_model, phi, external_obs, projection)

    def _init(self, V, err_model, phi, external_obs, projection):
        self.fwd_time = 0.0
        self.rom_grad_time = 0.0
        self.romml_grad_time = 0.0
        self.romml_grad_time_dl = 0.0
        self.num_params = 9
        self.phi = np.ascontiguousarray(phi, dtype=np.float64)
        (self.n, self.n_r) = self.phi.shape
        self.V = V
        self.dofs = len(V.dofmap().dofs())
        if self.n != self.dofs:
            raise ValueError(f"basis has {self.n} rows but the space has {self.dofs} dofs")
        self.Bi = BIOT
        self.ops = ops = V.operators()
        self.B = ops.F.copy()                                   # :171,178
        if external_obs:
            self.n_obs = 40
            self.B_obs = external_observation_matrix(ops, self.n_obs)
        else:
            self.n_obs = 9
            self.B_obs = self.observation_operator()
        self.dsigma_dk = self.observation_operator()            # :210
        self.B_obs_phi = np.dot(self.B_obs, self.phi)           # :212
        # :215-220  Psi_i = A_i Phi (the reference also stores the dense 9 x n x n A_i: 150 MB
        # at n = 1446; kept sparse here -- see .dA_dsigmak)
        self._A_sub = [ops.csr(ops.sub_vals[i]) for i in range(9)]
        self.dA_dsigmak_phi = np.stack([A @ self.phi for A in self._A_sub])
        self.dl_model = None
        self._dev_model = None
        self.data = None
        self.psi = None
        self._state = None                                      # (A_r, B_r) of the last scalar solve, computed on demand
        self._theta = None
        robin_phi = ops.csr(ops.robin_vals) @ self.phi
        terms = [(0, robin_phi)] + [(i + 1, self.dA_dsigmak_phi[i]) for i in range(9)]
        self._rom = RomEngine(self.n, self.n_r, 9, terms, ops.F, self.B_obs_phi)
        self._avg = SubfinAverager(ops.S)
        self._plan = _plan_for(V)
        self._fom = None
        self._grad_ready = False
        self._psi_tables = [robin_phi] + [self.dA_dsigmak_phi[i] for i in range(9)]
        self.projection = "direct"
        self.set_projection(projection or os.environ.get("FINROM_PROJECTION", "direct"))
        self.set_dl_model(err_model)

    def set_projection(self, mode):
        """How the reduced operator is formed per sample.  'direct' (default): psi = A(theta) Phi, A_r = psi^T psi on the
        fp64 matrix cores, the contraction the reference executes (:291-297).  'offline_online': A_r = sum theta_p theta_q G_pq
        from blocks G_pq = Psi_p^T Psi_q precomputed here once (the reference precomputes Psi_p, :215-220); same results to
        round-off, ~50x fewer flops per sample."""
        if mode not in ("direct", "offline_online"):
            raise ValueError(f"unknown projection {mode!r}")
        if mode == "offline_online" and not getattr(self, "_gram_ready", False):
            pairs, G = [], []
            for p in range(10):
                for q in range(p, 10):
                    with deterministic_blas():
                        M = self._psi_tables[p].T @ self._psi_tables[q]
                    if p == q or np.any(M):
                        pairs.append((p, q)); G.append(M if p == q else M + M.T)
            self._rom.set_gram_blocks(pairs, np.stack(G))
            self._gram_ready = True
        self._rom.set_projection(mode)
        self.projection = mode

    # The reference leaves _A_r / _B_r behind after forward_nine_param_reduced (:296-297) for its gradient methods; the
    # gradients here recompute what they need on the device, so the dense state is only materialised when somebody reads it
    # (one extra library call with the A_r / B_r outputs requested).
    def _reduced_state(self):
        if self._state is None and self._theta is not None:
            res = self._rom.solve(self._theta[None, :], want_state=True)
            self._state = (res["A_r"][0], res["B_r"][0])
        return self._state or (None, None)

    @property
    def _A_r(self):
        return self._reduced_state()[0]

    @property
    def _B_r(self):
        return self._reduced_state()[1]

    @property
    def dA_dsigmak(self):
        """Dense [9, n, n] stack as in the reference (:215-218); materialised on demand."""
        return np.stack([A.toarray() for A in self._A_sub])

    # ---- batched extensions (new, additive) ---------------------------------------------
    def subfin_avg_batch(self, K):
        """theta = S k on the device (finrom_subfin_avg), NumPy in -> NumPy out / torch in -> torch out."""
        return self._avg(K)

    def _theta_dev(self, K):
        """theta for the NEXT library call: NumPy fields are averaged on the device and stay there (a DeviceArray the reduced
        solve reads in place: one copy in, one copy of the results back -- the one-sample call pattern pays for every round trip)."""
        if isinstance(K, np.ndarray):
            return self._avg.on_device(np.ascontiguousarray(K, dtype=np.float64).reshape(-1, self.n))
        return self._avg(K)

    def forward_nine_param_reduced_batch(self, theta, want_state=False, want_w=True):
        """theta [S, 9] -> dict(w_r [S, r] (want_w), qoi_r [S, n_obs], info [S] (, A_r, B_r))."""
        return self._rom.solve(theta, want_state=want_state, want_w=want_w)

    def forward_reduced_batch(self, K, want_state=False, want_w=True):
        """K [S, n] nodal fields -> theta = S k on the device -> reduced solve."""
        return self._rom.solve(self._theta_dev(K), want_state=want_state, want_w=want_w)

    def forward_batch(self, K, want_w=True):
        """'Averaged FOM' (:237-258) for a batch of nodal fields."""
        if self._fom is None:
            ops = self.ops
            self._fom = FomEngine(self._plan, ops.robin_vals, sp.csr_matrix(ops.sub_vals.T), ops.F, self.B_obs, ops=ops)
        return self._fom.solve(self._theta_dev(K), want_w=want_w)

    def _ensure_gradient(self):
        """One-time: G_pi = (A_p Phi)^T (A_i Phi) for the region pairs that share nodes (finrom_rom_set_gradient)."""
        if self._grad_ready:
            return
        pairs, G = [], []
        for p in range(10):
            for i in range(9):
                with deterministic_blas():
                    M = self._psi_tables[p].T @ self.dA_dsigmak_phi[i]
                if np.any(M):
                    pairs.append((p, i)); G.append(M)
        self._rom.set_gradient_blocks(pairs, np.stack(G))
        self._grad_ready = True

    def grad_reduced_batch(self, K, data=None, theta=None):
        """Batched grad_reduced (:335-356): K [S, n] nodal fields (or theta [S, 9] directly) ->
        dict(J [S], g_theta [S, 9], w_r, qoi_r, info);  dJ_dk = g_theta @ dsigma_dk."""
        self._ensure_gradient()
        data = self.data if data is None else data
        th = self._theta_dev(K) if theta is None else theta
        res = self._rom.grad(th, data)
        res["g_theta"] = res.pop("g")
        return res

    # ---- reference call surface -------------------------------------------------------------
    def forward(self, k):
        res = self.forward_batch(as_nodal(k)[None, :])
        if res["info"][0]:
            raise np.linalg.LinAlgError("averaged operator not positive definite")
        w = Function(self.V, res["w"][0])
        w._qoi = res["qoi"][0].copy()
        return w

    def forward_reduced(self, k):
        t_i = time.time()
        k_s = self.subfin_avg_op(k)
        self.fwd_time += (time.time() - t_i)
        return self.forward_nine_param_reduced(k_s)

    def forward_nine_param_reduced(self, k_s):
        t_i = time.time()
        res = self._rom.solve(np.asarray(k_s, dtype=np.float64)[None, :])
        self.fwd_time += (time.time() - t_i)
        if res["info"][0]:
            raise np.linalg.LinAlgError("reduced operator not positive definite")
        self._theta, self._state = np.asarray(k_s, dtype=np.float64).copy(), None
        w_r = res["w_r"][0]
        self._last = (w_r.copy(), res["qoi_r"][0].copy())
        return w_r

    def qoi(self, w):
        q = getattr(w, "_qoi", None)
        if q is not None and q.shape[0] == self.n_obs:
            return q.copy()
        return np.dot(self.B_obs, as_nodal(w))

    def qoi_reduced(self, w_r):
        t_i = time.time()
        last = getattr(self, "_last", None)
        if last is not None and w_r.shape == last[0].shape and np.array_equal(w_r, last[0]):
            qoi_vals = last[1].copy()                           # computed by the solve kernel
        else:
            qoi_vals = np.dot(self.B_obs_phi, w_r)
        self.fwd_time += (time.time() - t_i)
        return qoi_vals

    def grad_reduced(self, k):
        t_i = time.time()
        res = self.grad_reduced_batch(as_nodal(k)[None, :])
        if res["info"][0]:
            raise np.linalg.LinAlgError("reduced operator not positive definite")
        dJ_dk = np.dot(res["g_theta"][0], self.dsigma_dk)       # [9] x [9, n]  (:349-351)
        self.rom_grad_time += (time.time() - t_i)
        return dJ_dk, float(res["J"][0])

    def grad_romml_batch(self, K, data=None):
        """Batched grad_romml (:358-396): ROM + learned-error value and gradient.  loss = 1/2 |data - (B_obs Phi w_r + e_NN(k))|^2;
        its gradient is the ROM adjoint gradient for the shifted data (data - e_NN) plus the network's vector-Jacobian
        product.  K [S, n] -> dict(grad [S, n], loss [S], qoi_r, e_NN, info).
        With a ResBnFcModel (deep_learning/dl_model.py) as the error model everything runs on the device in ONE library call
        (finrom_romml_grad: network forward, ROM adjoint, network backward, chain rule through the sub-fin averages); any other
        object with predict / vjp (a Keras wrapper, say) is evaluated where it lives, around the device ROM adjoint."""
        if self.dl_model is None:
            raise ValueError("grad_romml needs an error model (set_dl_model)")
        data = self.data if data is None else data
        if self._dev_model is not None:
            self._ensure_gradient()
            t_i = time.time()
            res = romml_grad(self._rom, self._dev_model, self._avg._S, K, data)
            self.romml_grad_time += (time.time() - t_i)
            return res
        K = np.ascontiguousarray(K, dtype=np.float64)
        data = np.asarray(data, dtype=np.float64)
        t_i = time.time()
        e_nn = np.asarray(self.dl_model.predict(K), dtype=np.float64)
        self.romml_grad_time_dl += (time.time() - t_i)
        t_i = time.time()
        res = self.grad_reduced_batch(K, data=np.broadcast_to(data, e_nn.shape) - e_nn)
        f_x = np.asarray(res["g_theta"]) @ self.dsigma_dk                              # :376-378
        self.romml_grad_time += (time.time() - t_i)
        resid = np.broadcast_to(data, e_nn.shape) - (np.asarray(res["qoi_r"]) + e_nn)
        t_i = time.time()
        nn_grad = -np.asarray(self.dl_model.vjp(K, resid), dtype=np.float64)           # d loss / d input (:226-228)
        self.romml_grad_time_dl += (time.time() - t_i)
        return {"grad": f_x + nn_grad, "loss": np.asarray(res["J"]), "qoi_r": res["qoi_r"], "e_NN": e_nn, "info": res["info"]}

    def grad_romml(self, k):
        res = self.grad_romml_batch(as_nodal(k)[None, :])
        if res["info"][0]:
            raise np.linalg.LinAlgError("reduced operator not positive definite")
        return res["grad"][0], float(res["loss"][0])

    def set_data(self, data):
        self.data = data

    def set_dl_model(self, model, device=True):
        """A ResBnFcModel whose hidden width the library supports goes to the device (device=False keeps it on the host, e.g.
        for A/B checks); other models are called on the host."""
        from ..deep_learning.dl_model import ResBnFcModel
        self.dl_model = model
        self._dev_model = None
        if device and isinstance(model, ResBnFcModel) and model.n_weights <= 64 and model.n_out <= 64 and model.n_in == self.n \
                and model.n_out == self.n_obs:
            self._dev_model = DeviceErrorModel(model)

    def subfin_avg_op(self, k):
        return np.asarray(self.subfin_avg_batch(as_nodal(k)[None, :]))[0]

    def observation_operator(self):
        return self.ops.S.copy()
