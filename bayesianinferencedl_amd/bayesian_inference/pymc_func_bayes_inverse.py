"""Misfit value-and-gradient wrappers of the reference's HMC / NUTS driver (bayesian_inference/pymc_func_bayes_inverse.py:29-167):
`SqError` evaluates 1/2 |obs(k) - data|^2 and its gradient with the full-order model, the ROM, or the ROM plus the learned
error model -- every evaluation a one-sample device call (finrom_fom_gradient / finrom_rom_grad through the small-batch
schedules); `SqErrorOpFOM / ROM / ROMML` have the shape of the reference's Theano operators (`perform(node, inputs, outputs)`
writes the value and the gradient) without importing theano, which this environment does not have: with theano / pytensor
installed, `make_op(kind, ...)` wraps the same callable in a real Op."""
from __future__ import annotations

import numpy as np

from ..fem import Function
from ..fom.forward_solve import Fin
from ..rom.averaged_affine_ROM import AffineROMFin


class SqError:
    def __init__(self, V, chol=None, randobs=False, *, phi, k_true=None, err_model=None, rng=None):
        """V: function space; chol: upper Cholesky factor of the prior covariance (used to draw the synthetic truth when
        k_true is not given, :47-52); phi: reduced basis; err_model: object with predict / vjp (deep_learning/dl_model.py)."""
        self._V = V
        self._solver = Fin(V, randobs)
        self._pred_k = Function(V)
        if k_true is None:
            rng = np.random.default_rng() if rng is None else rng
            k_true = np.exp(0.5 * np.asarray(chol).T @ rng.standard_normal(V.dim()))
        self.k_true = Function(V, np.asarray(k_true, dtype=np.float64))
        w = self._solver.forward(self.k_true)[0]
        self.obs_data = self._solver.qoi_operator(w)                      # synthetic observations (:54-55)
        self._err_model = err_model
        self.phi = phi
        self._solver_r = AffineROMFin(V, err_model, phi, randobs)
        self._solver_r.set_data(self.obs_data)

    def err_grad_FOM(self, pred_k):                                       # :70-80
        self._pred_k.vector().set_local(pred_k)
        res = self._solver.gradient_batch(self._pred_k.vector()[:][None, :], self.obs_data)
        if res["info"][0]:
            raise np.linalg.LinAlgError("FOM operator not positive definite for this conductivity")
        return float(res["J"][0]), np.asarray(res["grad"][0])

    def err_grad_ROM(self, pred_k):                                       # :82-92
        self._pred_k.vector().set_local(pred_k)
        grad_r, err_r = self._solver_r.grad_reduced(self._pred_k)
        return err_r, grad_r

    def err_grad_ROMML(self, pred_k):                                     # :94-107
        self._pred_k.vector().set_local(pred_k)
        grad_t, err_t = self._solver_r.grad_romml(self._pred_k)
        return err_t, grad_t


class _SqErrorOp:
    """Theano-Op-shaped callable: inputs [k: dvector] -> outputs [value: dscalar, gradient: dvector] (:109-124)."""
    kind = None

    def __init__(self, V, chol=None, randobs=False, **kw):
        self._error_op = SqError(V, chol, randobs, **kw)

    def __call__(self, pred_k):
        return getattr(self._error_op, "err_grad_" + self.kind)(np.asarray(pred_k, dtype=np.float64))

    def perform(self, node, inputs, outputs):
        value, grad = self(inputs[0])
        outputs[0][0] = np.asarray(value)
        outputs[1][0] = grad

    def grad(self, inputs, output_gradients):
        return [output_gradients[0] * self(inputs[0])[1]]


class SqErrorOpFOM(_SqErrorOp):
    kind = "FOM"


class SqErrorOpROM(_SqErrorOp):
    kind = "ROM"


class SqErrorOpROMML(_SqErrorOp):
    kind = "ROMML"


def make_op(kind, V, chol=None, randobs=False, **kw):
    """A real theano / pytensor Op around the same evaluation, when one of the two packages is importable."""
    try:
        import pytensor as backend, pytensor.tensor as tt
    except ImportError:
        import theano as backend, theano.tensor as tt          # raises ImportError if neither is installed
    base = {"FOM": SqErrorOpFOM, "ROM": SqErrorOpROM, "ROMML": SqErrorOpROMML}[kind]

    class _Op(backend.Op if hasattr(backend, "Op") else backend.graph.op.Op):
        itypes, otypes, __props__ = [tt.dvector], [tt.dscalar, tt.dvector], ()

        def __init__(self):
            self._impl = base(V, chol, randobs, **kw)

        def perform(self, node, inputs, outputs):
            self._impl.perform(node, inputs, outputs)

        def grad(self, inputs, output_gradients):
            return [output_gradients[0] * self(*inputs)[1]]
    return _Op()
