"""Inference-time stand-in for the reference's error model (deep_learning/dl_model.py:149-176 `res_bn_fc_model`): a
residual, batch-normalised, fully connected network  R^n -> R^n_obs  evaluated in fp32 like the Keras original, with the
two operations the inverse-problem callers need -- `predict` (rom/averaged_affine_ROM.py:360) and the vector-Jacobian
product behind `tf.gradients(loss, model.input)` (:226-228).  Host NumPy: at one sample per call (the HMC / MAP call
pattern) the network is two skinny GEMVs; training is out of scope.

Architecture as the reference builds it (the first BN-activation-Dense triple of `residual_unit` is overwritten before it
is used, :150-158, so a unit is  x + Dense(act(BN(x)))):
    y0 = Dense(n_in -> n_w)(x);  y_{i+1} = y_i + Dense(n_w -> n_w)(ELU(BN(y_i))), i < n_layers;
    out = Dense(n_w -> n_out)(ELU(BN(y_L)))
Weights live in an .npz (`save` / `load`); Keras checkpoints cannot be read here (no h5py / tensorflow)."""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-3                     # Keras BatchNormalization default


def _elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def _elu_grad(x):
    return np.where(x > 0, 1.0, np.exp(np.minimum(x, 0))).astype(x.dtype)


class ResBnFcModel:
    def __init__(self, n_in, n_out=9, n_layers=5, n_weights=50, seed=0):
        rng = np.random.default_rng(seed)
        self.n_in, self.n_out, self.n_layers, self.n_weights = n_in, n_out, n_layers, n_weights

        def glorot(a, b):
            lim = np.sqrt(6.0 / (a + b))
            return rng.uniform(-lim, lim, (a, b)).astype(np.float32)
        self.W0, self.b0 = glorot(n_in, n_weights), np.zeros(n_weights, np.float32)
        self.units = [{"gamma": np.ones(n_weights, np.float32), "beta": np.zeros(n_weights, np.float32),
                       "mean": np.zeros(n_weights, np.float32), "var": np.ones(n_weights, np.float32),
                       "W": glorot(n_weights, n_weights), "b": np.zeros(n_weights, np.float32)} for _ in range(n_layers)]
        self.head = {"gamma": np.ones(n_weights, np.float32), "beta": np.zeros(n_weights, np.float32),
                     "mean": np.zeros(n_weights, np.float32), "var": np.ones(n_weights, np.float32),
                     "W": glorot(n_weights, n_out), "b": np.zeros(n_out, np.float32)}

    # -- the two operations used by AffineROMFin.grad_romml --------------------------------------------------------
    def _as_batch(self, x):
        return np.asarray(x, dtype=np.float32).reshape(-1, self.n_in)

    @staticmethod
    def _bn(u, y):
        s = u["gamma"] / np.sqrt(u["var"] + BN_EPS)
        return y * s + (u["beta"] - u["mean"] * s), s

    def _forward(self, X):
        y = X @ self.W0 + self.b0
        tape = []
        for u in self.units + [self.head]:
            z, s = self._bn(u, y)
            tape.append((z, s))
            d = _elu(z) @ u["W"] + u["b"]
            y = d if u is self.head else y + d
        return y, tape

    def predict(self, x):
        """x [S, n_in] (or Keras-style nested lists) -> [S, n_out] float32."""
        return self._forward(self._as_batch(x))[0]

    def vjp(self, x, upstream):
        """upstream [S, n_out] = dLoss/d(output) -> dLoss/d(input) [S, n_in] (what tf.gradients(loss, input) returns)."""
        X = self._as_batch(x)
        _, tape = self._forward(X)
        g = np.asarray(upstream, dtype=np.float32).reshape(-1, self.n_out)
        layers = self.units + [self.head]
        z, s = tape[-1]
        g = (g @ self.head["W"].T) * _elu_grad(z) * s                       # through the head: no skip connection
        for u, (z, s) in zip(reversed(layers[:-1]), reversed(tape[:-1])):
            g = g + (g @ u["W"].T) * _elu_grad(z) * s                       # skip + branch
        return g @ self.W0.T

    # -- persistence -------------------------------------------------------------------------------------------------
    def save(self, path):
        arrs = {"meta": np.array([self.n_in, self.n_out, self.n_layers, self.n_weights]), "W0": self.W0, "b0": self.b0}
        for i, u in enumerate(self.units + [self.head]):
            for k, v in u.items():
                arrs[f"l{i}_{k}"] = v
        np.savez(path, **arrs)

    @classmethod
    def load(cls, path):
        d = np.load(path)
        n_in, n_out, n_layers, n_weights = (int(v) for v in d["meta"])
        m = cls(n_in, n_out, n_layers, n_weights)
        m.W0, m.b0 = d["W0"], d["b0"]
        for i, u in enumerate(m.units + [m.head]):
            for k in u:
                u[k] = d[f"l{i}_{k}"]
        return m


def load_dataset_avg_rom(load_prev=True, tr_size=6000, v_size=500, genrand=False, data_dir='../data', **gen_kwargs):
    """Reader contract of the reference's training scripts (deep_learning/dl_model.py:19-36): conductivity fields and
    QoI errors for training and validation, loaded from `<data_dir>/z_aff_avg_{tr,eval}.npy` /
    `errors_aff_avg_{tr,eval}.npy` when present (and load_prev), otherwise generated on the device by
    gen_affine_avg_rom_dataset (which writes the `*_avg_obs_3` files of its own, :102-110).
    -> (z_train, errors_train, z_val, errors_val)."""
    import os
    from .generate_fin_dataset import gen_affine_avg_rom_dataset
    out = []
    for tag, size in (("tr", tr_size), ("eval", v_size)):
        zf, ef = os.path.join(data_dir, f"z_aff_avg_{tag}.npy"), os.path.join(data_dir, f"errors_aff_avg_{tag}.npy")
        if load_prev and os.path.isfile(zf) and os.path.isfile(ef):
            out += [np.load(zf), np.load(ef)]
        else:
            out += list(gen_affine_avg_rom_dataset(size, genrand=genrand, out_dir=data_dir, **gen_kwargs))
    return tuple(out)


def res_bn_fc_model(n_layers, n_weights, input_shape=1446, output_shape=9, seed=0):
    """Constructor with the reference's argument meaning (activation / optimiser / learning rate dropped: inference only)."""
    return ResBnFcModel(input_shape, output_shape, n_layers, n_weights, seed)
