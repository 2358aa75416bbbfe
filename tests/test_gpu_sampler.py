"""The Gaussian-field sampler's two GEMM kernels (csrc/util_kernels.hip): k = exp(0.5 xi U), U upper triangular
(deep_learning/generate_fin_dataset.py:87-88 with U = make_cov_chol(...), bayesian_inference/gaussian_field.py:9-31).
Batches of >= 4096 samples take the blocked throughput kernel (256 x 128 tiles, super-tiles per XCD), smaller ones the 64 x 64
kernel; both add the same products in the same order, so they must agree BIT FOR BIT, and both must match NumPy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _factor(n, seed):
    rng = np.random.default_rng(seed)
    U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n)
    U[np.arange(n), np.arange(n)] = rng.uniform(0.2, 1.0, n)
    return U


@pytest.mark.parametrize("n,S", [(130, 4096), (148, 4100), (300, 4357), (517, 5000), (1597, 4200)])
def test_throughput_sampler_kernel_matches_numpy_and_the_small_kernel_bit_for_bit(n, S, monkeypatch):
    """n = 130 / 148 / 300 / 517 / 1597: the last column tile holds 2 / 20 / 44 / 5 / 61 live columns (one, two, three, one, four
    live 16-column MFMA tiles on the first wave column, none on the second); S not a multiple of 256 nor of 1024: partial sample
    tiles and a partial sample group; several column groups at n = 1597 (13 column tiles: groups of 8 and 5)."""
    import torch
    from bayesianinferencedl_amd.engine import FieldSampler
    dev = torch.device("cuda", torch.cuda.current_device())
    U = _factor(n, n)
    xi = np.random.default_rng(S).standard_normal((S, n))
    smp = FieldSampler(U)
    xt = torch.from_numpy(xi).to(dev)
    monkeypatch.setenv("FINROM_SAMPLER_GEMM_MIN", "1")            # the throughput kernel
    big = smp(xt).cpu().numpy()
    monkeypatch.setenv("FINROM_SAMPLER_NO_PAD", "1")              # every tile stops at its own K end: the padding adds exact zeros
    nopad = smp(xt).cpu().numpy()
    monkeypatch.delenv("FINROM_SAMPLER_NO_PAD")
    monkeypatch.setenv("FINROM_SAMPLER_WM", "2")                  # the 128 x 128 instantiation (two workgroups per CU; A/B only)
    half = smp(xt).cpu().numpy()
    monkeypatch.delenv("FINROM_SAMPLER_WM")
    monkeypatch.setenv("FINROM_SAMPLER_GEMM_MIN", str(1 << 40))   # the 64 x 64 kernel
    small = smp(xt).cpu().numpy()
    ref = np.exp(0.5 * (xi @ U))
    assert np.max(np.abs(big - ref) / ref) < 1e-13
    assert np.array_equal(big, small) and np.array_equal(big, nopad) and np.array_equal(big, half)


def test_sampler_dispatches_by_batch_size_and_seeded_draws_do_not_depend_on_it():
    """Default thresholds: the same seeded stream drawn as one batch of 4608 (throughput kernel) and as pieces of 1536 (small
    kernel) gives identical fields (Philox keyed by the global sample index + bit-identical kernels)."""
    import torch
    from bayesianinferencedl_amd.engine import FieldSampler
    dev = torch.device("cuda", torch.cuda.current_device())
    like = torch.empty(0, dtype=torch.float64, device=dev)
    smp = FieldSampler(_factor(245, 3))
    whole = smp.draw(9, 100, 4608, like=like).cpu().numpy()
    parts = np.concatenate([smp.draw(9, 100 + i, 1536, like=like).cpu().numpy() for i in range(0, 4608, 1536)])
    assert np.array_equal(whole, parts)
