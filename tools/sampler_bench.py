"""Stand-alone timing of the Gaussian-field sampler's GEMM kernel (finrom_sampler_draw) at BASELINE configs[3]'s size:
n = 4101 (m = 20), S samples per launch; library HIP-event timers.  usage: python tools/sampler_bench.py [S] [n] [variant] [reps]
(variant: one of small_64x64 / gemm_256x128 / gemm_256x128_nopad -- for rocprofv3 --pmc passes, tools/pmc_sampler.sh)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianinferencedl_amd import _ffi                      # noqa: E402
from bayesianinferencedl_amd.engine import FieldSampler       # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4101
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n)
smp = FieldSampler(U)
xi = torch.randn(S, n, dtype=torch.float64, device=dev)
L = _ffi.lib()
only = sys.argv[3] if len(sys.argv) > 3 else None
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
res = {}
for tag, env in (("small_64x64", {"FINROM_SAMPLER_GEMM_MIN": str(1 << 40)}),
                 ("gemm_256x128", {"FINROM_SAMPLER_GEMM_MIN": "1"}),
                 ("gemm_256x128_nopad", {"FINROM_SAMPLER_GEMM_MIN": "1", "FINROM_SAMPLER_NO_PAD": "1"}),
                 ("pad_top1", {"FINROM_SAMPLER_GEMM_MIN": "1", "FINROM_SAMPLER_PAD_GROUPS": "1"}),
                 ("pad_top2", {"FINROM_SAMPLER_GEMM_MIN": "1", "FINROM_SAMPLER_PAD_GROUPS": "2"}),
                 ("pad_top3", {"FINROM_SAMPLER_GEMM_MIN": "1", "FINROM_SAMPLER_PAD_GROUPS": "3"}),
                 ("x_nobarrier", {"FINROM_SAMPLER_GEMM_MIN": "1", "FINROM_SAMPLER_XFLAGS": "1"}),
                 ("x_nostaging", {"FINROM_SAMPLER_GEMM_MIN": "1", "FINROM_SAMPLER_XFLAGS": "2"}),
                 ("x_neither", {"FINROM_SAMPLER_GEMM_MIN": "1", "FINROM_SAMPLER_XFLAGS": "3"}),
                 ("gemm_128x128", {"FINROM_SAMPLER_GEMM_MIN": "1", "FINROM_SAMPLER_WM": "2"}),
                 ("gemm_128x128_nopad", {"FINROM_SAMPLER_GEMM_MIN": "1", "FINROM_SAMPLER_NO_PAD": "1", "FINROM_SAMPLER_WM": "2"})):
    if only and tag != only:
        continue
    for k in ("FINROM_SAMPLER_GEMM_MIN", "FINROM_SAMPLER_NO_PAD", "FINROM_SAMPLER_WM", "FINROM_SAMPLER_XFLAGS", "FINROM_SAMPLER_PAD_GROUPS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    out = smp(xi)
    torch.cuda.synchronize()
    L.finrom_profile_reset(); L.finrom_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = smp(xi)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    L.finrom_profile_enable(0)
    cnt, ms = _ffi.profile_read()["sampler_gemm_exp"]
    res[tag] = out
    tf = S * float(n) * n / (ms / cnt * 1e-3) / 1e12
    print(f"{tag}: {ms / cnt:.3f} ms per launch ({wall * 1e3:.3f} wall), {tf:.1f} TFLOP/s useful = {tf / 78.6:.3f} of peak", flush=True)
if not only:
    print("bit-identical:", {k_: bool(torch.equal(res["small_64x64"], v)) for k_, v in res.items()})
