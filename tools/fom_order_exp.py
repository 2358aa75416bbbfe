"""Experiment driver: the FOM interpreter alone on the headline batch under different op-stream orderings, all in one
process on one box (box-to-box variance is ~5 %).  Usage: python tools/fom_order_exp.py "FWD=postorder,BWD=rpo" "FWD=height" ...
Each spec is a comma-separated list of FINROM_<KEY>_ORDER=value settings read by symbolic.build_op_streams, or CACHE=<slots>."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianinferencedl_amd.fom.thermal_fin import get_space      # noqa: E402
from bayesianinferencedl_amd.fom.forward_solve import Fin          # noqa: E402


def main():
    specs = sys.argv[1:] or ["FWD=height"]
    S = int(os.environ.get("EXP_SAMPLES", "100000"))
    m = int(os.environ.get("EXP_M", "12"))
    X = torch.from_numpy(np.random.default_rng(3).uniform(0.1, 10.0, (S, 5))).cuda()
    engines = []
    exp_env = set()
    for spec in specs:
        for k in list(os.environ):
            if k.startswith("FINROM_") and (k.endswith("_ORDER") or k in exp_env):
                del os.environ[k]
        import bayesianinferencedl_amd.engine as E
        E.ROW_CACHE_SLOTS = 40
        E.FWD_CHUNK = 8
        for kv in spec.split(","):
            if kv:
                k, v = kv.split("=")
                if k.startswith("ENV_"):             # ENV_NAME=v -> FINROM_NAME=v for this engine's creation
                    os.environ["FINROM_" + k[4:]] = v; exp_env.add("FINROM_" + k[4:])
                elif k == "CHUNK":
                    E.FWD_CHUNK = int(v)
                elif k == "CACHE":
                    E.ROW_CACHE_SLOTS = int(v)          # LDS slots incl. the x slots of the fused assembly
                else:
                    os.environ[f"FINROM_{k}_ORDER"] = v
        V = get_space(None, m=m)
        V._chol_plan = None
        fin = Fin(V)
        fin.forward_batch(X, want_w=False, params="five")
        engines.append((spec, fin))
    torch.cuda.synchronize()
    times = {s: [] for s, _ in engines}
    ref = None
    for rep in range(int(os.environ.get("EXP_REPS", "5"))):
        for spec, fin in engines:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = fin.forward_batch(X, want_w=False, params="five")
            torch.cuda.synchronize(); times[spec].append(1e3 * (time.perf_counter() - t0))
            if os.environ.get("EXP_NOCHECK"):
                pass
            elif ref is None:
                ref = res["qoi"].clone()
            else:
                assert float((res["qoi"] - ref).abs().max() / ref.abs().max()) < 1e-10
    for spec, _ in engines:
        t = sorted(times[spec])
        print(f"{spec:40s} min {t[0]:.3f} ms  median {t[len(t) // 2]:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
