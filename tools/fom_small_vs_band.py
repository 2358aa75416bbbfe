import sys, time
import numpy as np
sys.path.insert(0, ".")
from bayesianinferencedl_amd.fom.forward_solve import Fin
from bayesianinferencedl_amd.fom.thermal_fin import get_space
for m in (12, 20):
    V = get_space(None, m=m); fin = Fin(V); rng = np.random.default_rng(0)
    K = np.exp(0.3 * rng.standard_normal((64, V.dim())))
    data = rng.uniform(0.1, 0.6, 9)
    for small in (True, False):
        eng = fin._engine("field")
        if not small: eng.set_small_max(0)
        for S in (1, 8, 64):
            for want_w in (False, True):
                for _ in range(3): fin.forward_batch(K[:S], want_w=want_w)
                t0 = time.perf_counter()
                for _ in range(30): fin.forward_batch(K[:S], want_w=want_w)
                dt = (time.perf_counter() - t0) / 30
                print(f"m={m} small={small} S={S} want_w={want_w}: {dt*1e3:.3f} ms  path={eng.last_path()}", flush=True)
            for _ in range(3): fin.gradient_batch(K[:S], data)
            t0 = time.perf_counter()
            for _ in range(30): fin.gradient_batch(K[:S], data)
            print(f"m={m} small={small} S={S} gradient: {(time.perf_counter()-t0)/30*1e3:.3f} ms path={eng.last_path()}", flush=True)
