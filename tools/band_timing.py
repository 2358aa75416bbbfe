"""Phase clocks of the LDS-window band sweep (FINROM_BAND_TIMING=1): fins forward, post forward, post backward, fins backward."""
import os, sys
os.environ.setdefault("FINROM_BAND_TIMING", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianinferencedl_amd.fom.thermal_fin import get_space
from bayesianinferencedl_amd.fom.forward_solve import Fin
m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
V = get_space(None, m=m); fin = Fin(V)
X = torch.from_numpy(np.random.default_rng(0).uniform(0.5, 2.0, (S, 9))).cuda()
for _ in range(2):
    res = fin.forward_batch(X, want_w=False, params="nine")
torch.cuda.synchronize()
q = res["qoi"][0, :8].cpu().numpy() if hasattr(res["qoi"], "cpu") else np.asarray(res["qoi"])[0, :8]
print("m", m, "S", S, "phase ms (fins fwd, post fwd, post bwd, fins bwd):", [round(float(x) / 1e5, 3) for x in q[:4]])
