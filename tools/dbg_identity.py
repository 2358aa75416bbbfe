import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bayesianinferencedl_amd.fom.thermal_fin import get_space
from bayesianinferencedl_amd.fom.forward_solve import Fin
from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
from bayesianinferencedl_amd.rom.basis import pod_basis
from bayesianinferencedl_amd.pairs import FinPairSolver
import hashlib
V = get_space(None, m=12); solver = Fin(V)
phi = pod_basis(solver, 80, n_snapshots=400, low=0.1, high=10.0, params="five", seed=1)
print("phi sha", hashlib.sha256(phi.tobytes()).hexdigest()[:16], "threads", os.environ.get("OMP_NUM_THREADS"))
rom = AffineROMFin(V, None, phi)
print("B_obs_phi sha", hashlib.sha256(rom.B_obs_phi.tobytes()).hexdigest()[:16], "psi", hashlib.sha256(rom.dA_dsigmak_phi.tobytes()).hexdigest()[:16])
pairs = FinPairSolver(V, phi, False, "five", solver, rom)
X = torch.from_numpy(bench.global_uniform(3, 0, 6000, 5)).cuda()
full = pairs.solve_pairs(X)
a = pairs.solve_pairs(X[:3000].contiguous()); b = pairs.solve_pairs(X[3000:].contiguous())
for k in ("qoi", "qoi_r"):
    cat = torch.cat([a[k], b[k]])
    print(k, "split == full:", bool(torch.equal(cat, full[k])), "maxdiff", float((cat - full[k]).abs().max()))
    print(k, "sha", hashlib.sha256(full[k].cpu().numpy().tobytes()).hexdigest()[:16])
