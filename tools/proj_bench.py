"""Stand-alone timing of the ROM half (projection + reduced solve) and of the FOM half at the headline sizes.
usage: python tools/proj_bench.py [r] [S] [reps]   (env switches of the library apply)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianinferencedl_amd import _ffi
from bayesianinferencedl_amd.fom.thermal_fin import get_space
from bayesianinferencedl_amd.fom.forward_solve import Fin
from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
from bayesianinferencedl_amd.rom.basis import pod_basis
r = int(sys.argv[1]) if len(sys.argv) > 1 else 80
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
what = sys.argv[4] if len(sys.argv) > 4 else "rom"
V = get_space(None, m=12); fin = Fin(V)
phi = pod_basis(fin, r, n_snapshots=max(2 * r, 200), low=0.1, high=10.0, params="nine", seed=1)
rom = AffineROMFin(V, None, phi, projection=os.environ.get("FINROM_PROJECTION", "direct"))
rng = np.random.default_rng(3)
th = torch.from_numpy(rng.uniform(0.1, 10.0, (S, 9))).cuda()
x5 = torch.from_numpy(rng.uniform(0.1, 10.0, (S, 5))).cuda()
L = _ffi.lib()
def run():
    if what == "rom": return rom.forward_nine_param_reduced_batch(th, want_w=os.environ.get("WANT_W", "0") == "1")
    return fin.forward_batch(x5, want_w=False, params="five")
run(); torch.cuda.synchronize()
L.finrom_profile_reset(); L.finrom_profile_enable(1)
t0 = time.perf_counter()
for _ in range(reps): out = run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
L.finrom_profile_enable(0)
print(what, "r", r, "S", S, "wall ms", round(1e3 * dt, 3), {k: round(v[1] / v[0], 3) for k, v in _ffi.profile_read().items() if v[0]},
      "bad", int((out["info"] != 0).sum()))
