import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from bayesianinferencedl_amd.fom.forward_solve import Fin
from bayesianinferencedl_amd.fom.thermal_fin import get_space
from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
from bayesianinferencedl_amd.rom.basis import pod_basis
V = get_space(40); fin = Fin(V); phi = pod_basis(fin, 80)
for proj in ("direct", "offline_online"):
    rom = AffineROMFin(V, None, phi, projection=proj)
    rng = np.random.default_rng(0)
    rom.set_data(rng.uniform(0.2, 1.0, 9))
    TH = torch.from_numpy(rng.uniform(0.1, 3.0, (100000, 9))).cuda()
    data = torch.from_numpy(np.asarray(rom.data)).cuda()
    rom._ensure_gradient()
    for _ in range(2): res = rom._rom.grad(TH, data)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): res = rom._rom.grad(TH, data)
    torch.cuda.synchronize(); print(proj, "grad batch 100k: %.2f ms" % (1e3 * (time.perf_counter() - t0) / 3), float(res["J"].sum()))
