"""Phase clocks of the one-sample solve kernel (rom_small_solve_kernel): needs a library built with
    FINROM_EXTRA_FLAGS=-DFINROM_SOLVE_CLOCKS python -m bayesianinferencedl_amd._build      (touch csrc/rom_onesample.hip first)
in which lane 0 writes the phases' durations (10 ns ticks of the real-time counter) over the sample's qoi_r -- results are garbage in
that build.  Phases: the NC partial triangles summed into LDS | B_r | forward (factorisation + Z) | middle | backward."""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bayesianinferencedl_amd.fom.thermal_fin import get_space
from bayesianinferencedl_amd.fom.forward_solve import Fin
from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin
from bayesianinferencedl_amd.rom.basis import pod_basis
V=get_space(None,m=12); solver=Fin(V)
phi=pod_basis(solver,81,n_snapshots=120,low=0.1,high=10.0,params="nine",seed=1)
model=bench.hmc_error_model(V.dim())
rom=AffineROMFin(V,model,phi); rom.set_data(np.zeros(9))
K=torch.from_numpy(np.exp(0.1*np.random.default_rng(0).standard_normal((4,V.dim())))).cuda()
for _ in range(5):
    res=rom.grad_romml_batch(K)
torch.cuda.synchronize()
print("ticks of 10 ns: partial sums, B_r, forward, middle, backward, (diagonal chains inside forward):\n", res["qoi_r"].cpu().numpy()[:, :8], "\n(columns 6, 7: the spare wave's wait for the staged weights, its walk through the layers)")
