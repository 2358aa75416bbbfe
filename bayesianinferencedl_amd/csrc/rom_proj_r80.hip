// The r = 65..80 projection kernel (the headline size) in its own translation unit: built at -O2, see _build.py.
#include "rom_proj_device.h"

namespace finrom {

// r = 65..80, the headline size: the same body under an explicit register cap -- 2 x 192 registers of this kernel and
// 2 x 56 of the FOM interpreter fill a SIMD's 512 exactly; without the cap hipcc's scheduler spends the
// 256 registers its occupancy target allows and the two kernels no longer fit together.
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(192))) void rom_proj_kernel_r80(
    RomDev p, const double* __restrict__ theta, int64_t S, double* __restrict__ Ar, double* __restrict__ Br, int factor,
    int* __restrict__ info, double* __restrict__ w_r, double* __restrict__ qoi_r) {
  rom_proj_entry<5, 1>(p, theta, S, Ar, Br, factor, info, w_r, qoi_r);
}


int launch_rom_proj_r80(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                        hipStream_t st, double* w_r, double* qoi_r) {
  hipLaunchKernelGGL(rom_proj_kernel_r80, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, st, p, theta, S, Ar, Br, factor, info, w_r, qoi_r);
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace finrom
