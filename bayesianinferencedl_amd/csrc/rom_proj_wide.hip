// The eight-waves-per-sample projection kernels (144 < r <= 208) in their own translation unit: with the fused factorisation
// every (NB, NW) instantiation is eight specialised wave bodies, and rom_kernels.hip alone took 2.5 minutes to compile.
#include "rom_proj_device.h"

namespace finrom {

int launch_rom_proj_wide(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                         hipStream_t st, double* w_r, double* qoi_r) {
#define FR_CASE(N)                                                                                               \
  case N: hipLaunchKernelGGL((rom_proj_kernel<N, 8>), dim3((unsigned)S), dim3(512), 0, st, p, theta, S, Ar, Br, \
                             factor, info, w_r, qoi_r, p.kmeta); break;
  switch (p.NB) {
    FR_CASE(10) FR_CASE(11) FR_CASE(12) FR_CASE(13)
    default: set_error("rom_proj_wide: basis size"); return FINROM_ERR_UNSUPPORTED;
  }
#undef FR_CASE
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace finrom
