// The multi-wave projection kernels for bases whose last 16-column block is at most half full (r mod 16 in 1..8: r = 120, 200),
// HalfCover form (rom_proj_device.h): their own translation unit, because every (NB, NW) instantiation is NW specialised wave bodies.
#include "rom_proj_device.h"

namespace finrom {

int launch_rom_proj_half(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                         hipStream_t st, double* w_r, double* qoi_r) {
#define FR_CASE(N, W)                                                                                    \
  case N: { constexpr int wpb = W > 4 ? W : 4;                                                           \
            hipLaunchKernelGGL((rom_proj_kernel<N, W, true>), dim3((unsigned)S), dim3(64 * wpb), 0, st,  \
                               p, theta, S, Ar, Br, factor, info, w_r, qoi_r, p.kmeta); } break;
  switch (p.NB) {
    FR_CASE(7, 4) FR_CASE(8, 4) FR_CASE(9, 4) FR_CASE(10, 8) FR_CASE(11, 8) FR_CASE(12, 8) FR_CASE(13, 8)
    default: set_error("rom_proj_half: basis size"); return FINROM_ERR_UNSUPPORTED;
  }
#undef FR_CASE
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace finrom
