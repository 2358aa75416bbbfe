// The single-wave projection kernels (r <= 80: one wave owns a sample, factorisation + substitutions in registers) in their
// own translation unit, built at -O2 (see _build.py): hipcc's -O3 inflates their register use (r = 80: 196 -> 256 VGPRs +
// scratch), and they must stay small enough to share a SIMD with a wave of the FOM band sweep.
#include "rom_proj_device.h"

namespace finrom {

// amdgpu_num_vgpr(100): on gfx90a+ the attribute counts the unified file in halves, i.e. this caps the kernel at 200
// architectural VGPRs (hipcc otherwise stops at whatever fits two waves per SIMD -- 202 at r = 80, no spills either way):
// 200 + the band sweep's 312 (256 + 50 accumulation registers, granule 8) fill a SIMD's 512 exactly, so that a projection wave
// and a sweep wave can share a SIMD; at 208 they could not, and the sweep would have to wait for a SIMD to drain completely.
template <int NB>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(100))) void rom_proj_single_kernel(RomDev p, const double* __restrict__ theta, int64_t S,
                                                                 double* __restrict__ Ar, double* __restrict__ Br, int factor,
                                                                 int* __restrict__ info, double* __restrict__ w_r,
                                                                 double* __restrict__ qoi_r, const int* __restrict__ kpat) {
  // (kpat = p.kpat as a __restrict__ kernel parameter of its own: only then are its reads scalar loads)
  // (with p.ext: the grouped main loop -- kpat is then p.kmg)
  rom_proj_entry<NB, 1, true>(p, theta, S, Ar, Br, factor, info, w_r, qoi_r, kpat);
}

// the samples' scalars for the grouped main loop (RomDev::ext_def): ext[s][l] = (theta'[a_l] / theta'[b_l]) ^ (1 + sq_l), theta'[0] = 1
__global__ __launch_bounds__(256) void rom_ext_kernel(const double* __restrict__ theta, int P, int64_t S, const int* __restrict__ def,
                                                      int n_ext, double* __restrict__ ext) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= S * n_ext) return;
  const int64_t s = i / n_ext;
  const int l = (int)(i - s * n_ext);
  const int a = def[3 * l], b = def[3 * l + 1];
  const double num = a ? theta[s * P + a - 1] : 1.0, den = b ? theta[s * P + b - 1] : 1.0;
  double v = num / den;
  if (def[3 * l + 2]) v *= v;
  ext[i] = v;
}

// small batches: the sample's k-steps split over four waves (rom_proj_entry_splitk)
template <int NB>
__global__ __launch_bounds__(256, 1) void rom_proj_splitk_kernel(RomDev p, const double* __restrict__ theta, int64_t S,
                                                                 double* __restrict__ Ar, double* __restrict__ Br, int factor,
                                                                 int* __restrict__ info, double* __restrict__ w_r,
                                                                 double* __restrict__ qoi_r, const int* __restrict__ kpat, RomGradArgs ga) {
  rom_proj_entry_splitk<NB, 4>(p, theta, S, Ar, Br, factor, info, w_r, qoi_r, kpat, ga);
}

template <int NB>
static int launch_splitk(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                         hipStream_t st, double* w_r, double* qoi_r, const RomGradArgs& ga) {
  constexpr int lds = 3 * (NB * (NB + 1) / 2) * 256 * (int)sizeof(double);
  static PerDeviceOnce once;
  if (int rc = once.run([&]() -> int { FR_HIP(hipFuncSetAttribute((const void*)rom_proj_splitk_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); return 0; })) return rc;
  hipLaunchKernelGGL(rom_proj_splitk_kernel<NB>, dim3((unsigned)S), dim3(256), lds, st, p, theta, S, Ar, Br, factor, info, w_r, qoi_r, p.kmeta, ga);
  FR_HIP(hipGetLastError());
  return 0;
}

bool rom_splitk_applies(const RomDev& p, int64_t S) {
  static const bool no_splitk = getenv("FINROM_NO_SPLITK") != nullptr;
  return S <= ROM_SPLITK_MAX_S && p.NB >= 4 && p.NB <= 6 && p.nku >= 64 && !no_splitk;
}

// factor 0 / 1 / 2 as in launch_rom_proj; 4 (with ga): the whole adjoint gradient (finrom_rom_grad) in this one kernel
int launch_rom_proj_splitk(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                           hipStream_t st, double* w_r, double* qoi_r, const RomGradArgs& ga) {
  if (S == 0) return 0;
  switch (p.NB) {
    case 4: return launch_splitk<4>(p, theta, S, Ar, Br, factor, info, st, w_r, qoi_r, ga);
    case 5: return launch_splitk<5>(p, theta, S, Ar, Br, factor, info, st, w_r, qoi_r, ga);
    case 6: return launch_splitk<6>(p, theta, S, Ar, Br, factor, info, st, w_r, qoi_r, ga);
    default: set_error("rom_proj_splitk: basis size"); return FINROM_ERR_UNSUPPORTED;
  }
}

int launch_rom_proj_single(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                           hipStream_t st, double* w_r, double* qoi_r, int* /*cu_ticket: unused (stagger experiment)*/) {
  const dim3 grid((unsigned)((S + 3) / 4)), block(256);
  static const size_t pad_lds = getenv("FINROM_PROJ_PAD_LDS") != nullptr ? (size_t)atoi(getenv("FINROM_PROJ_PAD_LDS")) : 0;      // occupancy experiments
  if (p.ext != nullptr)
    hipLaunchKernelGGL(rom_ext_kernel, dim3((unsigned)((S * p.n_ext + 255) / 256)), dim3(256), 0, st, theta, p.P, S, p.ext_def, p.n_ext, p.ext);
  const int* kp = p.ext != nullptr ? p.kmg : p.kmeta;
  switch (p.NB) {
#define FR_ONE(N) case N: if (pad_lds > 65536) (void)hipFuncSetAttribute((const void*)rom_proj_single_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad_lds); \
                       hipLaunchKernelGGL(rom_proj_single_kernel<N>, grid, block, pad_lds, st, p, theta, S, Ar, Br, factor, info, w_r, qoi_r, kp); break;
    FR_ONE(1) FR_ONE(2) FR_ONE(3) FR_ONE(4) FR_ONE(5)
#undef FR_ONE
    default: set_error("rom_proj_single: basis size > 80"); return FINROM_ERR_UNSUPPORTED;
  }
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace finrom
