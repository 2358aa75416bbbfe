// The single-wave projection kernels (r <= 80: one wave owns a sample, factorisation + substitutions in registers) in their
// own translation unit, built at -O2 (see _build.py): hipcc's -O3 inflates their register use (r = 80: 188 -> 256 VGPRs +
// scratch), and they must stay small enough to share a SIMD with the FOM interpreter's waves: 2 x 192 registers of the
// r = 80 kernel and 2 x 56 of the interpreter fill a SIMD's 512 exactly.
#include "rom_proj_device.h"

namespace finrom {

// One sample of one wave.  NOT inlined into the persistent loop below: inlined, hipcc's loop passes stretch live ranges across
// iterations (r = 80: 188 -> 256 VGPRs + 308 B of scratch per lane, spills inside the MFMA loop).
template <int NB>
__device__ __attribute__((noinline)) void proj_one_sample(const RomDev& p, const double* thw, int64_t s, int lane,
                                                          double* __restrict__ Ar, double* __restrict__ Br, int factor,
                                                          int* __restrict__ info, double* __restrict__ w_r,
                                                          double* __restrict__ qoi_r) {
  rom_proj_body<NB, 1, 0>(p, thw, s, lane, Ar, Br, factor, info, w_r, qoi_r);
}

// Persistent form: a workgroup's four waves each walk over samples s, s + 4 * gridDim.x, ... -- the grid is sized to what
// is resident at once (2 workgroups per CU at r = 80), so the two waves that share a SIMD (= its ONE fp64 MFMA pipe) stay
// paired for the whole launch.  Launched together they run in lockstep: both in the MFMA main loop (each at half rate), then
// both in the MFMA-free epilogue (in-register Cholesky, substitutions) with the pipe idle -- measured 0.69 pipe occupancy.
// STAGGER: the workgroup that arrives second on its CU (atomic ticket per CU, keyed by XCC_ID / HW_ID) first sleeps for
// about one main loop; from then on one wave's epilogue runs beside its partner's main loop (MI355X_MICROARCH.md, "Two
// waves per SIMD", item 9).  Placement is never assumed for correctness: a wrong guess only costs the overlap.
template <int NB>
__global__ __launch_bounds__(256, 2) void rom_proj_single_kernel(RomDev p, const double* __restrict__ theta, int64_t S,
                                                                 double* __restrict__ Ar, double* __restrict__ Br, int factor,
                                                                 int* __restrict__ info, double* __restrict__ w_r,
                                                                 double* __restrict__ qoi_r, int* __restrict__ cu_ticket,
                                                                 int stagger) {
  __shared__ double th[4][32];
  __shared__ int order;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  trace_begin(p.trace, blockIdx.x);
  if (cu_ticket != nullptr) {
    if (threadIdx.x == 0) {
      const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));       // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
      const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));     // HW_REG_XCC_ID
      order = atomicAdd(&cu_ticket[((xcc & 15u) << 8) | ((hw >> 8) & 255u)], 1);
    }
    __syncthreads();
    if (order & 1)
      for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < S; s += (int64_t)gridDim.x * 4) {
    if (lane == 0) th[wave][0] = 1.0;
    if (lane < p.P) th[wave][lane + 1] = theta[s * p.P + lane];
    __builtin_amdgcn_wave_barrier();
    proj_one_sample<NB>(p, th[wave], s, lane, Ar, Br, factor, info, w_r, qoi_r);
    __builtin_amdgcn_wave_barrier();
  }
  trace_end(p.trace, blockIdx.x);
}

int launch_rom_proj_single(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                           hipStream_t st, double* w_r, double* qoi_r, int* cu_ticket) {
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0; hipDeviceProp_t prop;
    FR_HIP(hipGetDevice(&dev)); FR_HIP(hipGetDeviceProperties(&prop, dev));
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int64_t want = (S + 3) / 4;
  switch (p.NB) {
#define FR_ONE(N) case N: {                                                                                            \
      int per_cu = 2;                                                                                                  \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rom_proj_single_kernel<N>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2; \
      const int64_t resident = (int64_t)per_cu * num_cu;                                                               \
      const bool persistent = want > resident && getenv("FINROM_PROJ_NOT_PERSISTENT") == nullptr;                      \
      const dim3 grid((unsigned)(persistent ? resident : want)), block(256);                                           \
      /* one main loop of a lone wave: k-steps x tiles x 64 cycles per fp64 MFMA, in s_sleep(127) units of 8128 cycles */ \
      int stagger = 0;                                                                                                 \
      if (persistent && per_cu >= 2 && cu_ticket != nullptr && getenv("FINROM_PROJ_NO_STAGGER") == nullptr) {          \
        int nks = 0; for (int ph = 0; ph < p.n_phases; ++ph) nks += p.phase_ks1[ph] - p.phase_ks0[ph];                 \
        stagger = (int)((int64_t)nks * (N * (N + 1) / 2) * 64 / 8128) + 1;                                             \
        if (const char* e = getenv("FINROM_PROJ_STAGGER")) stagger = atoi(e);                                          \
        FR_HIP(hipMemsetAsync(cu_ticket, 0, 4096 * sizeof(int), st));                                                  \
      }                                                                                                                \
      hipLaunchKernelGGL(rom_proj_single_kernel<N>, grid, block, 0, st, p, theta, S, Ar, Br, factor, info, w_r, qoi_r, \
                         stagger > 0 ? cu_ticket : nullptr, stagger); } break;
    FR_ONE(1) FR_ONE(2) FR_ONE(3) FR_ONE(4) FR_ONE(5)
#undef FR_ONE
    default: set_error("rom_proj_single: basis size > 80"); return FINROM_ERR_UNSUPPORTED;
  }
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace finrom
