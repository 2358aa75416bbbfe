// ROM side of the hot path: batched least-squares Petrov-Galerkin reduced solve.
//
// Replaces AffineROMFin.forward_nine_param_reduced + .qoi_reduced
// (rom/averaged_affine_ROM.py:278-310, 323-333) for a batch of parameter vectors:
//     psi = (sum_p theta_p A_p + Bi M) Phi          :289-295  (assemble + AIJ x Dense)
//     A_r = psi^T psi ,  B_r = psi^T F              :296-297  (Dense^T x Dense, gemv)
//     w_r = solve(A_r, B_r)                         :304      (LAPACK dgesv in the reference)
//     qoi_r = (B_obs Phi) w_r                       :323-333
//
// rom_proj_kernel: one wave per sample.  psi is never materialised: each k-step builds a
// 4 x (16*NB) slab of psi in registers from the row-sparse tables Psi_p = A_p Phi (the
// reference's precomputed dA_dsigmak_phi, :215-220) and feeds it to v_mfma_f64_16x16x4_f64
// as BOTH operands (A = slab^T, B = slab hold the same lane values), accumulating only the
// upper block triangle of A_r (NB(NB+1)/2 tiles of 16x16).  Bound: fp64 MFMA issue.
#include "finrom_internal.h"

namespace finrom {

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NB>
__global__ __launch_bounds__(256) void rom_proj_kernel(RomDev p, const double* __restrict__ theta, int64_t S,
                                                       double* __restrict__ Ar, double* __restrict__ Br) {
  constexpr int NT = NB * (NB + 1) / 2;
  __shared__ double th[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t s = (int64_t)blockIdx.x * 4 + wave;
  if (s >= S) return;                       // no block-wide barrier below
  if (lane == 0) th[wave][0] = 1.0;
  if (lane < p.P) th[wave][lane + 1] = theta[s * p.P + lane];
  __builtin_amdgcn_wave_barrier();
  const int q = lane >> 4, c = lane & 15;

  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
  double bacc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) bacc[b] = 0.0;

  for (int ks = 0; ks < p.nk; ++ks) {
    double v[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) v[b] = 0.0;
    for (int t = p.kstep_ptr[ks], t1 = p.kstep_ptr[ks + 1]; t < t1; ++t) {
      const int2 sl = p.slot[t * 4 + q];
      const double thp = th[wave][sl.y];
      const double* src = p.term_val + sl.x + c;
#pragma unroll
      for (int b = 0; b < NB; ++b) v[b] = fma(thp, src[16 * b], v[b]);
    }
    const double fk = p.rhs4[ks * 4 + q];
#pragma unroll
    for (int b = 0; b < NB; ++b) bacc[b] = fma(v[b], fk, bacc[b]);
    int idx = 0;
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
      for (int tj = ti; tj < NB; ++tj) {
        acc[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[ti], v[tj], acc[idx], 0, 0, 0);
        ++idx;
      }
  }

  // C/D layout of v_mfma_f64_16x16x4_f64: lane holds D[row = (lane>>4) + 4*g][col = lane&15]
  double* A = Ar + s * (int64_t)p.rp * p.rp;
  int idx = 0;
#pragma unroll
  for (int ti = 0; ti < NB; ++ti)
#pragma unroll
    for (int tj = ti; tj < NB; ++tj) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = 16 * ti + q + 4 * g, col = 16 * tj + c;
        A[row * p.rp + col] = acc[idx][g];
        if (ti != tj) A[col * p.rp + row] = acc[idx][g];
      }
      ++idx;
    }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    double x = bacc[b];
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    if (q == 0) Br[s * p.rp + 16 * b + c] = x;
  }
}

int launch_rom_proj(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, hipStream_t st) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_ROM_PROJ, st);
  dim3 grid((unsigned)((S + 3) / 4)), block(256);
  switch (p.NB) {
#define FR_CASE(N) case N: hipLaunchKernelGGL(rom_proj_kernel<N>, grid, block, 0, st, p, theta, S, Ar, Br); break;
    FR_CASE(1) FR_CASE(2) FR_CASE(3) FR_CASE(4) FR_CASE(5) FR_CASE(6) FR_CASE(7) FR_CASE(8)
#undef FR_CASE
    default:
      set_error("rom_proj: basis size " + std::to_string(p.r) + " > 128 not supported yet");
      return FINROM_ERR_UNSUPPORTED;
  }
  FR_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// reduced solve: one workgroup per sample, A_r in LDS, right-looking Cholesky (A_r is SPD;
// the reference calls LAPACK dgesv, :304 -- same solution), two triangular solves, QoI.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rom_solve_kernel(RomDev p, const double* __restrict__ Ar,
                                                        const double* __restrict__ Br, int64_t S,
                                                        double* __restrict__ w_r, double* __restrict__ qoi_r,
                                                        double* __restrict__ Ar_out, double* __restrict__ Br_out,
                                                        int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int r = p.r, rp = p.rp, ld = rp + 1;
  double* A = sm;                 // [rp][ld]
  double* b = sm + rp * ld;       // [rp]
  int& bad_s = *reinterpret_cast<int*>(b + rp);   // keep ALL LDS in the dynamic region (16-B aligned base)
  const int tid = threadIdx.x;
  const int64_t s = blockIdx.x;
  const double* As = Ar + s * (int64_t)rp * rp;
  if (tid == 0) bad_s = 0;
  for (int t = tid; t < rp * rp; t += 256) A[(t / rp) * ld + (t % rp)] = As[t];
  for (int t = tid; t < rp; t += 256) b[t] = Br[s * rp + t];
  __syncthreads();
  if (Ar_out != nullptr)
    for (int t = tid; t < r * r; t += 256) Ar_out[s * (int64_t)r * r + t] = A[(t / r) * ld + (t % r)];
  if (Br_out != nullptr)
    for (int t = tid; t < r; t += 256) Br_out[s * r + t] = b[t];

  const int tx = tid & 15, ty = tid >> 4;
  for (int k = 0; k < r; ++k) {
    __syncthreads();
    const double d = A[k * ld + k];
    if (!(d > 0.0) && tid == 0) bad_s = 1;
    const double inv = 1.0 / sqrt(d);
    __syncthreads();
    for (int i = k + tid; i < r; i += 256) A[i * ld + k] = (i == k) ? sqrt(d) : A[i * ld + k] * inv;
    __syncthreads();
    for (int i = k + 1 + ty; i < r; i += 16) {
      const double lik = A[i * ld + k];
      for (int j = k + 1 + tx; j <= i; j += 16) A[i * ld + j] = fma(-lik, A[j * ld + k], A[i * ld + j]);
    }
  }
  // L y = b
  for (int k = 0; k < r; ++k) {
    __syncthreads();
    const double yk = b[k] / A[k * ld + k];
    __syncthreads();
    if (tid == 0) b[k] = yk;
    for (int i = k + 1 + tid; i < r; i += 256) b[i] = fma(-A[i * ld + k], yk, b[i]);
  }
  // L^T x = y
  for (int k = r - 1; k >= 0; --k) {
    __syncthreads();
    const double xk = b[k] / A[k * ld + k];
    __syncthreads();
    if (tid == 0) b[k] = xk;
    for (int i = tid; i < k; i += 256) b[i] = fma(-A[k * ld + i], xk, b[i]);
  }
  __syncthreads();
  const int bad = bad_s;
  const double nanv = __builtin_nan("");
  if (w_r != nullptr)
    for (int t = tid; t < r; t += 256) w_r[s * r + t] = bad ? nanv : b[t];
  for (int o = tid; o < p.n_obs; o += 256) {
    double qv = 0.0;
    for (int t = 0; t < r; ++t) qv = fma(p.obs_phi[o * r + t], b[t], qv);
    qoi_r[s * p.n_obs + o] = bad ? nanv : qv;
  }
  if (info != nullptr && tid == 0 && bad) info[s] |= 2;
}

int launch_rom_solve(const RomDev& p, const double* Ar, const double* Br, int64_t S, double* w_r, double* qoi_r,
                     double* Ar_out, double* Br_out, int* info, hipStream_t st) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_ROM_SOLVE, st);
  size_t lds = ((size_t)p.rp * (p.rp + 1) + p.rp + 2) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    FR_HIP(hipFuncSetAttribute((const void*)rom_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    attr_set = true;
  }
  hipLaunchKernelGGL(rom_solve_kernel, dim3((unsigned)S), dim3(256), lds, st, p, Ar, Br, S, w_r, qoi_r, Ar_out, Br_out, info);
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace finrom
