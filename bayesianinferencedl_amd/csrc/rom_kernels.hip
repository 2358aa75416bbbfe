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

  // C/D layout of v_mfma_f64_16x16x4_f64: lane holds D[row = (lane>>4) + 4*g][col = lane&15].
  // A_r is symmetric: tile (ti <= tj) element (row, col) is written as the LOWER element
  // (i = col, k = row) of the packed column-major lower triangle the solve kernel reads.
  const int R = p.rp;
  double* A = Ar + s * (int64_t)(R * (R + 1) / 2);
  int idx = 0;
#pragma unroll
  for (int ti = 0; ti < NB; ++ti)
#pragma unroll
    for (int tj = ti; tj < NB; ++tj) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = 16 * ti + q + 4 * g, col = 16 * tj + c;
        if (ti != tj || col >= row) A[row * R - (row * (row - 1)) / 2 + col - row] = acc[idx][g];
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
// reduced solve: one WAVE per sample (workgroup = 64 threads), lane = row of A_r.
// A_r arrives as the packed lower triangle, stored by columns (column k holds rows k..R-1
// contiguously, so a wave reading column k over its rows is conflict-free in LDS).
// Left-looking Cholesky in panels of 4 columns: per k the wave does 1 vector + 4 broadcast
// LDS reads for 4 (8 with the second row set) multiply-adds; pivots travel by lane shuffles.
// A_r is SPD; the reference calls LAPACK dgesv (rom :304) -- same solution.
// Rows r..R-1 are padding (zero rows of psi^T psi): they get a unit diagonal.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int col_start(int k, int R) { return k * R - (k * (k - 1)) / 2; }

__global__ __launch_bounds__(64) void rom_solve_kernel(RomDev p, const double* __restrict__ Arp,
                                                       const double* __restrict__ Br, int64_t S,
                                                       double* __restrict__ w_r, double* __restrict__ qoi_r,
                                                       double* __restrict__ Ar_out, double* __restrict__ Br_out,
                                                       int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int r = p.r, R = p.rp, np = R * (R + 1) / 2;
  double* Lm = sm;            // [np] packed lower triangle, column-major
  double* invd = sm + np;     // [R]
  double* xs = invd + R;      // [R]
  const int lane = threadIdx.x;
  const int64_t s = blockIdx.x;
  const double* As = Arp + s * (int64_t)np;
  {   // np = R(R+1)/2 is even for R a multiple of 16: copy as 16-byte vectors, 8 loads in flight
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* src = reinterpret_cast<const d2*>(As);
    d2* dst = reinterpret_cast<d2*>(Lm);
    const int nv = np / 2;
    int t = lane;
    for (; t + 7 * 64 < nv; t += 8 * 64) {
      d2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[t + u * 64];
#pragma unroll
      for (int u = 0; u < 8; ++u) dst[t + u * 64] = v[u];
    }
    for (; t < nv; t += 64) dst[t] = src[t];
  }
  __syncthreads();
  for (int i = r + lane; i < R; i += 64) Lm[col_start(i, R)] = 1.0;
  const int i0 = lane, i1 = lane + 64;
  const bool v1 = i1 < R;
  double b0 = (i0 < R) ? Br[s * R + i0] : 0.0;
  double b1 = v1 ? Br[s * R + i1] : 0.0;
  if (Ar_out != nullptr)
    for (int t = lane; t < r * r; t += 64) {
      const int i = t / r, j = t - i * r;
      const int hi = i > j ? i : j, lo = i > j ? j : i;
      Ar_out[s * (int64_t)r * r + t] = Lm[col_start(lo, R) + hi - lo];
    }
  if (Br_out != nullptr) {
    if (i0 < r) Br_out[s * r + i0] = b0;
    if (i1 < r) Br_out[s * r + i1] = b1;
  }
  __syncthreads();

  int bad = 0;
  for (int j0 = 0; j0 < R; j0 += 4) {
    double a0[4], a1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int pc = j0 + c;
      a0[c] = (i0 >= pc && i0 < R) ? Lm[col_start(pc, R) + i0 - pc] : 0.0;
      a1[c] = (v1 && i1 >= pc) ? Lm[col_start(pc, R) + i1 - pc] : 0.0;
    }
    const bool act0 = i0 >= j0 && i0 < R;
#pragma unroll 4
    for (int k = 0; k < j0; ++k) {
      const int ck = col_start(k, R) - k;
      const double l0 = act0 ? Lm[ck + i0] : 0.0;
      const double l1 = (v1 && i1 >= j0) ? Lm[ck + i1] : 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double bc = Lm[ck + j0 + c];
        a0[c] = fma(-l0, bc, a0[c]);
        a1[c] = fma(-l1, bc, a1[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int pc = j0 + c;
      const double piv = (pc < 64) ? __shfl(a0[c], pc) : __shfl(a1[c], pc - 64);
      bad |= !(piv > 0.0);
      const double d = sqrt(piv), inv = 1.0 / d;
      const double l0 = (i0 > pc) ? a0[c] * inv : ((i0 == pc) ? d : 0.0);
      const double l1 = (i1 > pc) ? a1[c] * inv : ((i1 == pc) ? d : 0.0);
      const int cp = col_start(pc, R) - pc;
      if (i0 >= pc && i0 < R) Lm[cp + i0] = l0;
      if (v1 && i1 >= pc) Lm[cp + i1] = l1;
      if (lane == 0) invd[pc] = inv;
#pragma unroll
      for (int c2 = c + 1; c2 < 4; ++c2) {
        const int p2 = j0 + c2;
        const double lp = (p2 < 64) ? __shfl(l0, p2) : __shfl(l1, p2 - 64);
        a0[c2] = fma(-l0, lp, a0[c2]);
        a1[c2] = fma(-l1, lp, a1[c2]);
      }
    }
    __syncthreads();
  }
  // L y = b   (lane = row; y_k broadcast by shuffle)
  for (int k = 0; k < R; ++k) {
    const double bk = (k < 64) ? __shfl(b0, k) : __shfl(b1, k - 64);
    const double yk = bk * invd[k];
    const int ck = col_start(k, R) - k;
    if (i0 == k) b0 = yk; else if (i0 > k && i0 < R) b0 = fma(-Lm[ck + i0], yk, b0);
    if (i1 == k) b1 = yk; else if (v1 && i1 > k) b1 = fma(-Lm[ck + i1], yk, b1);
  }
  // L^T x = y
  for (int k = R - 1; k >= 0; --k) {
    const double bk = (k < 64) ? __shfl(b0, k) : __shfl(b1, k - 64);
    const double xk = bk * invd[k];
    if (i0 == k) b0 = xk; else if (i0 < k) b0 = fma(-Lm[col_start(i0, R) + k - i0], xk, b0);
    if (i1 == k) b1 = xk; else if (v1 && i1 < k) b1 = fma(-Lm[col_start(i1, R) + k - i1], xk, b1);
  }
  const double nanv = __builtin_nan("");
  if (i0 < R) xs[i0] = b0;
  if (v1) xs[i1] = b1;
  __syncthreads();
  if (w_r != nullptr) {
    if (i0 < r) w_r[s * r + i0] = bad ? nanv : b0;
    if (i1 < r) w_r[s * r + i1] = bad ? nanv : b1;
  }
  for (int o = lane; o < p.n_obs; o += 64) {
    double qv = 0.0;
    for (int t = 0; t < r; ++t) qv = fma(p.obs_phi[o * r + t], xs[t], qv);
    qoi_r[s * p.n_obs + o] = bad ? nanv : qv;
  }
  if (info != nullptr && lane == 0 && bad) info[s] |= 2;
}

int launch_rom_solve(const RomDev& p, const double* Ar, const double* Br, int64_t S, double* w_r, double* qoi_r,
                     double* Ar_out, double* Br_out, int* info, hipStream_t st) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_ROM_SOLVE, st);
  size_t lds = ((size_t)p.rp * (p.rp + 1) / 2 + 2 * (size_t)p.rp) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    FR_HIP(hipFuncSetAttribute((const void*)rom_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    attr_set = true;
  }
  hipLaunchKernelGGL(rom_solve_kernel, dim3((unsigned)S), dim3(64), lds, st, p, Ar, Br, S, w_r, qoi_r, Ar_out, Br_out, info);
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace finrom
