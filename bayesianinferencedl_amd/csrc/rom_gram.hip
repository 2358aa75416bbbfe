// Offline/online form of the reduced operator (opt-in, finrom_rom_set_projection(h, FINROM_PROJECTION_GRAM)).
// psi = A(theta) Phi = sum_p theta_p Psi_p is affine in theta (rom/averaged_affine_ROM.py:282-290, theta_0 = 1 for the Robin
// term), hence
//     A_r = psi^T psi = sum_{p <= q} theta_p theta_q G_pq,   G_pq = Psi_p^T Psi_q (+ its transpose for p != q),
//     B_r = psi^T F   = sum_p theta_p h_p,                    h_p  = Psi_p^T F,
// with G_pq, h_p computed once (the reference precomputes Psi_p itself, :215-220).  Per sample this is npairs * r(r+1)/2
// multiply-adds instead of n r (r+1): 0.2 MFLOP instead of 11.1 at r = 80.  Same results to round-off (parity tests run both
// forms against the same oracle); NOT the contraction the reference executes per sample, which is why it is not the default.
//
// rom_gram_kernel<NB> (r <= 96): wave = sample, the tiles of A_r accumulate in the C/D register layout of the MFMA
// projection kernel, so the in-register Cholesky / substitutions / QoI of rom_proj_device.h follow unchanged.
// rom_gram_store_kernel (any r): thread = packed entry of A_r for 32 samples, written to the scratch the blocked Cholesky
// and the substitution kernel read.
#include "rom_proj_device.h"

namespace finrom {

// everything after the tiles of A_r are complete (same steps as the projection kernel's epilogue)
template <int NB>
__device__ __forceinline__ void gram_finish(const RomDev& p, const RomGramDev& gm, const double* thw, int64_t s, int lane,
                                            d4 (&acc)[NB * (NB + 1) / 2], double* __restrict__ Ar, double* __restrict__ Br,
                                            int factor, int* __restrict__ info, double* __restrict__ w_r, double* __restrict__ qoi_r) {
  const int q = lane >> 4, c = lane & 15;
  int bad = 0;
  if (factor) {      // A_r = U^T U in registers; what is written below is then U^T (= L, packed by columns)
    bad = chol_tiles<NB>(acc, q, c, p.r);
    if (bad && info != nullptr && lane == 0) atomicOr(&info[s], 2);
  }
  const bool fused_solve = NB <= 5 && factor == 2;
  const int R = p.rp;
  if (!fused_solve) {      // packed store, same element mapping as the projection kernel's
    double* A = Ar + s * (int64_t)(R * (R + 1) / 2);
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
      for (int tj = ti; tj < NB; ++tj) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * ti + q + 4 * g, col = 16 * tj + c;
          if (ti != tj || col >= row) A[row * R - (row * (row - 1)) / 2 + col - row] = acc[tile_index<NB>(ti, tj)][g];
        }
      }
  }
  double bacc[NB];     // B_r[16 b + c], the same in all four row groups
#pragma unroll
  for (int b = 0; b < NB; ++b) bacc[b] = 0.0;
#pragma unroll 1
  for (int pp = 0; pp <= p.P; ++pp) {
    const double thp = thw[pp];
#pragma unroll
    for (int b = 0; b < NB; ++b) bacc[b] = fma(thp, gm.h[pp * R + 16 * b + c], bacc[b]);
  }
  if (!fused_solve && q == 0) {
#pragma unroll
    for (int b = 0; b < NB; ++b) Br[s * R + 16 * b + c] = bacc[b];
  }
  if constexpr (NB <= 5) {
    if (fused_solve) {
      double xr[NB];
      solve_tiles<NB>(acc, bacc, q, c, xr);
      const double nanv = __builtin_nan("");
      if (w_r != nullptr && q == 0) {
#pragma unroll
        for (int t = 0; t < NB; ++t)
          if (16 * t + c < p.r) w_r[s * p.r + 16 * t + c] = bad ? nanv : xr[t];
      }
      if (qoi_r != nullptr)      // qoi_r = (B_obs Phi) w_r (rom :323-333): row group q takes observations q, q + 4, ...
        for (int o0 = 0; o0 < p.n_obs; o0 += 4) {
          const int o = o0 + q;
          double d = 0.0;
#pragma unroll
          for (int t = 0; t < NB; ++t)
            if (o < p.n_obs && 16 * t + c < p.r) d = fma(p.obs_phi[o * p.r + 16 * t + c], xr[t], d);
          d += __shfl_xor(d, 8); d += __shfl_xor(d, 4); d += __shfl_xor(d, 2); d += __shfl_xor(d, 1);
          if (o < p.n_obs && c == 0) qoi_r[s * p.n_obs + o] = bad ? nanv : d;
        }
    }
  }
}

// NS samples per wave: a tile image of G is loaded once and applied to NS accumulator sets (the kernel's memory traffic is
// the G tiles, npairs * NB(NB+1)/2 * 2 KiB per wave from L2, and it runs beside the HBM-bound FOM interpreter); the
// factorisations and substitutions then run one sample after the other.  NS = 2 only where both accumulator sets and the
// factorisation fit 256 VGPRs (r <= 48): at r = 80 it needs 328 registers = one wave per SIMD, measured 8.9 ms per 100k
// samples stand-alone against 4.8 ms for NS = 1 at two waves per SIMD.
template <int NB, int NS>
__global__ __launch_bounds__(256, 2) void rom_gram_kernel(RomDev p, RomGramDev gm, const double* __restrict__ theta, int64_t S,
                                                          double* __restrict__ Ar, double* __restrict__ Br, int factor,
                                                          int* __restrict__ info, double* __restrict__ w_r,
                                                          double* __restrict__ qoi_r) {
  constexpr int NT = NB * (NB + 1) / 2;
  __shared__ double th[4][NS][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t s0 = ((int64_t)blockIdx.x * 4 + wave) * NS;
  if (s0 >= S) return;                      // no block-wide barrier below
  int64_t sidx[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    sidx[u] = s0 + u < S ? s0 + u : S - 1;  // a ragged tail repeats the last sample (same values stored twice)
    if (lane == 0) th[wave][u][0] = 1.0;
    if (lane < p.P) th[wave][u][lane + 1] = theta[sidx[u] * p.P + lane];
  }
  __builtin_amdgcn_wave_barrier();

  d4 acc[NS][NT];
#pragma unroll
  for (int u = 0; u < NS; ++u)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[u][t] = (d4){0.0, 0.0, 0.0, 0.0};
  // tile images [pair][tile][lane] of 4 doubles: one 32-byte load per lane, tile and pair, 2 KiB contiguous per wave
  const d4* __restrict__ G = reinterpret_cast<const d4*>(gm.Gt) + lane;
#pragma unroll 1
  for (int pr = 0; pr < gm.npairs; ++pr) {
    const int pp = gm.pair_p[pr], pq = gm.pair_q[pr];
    double cf[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) cf[u] = th[wave][u][pp] * th[wave][u][pq];
    const d4* __restrict__ Gp = G + (int64_t)pr * NT * 64;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const d4 g = Gp[t * 64];
#pragma unroll
      for (int u = 0; u < NS; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[u][t][k] = fma(cf[u], g[k], acc[u][t][k]);
    }
  }
  gram_finish<NB>(p, gm, th[wave][0], sidx[0], lane, acc[0], Ar, Br, factor, info, w_r, qoi_r);
  if constexpr (NS > 1) gram_finish<NB>(p, gm, th[wave][1], sidx[1], lane, acc[1], Ar, Br, factor, info, w_r, qoi_r);
  static_assert(NS <= 2, "one call per sample, written out so that the accumulators stay in registers");
}

// A_r for wide bases: grid (sample groups of 32, chunks of 256 packed entries).  Workgroups that run together share a chunk
// of G (x is the fast grid dimension), so G is read from L2; the coefficient of a pair is the same for the whole wave and
// comes from LDS as a broadcast read.
constexpr int GRAM_SPB = 32;
__global__ __launch_bounds__(256) void rom_gram_store_kernel(RomDev p, RomGramDev gm, const double* __restrict__ theta, int64_t S,
                                                             double* __restrict__ Ar, double* __restrict__ Br) {
  __shared__ __attribute__((aligned(16))) double coef[ROM_GRAM_MAX_PAIRS][GRAM_SPB];
  __shared__ double th[GRAM_SPB][33];
  const int tid = threadIdx.x;
  const int64_t s0 = (int64_t)blockIdx.x * GRAM_SPB;
  const int ns = (int)(S - s0 < GRAM_SPB ? S - s0 : GRAM_SPB);
  const int R = p.rp, np = R * (R + 1) / 2;
  for (int i = tid; i < GRAM_SPB * (p.P + 1); i += 256) {
    const int j = i / (p.P + 1), k = i - j * (p.P + 1);
    th[j][k] = k == 0 ? 1.0 : (j < ns ? theta[(s0 + j) * p.P + k - 1] : 0.0);
  }
  __syncthreads();
  for (int i = tid; i < gm.npairs * GRAM_SPB; i += 256) {
    const int pr = i / GRAM_SPB, j = i - pr * GRAM_SPB;
    coef[pr][j] = th[j][gm.pair_p[pr]] * th[j][gm.pair_q[pr]];
  }
  __syncthreads();
  const int e = blockIdx.y * 256 + tid;
  if (e < np) {
    double acc[GRAM_SPB];
#pragma unroll
    for (int j = 0; j < GRAM_SPB; ++j) acc[j] = 0.0;
#pragma unroll 1
    for (int pr = 0; pr < gm.npairs; ++pr) {
      const double g = gm.Gp[(int64_t)pr * np + e];
      const d4* cf = reinterpret_cast<const d4*>(coef[pr]);
#pragma unroll
      for (int j4 = 0; j4 < GRAM_SPB / 4; ++j4) {
        const d4 c4 = cf[j4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[4 * j4 + k] = fma(c4[k], g, acc[4 * j4 + k]);
      }
    }
#pragma unroll
    for (int j = 0; j < GRAM_SPB; ++j)
      if (j < ns) Ar[(s0 + j) * (int64_t)np + e] = acc[j];
  }
  if (blockIdx.y == 0)
    for (int i = tid; i < ns * R; i += 256) {
      const int j = i / R, col = i - j * R;
      double b = 0.0;
      for (int pp = 0; pp <= p.P; ++pp) b = fma(th[j][pp], gm.h[pp * R + col], b);
      Br[(s0 + j) * R + col] = b;
    }
}

int launch_rom_gram(const RomDev& p, const RomGramDev& gm, const double* theta, int64_t S, double* Ar, double* Br, int factor,
                    int* info, hipStream_t st, double* w_r, double* qoi_r) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_ROM_PROJ, st);
  if (p.NB <= 6) {
    const dim3 block(256);
    switch (p.NB) {
#define FR_ONE(N, NS) case N: hipLaunchKernelGGL((rom_gram_kernel<N, NS>), dim3((unsigned)((S + 4 * NS - 1) / (4 * NS))), block, 0, st, p, gm, \
                                                 theta, S, Ar, Br, factor, info, w_r, qoi_r); break;
      FR_ONE(1, 2) FR_ONE(2, 2) FR_ONE(3, 2) FR_ONE(4, 1) FR_ONE(5, 1) FR_ONE(6, 1)
#undef FR_ONE
    }
  } else {
    const int np = p.rp * (p.rp + 1) / 2;
    const dim3 grid((unsigned)((S + GRAM_SPB - 1) / GRAM_SPB), (unsigned)((np + 255) / 256)), block(256);
    hipLaunchKernelGGL(rom_gram_store_kernel, grid, block, 0, st, p, gm, theta, S, Ar, Br);
  }
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace finrom
