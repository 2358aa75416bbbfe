// Small helper kernels of the hot path: sub-fin averages, Gaussian-field sampler, difference.
#include "finrom_internal.h"

namespace finrom {

typedef double d4 __attribute__((ext_vector_type(4)));

// theta[s][p] = sum_j Sop[p][j] * k[s][j]   (fom/forward_solve.py:466-480, rom :404-418:
// nine whole-mesh scalar assembles per sample in the reference; here one wave per sample)
constexpr int MAXP = 16;
__global__ __launch_bounds__(256) void subfin_avg_kernel(const double* __restrict__ Sop, int P, int n,
                                                         const double* __restrict__ k, int64_t S,
                                                         double* __restrict__ theta) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t s = (int64_t)blockIdx.x * 4 + wave;
  if (s >= S) return;
  double acc[MAXP];
#pragma unroll
  for (int p = 0; p < MAXP; ++p) acc[p] = 0.0;
  const double* ks = k + s * n;
  for (int j = lane; j < n; j += 64) {
    const double kj = ks[j];
#pragma unroll
    for (int p = 0; p < MAXP; ++p)
      if (p < P) acc[p] = fma(Sop[(int64_t)p * n + j], kj, acc[p]);
  }
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    double x = acc[p];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    if (lane == 0 && p < P) theta[s * P + p] = x;
  }
}

int launch_subfin_avg(const double* Sop, int P, int n, const double* k, int64_t S, double* theta, hipStream_t st) {
  if (S == 0) return 0;
  if (P > MAXP) { set_error("subfin_avg: P > 16"); return FINROM_ERR_UNSUPPORTED; }
  ScopedKernelTimer t(K_AVG, st);
  hipLaunchKernelGGL(subfin_avg_kernel, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, st, Sop, P, n, k, S, theta);
  FR_HIP(hipGetLastError());
  return 0;
}

// k[s][j] = exp(0.5 * sum_{i<=j} xi[s][i] U[i][j])   (generate_fin_dataset.py:87-88; U upper)
// fp64 MFMA GEMM, 64 x 64 output tile per workgroup, K-chunks of 16 through LDS; only the
// K-range i < j0+64 of the upper-triangular factor is visited.
__global__ __launch_bounds__(256) void sampler_kernel(const double* __restrict__ U, int n,
                                                      const double* __restrict__ xi, int64_t S,
                                                      double* __restrict__ kout) {
  __shared__ double Xs[64][17];
  __shared__ double Us[16][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t s0 = (int64_t)blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;
  const int q = lane >> 4, c = lane & 15;
  d4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
  const int kend = min(n, j0 + 64);
  for (int k0 = 0; k0 < kend; k0 += 16) {
    {
      const int kk = tid & 15, rr = tid >> 4;           // 16 x 16 threads -> 64 rows in 4 passes
#pragma unroll
      for (int pss = 0; pss < 4; ++pss) {
        const int row = rr + 16 * pss;
        const int64_t s = s0 + row;
        Xs[row][kk] = (s < S && k0 + kk < n) ? xi[s * n + k0 + kk] : 0.0;
      }
      const int cc = tid & 63, kr = tid >> 6;           // 64 cols x 4 rows -> 16 rows in 4 passes
#pragma unroll
      for (int pss = 0; pss < 4; ++pss) {
        const int kx = kr + 4 * pss;
        Us[kx][cc] = (k0 + kx < n && j0 + cc < n) ? U[(int64_t)(k0 + kx) * n + j0 + cc] : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const double a = Xs[16 * wave + c][4 * ks + q];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Us[4 * ks + q][16 * t + c], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t s = s0 + 16 * wave + q + 4 * g;
      const int j = j0 + 16 * t + c;
      if (s < S && j < n) kout[s * n + j] = exp(0.5 * acc[t][g]);
    }
}

int launch_sampler(const double* U, int n, const double* xi, int64_t S, double* k, hipStream_t st) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_SAMPLER, st);
  dim3 grid((unsigned)((S + 63) / 64), (unsigned)((n + 63) / 64));
  hipLaunchKernelGGL(sampler_kernel, grid, dim3(256), 0, st, U, n, xi, S, k);
  FR_HIP(hipGetLastError());
  return 0;
}

__global__ void sub_kernel(const double* __restrict__ a, const double* __restrict__ b, int64_t count,
                           double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = a[i] - b[i];
}

int launch_sub(const double* a, const double* b, int64_t count, double* out, hipStream_t st) {
  if (count == 0) return 0;
  ScopedKernelTimer t(K_MISC, st);
  int64_t blocks = (count + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sub_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, count, out);
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace finrom
