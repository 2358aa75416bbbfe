// Small helper kernels of the hot path: sub-fin averages, Gaussian-field sampler, difference.
#include "finrom_internal.h"
#include <algorithm>

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
  // (four passes of 64 columns requested together -- one value of k and P of S per pass -- and added in the order of the
  // columns, as before: the first version waited for every pass on its own, 25 dependent trips to memory for a 1597-node field --
  // 34 us for one sample, and 0.4 TB/s for a batch of them)
  constexpr int UP = 4;
  for (int j0 = lane; j0 < n; j0 += 64 * UP) {
    double kv[UP], sv[MAXP][UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) kv[u] = j0 + 64 * u < n ? ks[j0 + 64 * u] : 0.0;
#pragma unroll
    for (int p = 0; p < MAXP; ++p)
      if (p < P) {
#pragma unroll
        for (int u = 0; u < UP; ++u) sv[p][u] = j0 + 64 * u < n ? Sop[(int64_t)p * n + j0 + 64 * u] : 0.0;
      }
#pragma unroll
    for (int u = 0; u < UP; ++u)
#pragma unroll
      for (int p = 0; p < MAXP; ++p)
        if (p < P && j0 + 64 * u < n) acc[p] = fma(sv[p][u], kv[u], acc[p]);
  }
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    if (p < P) {                                         // (wave-uniform: the unused sums are not reduced)
      double x = acc[p];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
      if (lane == 0) theta[s * P + p] = x;
    }
  }
}

// The same product for parameter vectors (n = 5 / 9 conductivities per sample, the dataset loop's five- and nine-parameter forms):
// a wave per sample would spend its time reducing 64 lanes of which n are busy (0.2 ms per 100k samples at the head of the ROM
// half's critical path) -- here a thread owns a sample, S (P x n doubles) sits in LDS, sums run over j in index order.
constexpr int SUBFIN_SMALL_N = 16;
__global__ __launch_bounds__(256) void subfin_avg_small_kernel(const double* __restrict__ Sop, int P, int n,
                                                               const double* __restrict__ k, int64_t S,
                                                               double* __restrict__ theta) {
  __shared__ double sl[MAXP * SUBFIN_SMALL_N];
  for (int t = threadIdx.x; t < P * n; t += 256) sl[t] = Sop[t];
  __syncthreads();
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  double kv[SUBFIN_SMALL_N];
#pragma unroll
  for (int j = 0; j < SUBFIN_SMALL_N; ++j) kv[j] = j < n ? k[s * n + j] : 0.0;
  for (int p = 0; p < P; ++p) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < SUBFIN_SMALL_N; ++j)
      if (j < n) acc = fma(sl[p * n + j], kv[j], acc);
    theta[s * P + p] = acc;
  }
}

int launch_subfin_avg(const double* Sop, int P, int n, const double* k, int64_t S, double* theta, hipStream_t st) {
  if (S == 0) return 0;
  if (P > MAXP) { set_error("subfin_avg: P > 16"); return FINROM_ERR_UNSUPPORTED; }
  ScopedKernelTimer t(K_AVG, st);
  if (n <= SUBFIN_SMALL_N)
    hipLaunchKernelGGL(subfin_avg_small_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, Sop, P, n, k, S, theta);
  else
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

// ---------------------------------------------------------------------------------------
// xi ~ N(0, I) on the device, keyed by the GLOBAL sample index (deep_learning/generate_fin_dataset.py:87 draws them with
// np.random.randn): Philox4x32-10, counter = (sample index lo, hi, pair index, 0), key = seed; the four output words make
// two 53-bit uniforms and Box-Muller turns them into the pair xi[2 jb], xi[2 jb + 1].  A sample's draw depends on nothing but
// (seed, its global index): any sharding of a dataset over GPUs produces the same fields.
// oracle/fin_oracle.py::philox_normal is the NumPy restatement the tests compare with.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__global__ __launch_bounds__(256) void philox_normal_kernel(unsigned long long seed, int64_t first, int64_t S, int n,
                                                            double* __restrict__ xi) {
  const int npair = (n + 1) / 2;
  const int64_t total = S * npair;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t sl = i / npair;
    const int jb = (int)(i - sl * npair);
    const unsigned long long g = (unsigned long long)(first + sl);
    unsigned o[4];
    philox4x32_10((unsigned)g, (unsigned)(g >> 32), (unsigned)jb, 0u, (unsigned)seed, (unsigned)(seed >> 32), o);
    const unsigned long long a = ((unsigned long long)o[1] << 32) | o[0], b = ((unsigned long long)o[3] << 32) | o[2];
    const double u1 = (double)((a >> 11) + 1) * 0x1.0p-53;       // (0, 1]
    const double u2 = (double)(b >> 11) * 0x1.0p-53;             // [0, 1)
    const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586476925286766559 * u2;
    xi[sl * n + 2 * jb] = rad * cos(ang);
    if (2 * jb + 1 < n) xi[sl * n + 2 * jb + 1] = rad * sin(ang);
  }
}

int launch_philox_normal(unsigned long long seed, int64_t first, int64_t S, int n, double* xi, hipStream_t st) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_MISC, st);
  const int64_t total = S * ((n + 1) / 2);
  const int64_t blocks = std::min<int64_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)blocks), dim3(256), 0, st, seed, first, S, n, xi);
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
