// The trajectory bookkeeping of the HMC rehearsal (BASELINE configs[4]) on the device, so that a proposal is library launches
// only: what PyMC3's sampler does around the reference's value-and-gradient op (bayesian_inference/pymc_func_bayes_inverse.py:
// 148-167 `SqErrorOpROMML.perform` / `.grad`, model :186-203) -- draw a momentum, leapfrog, Metropolis test -- restated for the
// i.i.d. Gaussian prior of bayesian_inference/hmc.py.  One workgroup per chain; sums over the field in a fixed order.
#include "finrom_internal.h"

namespace finrom {

namespace {

__device__ __forceinline__ double block_sum_256(double v, double* red) {      // 256 threads, fixed order
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// proposal jt of the uploaded block: momentum P0, Hamiltonian at the start, first half step of the momentum
__global__ __launch_bounds__(256) void hmc_begin_kernel(HmcDev h) {
  __shared__ double red[4];
  const int64_t c = blockIdx.x, j = *h.jt;
  const double* __restrict__ p0 = h.P_block + (j * h.C + c) * h.n;
  const int64_t o = c * h.n;
  double s = 0.0;
  for (int i = threadIdx.x; i < h.n; i += 256) {
    const double p = p0[i];
    s = fma(p, p, s);
    h.Kq0[o + i] = h.K[o + i];
    const double du = h.dU[o + i];
    h.dUq[o + i] = du;
    h.P[o + i] = fma(-0.5 * h.eps * h.c_pri, du, p);
  }
  const double pp = block_sum_256(s, red);
  if (threadIdx.x == 0) h.H0[c] = h.U[c] + 0.5 * pp;
}

// end of the trajectory: the last momentum update was a whole step (back to a half), potential and Hamiltonian at the end point,
// Metropolis test against the uploaded log-uniform, state update, trace row
__global__ __launch_bounds__(256) void hmc_end_kernel(HmcDev h, const double* __restrict__ Kq) {
  __shared__ double red[4];
  __shared__ int ok_s;
  const int64_t c = blockIdx.x, j = *h.jt, row = *h.pt + 1;
  const int64_t o = c * h.n;
  double sp = 0.0, sd = 0.0;
  for (int i = threadIdx.x; i < h.n; i += 256) {
    const double p = fma(0.5 * h.eps * h.c_pri, h.dUq[o + i], h.P[o + i]);
    const double d = Kq[o + i] - h.mean[o + i];
    sp = fma(p, p, sp); sd = fma(d, d, sd);
  }
  const double pp = block_sum_256(sp, red), dd = block_sum_256(sd, red);
  if (threadIdx.x == 0) {
    double Uq = fma(h.c_lik, h.loss[c], 0.5 * h.c_pri * dd);
    if (h.info[c] != 0 || !(Uq == Uq) || Uq > 1.7e308 || Uq < -1.7e308) Uq = __builtin_inf();
    const double H1 = Uq + 0.5 * pp;
    const int ok = h.lu_block[j * h.C + c] < h.H0[c] - H1;      // (H1 = inf or nan compares false: rejected)
    if (ok) h.U[c] = Uq;
    h.accept[c] += ok;
    ok_s = ok;
  }
  __syncthreads();
  const bool ok = ok_s != 0;
  for (int i = threadIdx.x; i < h.n; i += 256) {
    if (ok) { h.K[o + i] = Kq[o + i]; h.dU[o + i] = h.dUq[o + i]; }
    if (h.trace != nullptr) h.trace[(row * h.C + c) * h.n + i] = ok ? Kq[o + i] : h.K[o + i];
  }
}

__global__ void hmc_advance_kernel(long long* jt, long long* pt) { *jt += 1; *pt += 1; }

}  // namespace

int launch_hmc_begin(const HmcDev& h, hipStream_t st) {
  if (h.C == 0) return 0;
  ScopedKernelTimer t(K_MISC, st);
  hipLaunchKernelGGL(hmc_begin_kernel, dim3((unsigned)h.C), dim3(256), 0, st, h);
  FR_HIP(hipGetLastError());
  return 0;
}

int launch_hmc_end(const HmcDev& h, const double* Kq, hipStream_t st) {
  if (h.C == 0) return 0;
  ScopedKernelTimer t(K_MISC, st);
  hipLaunchKernelGGL(hmc_end_kernel, dim3((unsigned)h.C), dim3(256), 0, st, h, Kq);
  FR_HIP(hipGetLastError());
  hipLaunchKernelGGL(hmc_advance_kernel, dim3(1), dim3(1), 0, st, h.jt, h.pt);
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace finrom
