// Sustained fp64 MFMA rate: back-to-back v_mfma_f64_16x16x4_f64 on 15 accumulators (the r = 80 projection's tile count), random
// operands, 1 or 2 waves per SIMD on every CU, for tens of milliseconds -- the rate a long MFMA-bound kernel can reach once
// the chip has settled its clock under this load (MI355X_MICROARCH.md, "DVFS give-back"); the in-kernel shader clock is
// read from s_memtime / s_memrealtime.  Build: hipcc -w --offload-arch=gfx950 -O3 tools/mfma_f64_sustained.hip -o tools/mfma_f64_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define NACC 15
__global__ __launch_bounds__(512) void k(double* out, const double* in, int iters, long long* stamp) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a[5];
  for (int i = 0; i < 5; ++i) a[i] = in[(blockIdx.x * blockDim.x + threadIdx.x) * 5 + i];
  const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    int idx = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = i; j < 5; ++j) { asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[idx]) : "v"(a[i]), "v"(a[j])); ++idx; }
  }
  const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamp[blockIdx.x * 2] = t1 - t0; stamp[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main(int argc, char** argv) {
  const int blocks = 256;
  double *d, *in; long long* st;
  (void)hipMalloc(&d, blocks * 512 * 8); (void)hipMalloc(&in, blocks * 512 * 5 * 8); (void)hipMalloc(&st, blocks * 16);
  double* h = (double*)malloc(blocks * 512 * 5 * 8);
  srand(1);
  for (int i = 0; i < blocks * 512 * 5; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5) * 1e-3;     // small: no overflow over 1e6 accumulations
  (void)hipMemcpy(in, h, blocks * 512 * 5 * 8, hipMemcpyHostToDevice);
  for (int thr : {256, 512})
    for (int iters : {2000, 20000, 60000}) {
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(thr), 0, 0, d, in, iters, st);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(thr), 0, 0, d, in, iters, st);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      long long hs[512]; (void)hipMemcpy(hs, st, blocks * 16, hipMemcpyDeviceToHost);
      double clk = 0; for (int b = 0; b < blocks; ++b) clk += (double)hs[2 * b] / (double)hs[2 * b + 1] * 100.0; clk /= blocks;
      const double flop = (double)blocks * (thr / 64) * iters * NACC * 2048.0;
      printf("%d wave(s)/SIMD  iters %6d  %7.2f ms  %6.2f TFLOP/s  shader clock %.0f MHz  -> %.1f cycles per MFMA per SIMD\n", thr / 256, iters, ms,
             flop / ms / 1e9, clk, ms * 1e-3 * clk * 1e6 / ((double)(thr / 256) * iters * NACC));
    }
  return 0;
}
