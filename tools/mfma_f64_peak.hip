// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on gfx950 (the guide has no fp64 row).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o tools/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, long long* cyc, double a0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 * (threadIdx.x * 1e-3 + 1.0), b = a0 * (1.0 - threadIdx.x * 1e-4);
  long long r0 = __builtin_amdgcn_s_memrealtime();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void kfma(double* out, int iters) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  double a = threadIdx.x * 1e-9 + 1.0, b = 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks, int iters, double* d, long long* dc, double a0 = 1.0) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, dc, a0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, dc, a0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long cc[2]; hipMemcpy(cc, dc, 16, hipMemcpyDeviceToHost); long long c = cc[0];
  double flop = (double)blocks * 4 * iters * NACC * 2048.0;
  printf("mfma_f64_16x16x4: NACC=%2d blocks=%5d a0=%g  %.2f TFLOP/s  %.1f memtime ticks/MFMA/wave  clock=%.0f MHz\n", NACC, blocks, a0,
         flop / ms / 1e9, (double)c / ((double)iters * NACC), (double)cc[0] / (double)cc[1] * 100.0);
}

int main() {
  double* d; long long* dc;
  hipMalloc(&d, 4096 * 256 * 8); hipMalloc(&dc, 16);
  run<1>(256, 20000, d, dc);
  run<4>(256, 5000, d, dc);
  run<15>(256, 2000, d, dc);
  run<15>(512, 2000, d, dc);
  run<15>(1024, 2000, d, dc);
  run<15>(1024, 20000, d, dc);
  run<15>(1024, 20000, d, dc, 0.0);
  run<15>(256, 20000, d, dc, 0.0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 1024, 2048}) {
    int iters = 20000;
    hipLaunchKernelGGL(kfma, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kfma, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("v_fma_f64: blocks=%d  %.2f TFLOP/s\n", blocks, (double)blocks * 256 * iters * 16 * 2.0 / ms / 1e9);
  }
  return 0;
}
