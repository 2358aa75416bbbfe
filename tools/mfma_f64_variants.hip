// Which register placement lets v_mfma_f64_16x16x4_f64 issue at its 64-cycle pipe occupancy?
// Build: hipcc -w --offload-arch=gfx950 -O3 tools/mfma_f64_variants.hip -o tools/mfma_f64_variants
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define NACC 15
template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, int iters, long long* cyc, double a0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a[5], b[5];
  for (int i = 0; i < 5; ++i) { a[i] = a0 * (threadIdx.x * 1e-3 + 1.0 + i); b[i] = a0 * (1.0 - threadIdx.x * 1e-4 * (i + 1)); }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // compiler's register choice, distinct operands
      int idx = 0;
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = i; j < 5; ++j) { acc[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], a[j], acc[idx], 0, 0, 0); ++idx; }
    } else if (MODE == 1) {   // AGPR accumulators forced through inline asm
      int idx = 0;
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = i; j < 5; ++j) { asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[idx]) : "v"(a[i]), "v"(a[j])); ++idx; }
    } else {                  // VGPR accumulators forced
      int idx = 0;
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = i; j < 5; ++j) { asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[idx]) : "v"(a[i]), "v"(a[j])); ++idx; }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + b[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(const char* name, int blocks, int threads, int iters, double* d, long long* dc) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, dc, 1.0);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, dc, 1.0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  double flop = (double)blocks * (threads / 64) * iters * NACC * 2048.0;
  printf("%-28s blocks=%4d x %3d thr  %.2f TFLOP/s  %.1f ticks/MFMA/wave\n", name, blocks, threads, flop / ms / 1e9, (double)c / ((double)iters * NACC));
}
int main() {
  double* d; long long* dc;
  (void)hipMalloc(&d, 4096 * 512 * 8); (void)hipMalloc(&dc, 16);
  for (int thr : {256, 512}) {
    run<0>("builtin (compiler regs)", 256, thr, 4000, d, dc);
    run<1>("asm AGPR acc", 256, thr, 4000, d, dc);
    run<2>("asm VGPR acc", 256, thr, 4000, d, dc);
  }
  return 0;
}
