// 15 back-to-back v_mfma_f64_16x16x4_f64 followed by a BURST of other work (the shape of the projection's k-step):
// NV v_fma_f64, NS s_add, NBR taken scalar branches.  2 waves per SIMD on every CU.  Which part of the burst costs MFMA rate?
// Build: hipcc -w --offload-arch=gfx950 -O3 tools/mfma_f64_burst.hip -o tools/mfma_f64_burst
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define NACC 15
template <int NV, int NS, int NBR, int PRIO, int THRB>
__global__ __launch_bounds__(THRB) void k(double* out, int iters, double a0) {
  if (PRIO == 1 && threadIdx.x >= 256) __builtin_amdgcn_s_setprio(1);      // the second wave of every SIMD
  if (PRIO == 2 && threadIdx.x >= 256) __builtin_amdgcn_s_setprio(3);
  if (PRIO == 3 && threadIdx.x >= 256) { __builtin_amdgcn_s_sleep(10); }      // the second wave of every SIMD starts ~640 cycles late
  if (PRIO == 4 && threadIdx.x >= 256) { for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(127); }
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a[5], x[8];
  for (int i = 0; i < 5; ++i) a[i] = a0 * (threadIdx.x * 1e-3 + 1.0 + i);
  for (int i = 0; i < 8; ++i) x[i] = a0 * (i + 1);
  const double y = a0 * 0.999;
  int sc = iters;
  for (int it = 0; it < iters; ++it) {
    int idx = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = i; j < 5; ++j) { asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[idx]) : "v"(a[i]), "v"(a[j])); ++idx; }
#pragma unroll
    for (int e = 0; e < NV; ++e) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[e & 7]) : "v"(y), "v"(y));
#pragma unroll
    for (int e = 0; e < NS; ++e) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sc));
#pragma unroll
    for (int e = 0; e < NBR; ++e) asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" : : "s"(sc) : "scc");
  }
  double s = sc;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int NS, int NBR, int PRIO = 0, int THR = 512>
void run(double* d) {
  const int blocks = 256, thr = THR, iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, NS, NBR, PRIO, THR>), dim3(blocks), dim3(thr), 0, 0, d, iters, 1.0);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NS, NBR, PRIO, THR>), dim3(blocks), dim3(thr), 0, 0, d, iters, 1.0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * (thr / 64) * iters * NACC * 2048.0;
  printf("burst per 15 MFMAs: %2d v_fma_f64 %3d s_add %2d branches  prio %d  %d wave(s)/SIMD  %6.2f TFLOP/s\n", NV, NS, NBR, PRIO, thr / 256, flop / ms / 1e9);
}
int main() {
  double* d; (void)hipMalloc(&d, 4096 * 512 * 8);
  run<0, 0, 0>(d); run<7, 0, 0>(d); run<15, 0, 0>(d); run<30, 0, 0>(d);
  run<0, 40, 0>(d); run<0, 100, 0>(d); run<0, 0, 8>(d); run<0, 0, 20>(d);
  run<7, 40, 8>(d); run<15, 40, 8>(d); run<30, 100, 20>(d);
  run<7, 40, 8, 0, 256>(d); run<7, 40, 8, 0, 768>(d); run<7, 40, 8, 3>(d); run<7, 40, 8, 4>(d);
  run<7, 40, 8, 1>(d); run<7, 40, 8, 2>(d); run<15, 40, 8, 1>(d); run<30, 100, 20, 1>(d); run<0, 0, 0, 1>(d);
  return 0;
}
