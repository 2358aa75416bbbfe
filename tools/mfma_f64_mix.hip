// What does one extra instruction between two v_mfma_f64_16x16x4_f64 cost the issuing wave / the pipe?
// 15 independent accumulators (as in the r = 80 projection loop); after every MFMA, K instructions of one kind.
// Build: hipcc -w --offload-arch=gfx950 -O3 tools/mfma_f64_mix.hip -o tools/mfma_f64_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define NACC 15
// KIND: 0 none, 1 v_fma_f64, 2 v_mul_f64, 3 v_fma_f32, 4 v_add_u32, 5 v_lshl_add_u64, 6 ds_read_b64, 7 global_load_dwordx2 (L1/L2 hit), 8 v_add_f64
template <int KIND, int K>
__global__ __launch_bounds__(512) void k(double* out, const double* tab, int iters, long long* cyc, double a0) {
  __shared__ double lds[512 * 4];
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a[5];
  for (int i = 0; i < 5; ++i) a[i] = a0 * (threadIdx.x * 1e-3 + 1.0 + i);
  double x[8]; float xf[8]; unsigned xi[8]; unsigned long long xl[8];
  for (int i = 0; i < 8; ++i) { x[i] = a0 * (i + 1); xf[i] = (float)x[i]; xi[i] = threadIdx.x + i; xl[i] = threadIdx.x + i; }
  lds[threadIdx.x] = a0;
  __syncthreads();
  const double* lp = lds + (threadIdx.x & 63);
  const double* gp = tab + (threadIdx.x & 63);
  const double y = a0 * 0.999;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    int idx = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = i; j < 5; ++j) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[idx]) : "v"(a[i]), "v"(a[j]));
#pragma unroll
        for (int e = 0; e < K; ++e) {
          const int r = (idx * K + e) & 7;
          if (KIND == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[r]) : "v"(y), "v"(y));
          if (KIND == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[r]) : "v"(y));
          if (KIND == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(xf[r]) : "v"(xf[(r + 1) & 7]), "v"(xf[(r + 2) & 7]));
          if (KIND == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(xi[r]) : "v"(xi[(r + 1) & 7]));
          if (KIND == 5) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(xl[r]) : "v"(xl[(r + 1) & 7]));
          if (KIND == 6) asm volatile("ds_read_b64 %0, %1" : "=v"(x[r]) : "v"((unsigned)(size_t)(lp + 64 * (e & 3)) ) : "memory");
          if (KIND == 7) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(x[r]) : "v"(gp + 64 * ((idx + e) & 15)) : "memory");
          if (KIND == 8) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[r]) : "v"(y));
        }
        ++idx;
      }
    if (KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (KIND == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += x[i] + xf[i] + xi[i] + (double)xl[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND, int K>
void run(const char* name, int threads, double* d, double* tab, long long* dc) {
  const int blocks = 256, iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, K>), dim3(blocks), dim3(threads), 0, 0, d, tab, iters, dc, 1.0);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, K>), dim3(blocks), dim3(threads), 0, 0, d, tab, iters, dc, 1.0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  double flop = (double)blocks * (threads / 64) * iters * NACC * 2048.0;
  printf("%-22s K=%d  %d waves/SIMD  %6.2f TFLOP/s  %6.1f ticks/MFMA/wave\n", name, K, threads / 256, flop / ms / 1e9,
         (double)c / ((double)iters * NACC));
}
int main() {
  double *d, *tab; long long* dc;
  (void)hipMalloc(&d, 4096 * 512 * 8); (void)hipMalloc(&tab, 1 << 20); (void)hipMemset(tab, 0, 1 << 20); (void)hipMalloc(&dc, 16);
  for (int thr : {256, 512}) {
    run<0, 0>("none", thr, d, tab, dc);
    run<1, 1>("v_fma_f64", thr, d, tab, dc); run<1, 2>("v_fma_f64", thr, d, tab, dc); run<1, 4>("v_fma_f64", thr, d, tab, dc);
    run<2, 1>("v_mul_f64", thr, d, tab, dc); run<8, 1>("v_add_f64", thr, d, tab, dc);
    run<3, 1>("v_fma_f32", thr, d, tab, dc); run<3, 4>("v_fma_f32", thr, d, tab, dc);
    run<4, 2>("v_add_u32", thr, d, tab, dc); run<5, 2>("v_lshl_add_u64", thr, d, tab, dc);
    run<6, 1>("ds_read_b64", thr, d, tab, dc); run<7, 1>("global_load_dwordx2", thr, d, tab, dc); run<7, 2>("global_load_dwordx2", thr, d, tab, dc);
  }
  return 0;
}
