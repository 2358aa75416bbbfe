// Semantics check of __builtin_amdgcn_global_load_lds (16 B per lane): wave-uniform LDS base + lane*16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k(const double* __restrict__ src, double* __restrict__ dst, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int off = wave * 1024; off < nbytes; off += nw * 1024) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)src + off + lane * 16),
                                     (__attribute__((address_space(3))) void*)(lds + off), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const double* l = (const double*)lds;
  for (int i = threadIdx.x; i < nbytes / 8; i += blockDim.x) dst[i] = l[i] * 2.0;
}
int main() {
  const int nbytes = 32 * 1024, n = nbytes / 8;
  std::vector<double> h(n), o(n);
  for (int i = 0; i < n; ++i) h[i] = i + 0.5;
  double *s, *d;
  (void)hipMalloc(&s, nbytes); (void)hipMalloc(&d, nbytes);
  (void)hipMemcpy(s, h.data(), nbytes, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(512), nbytes, 0, s, d, nbytes);
  (void)hipMemcpy(o.data(), d, nbytes, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) if (o[i] != 2.0 * h[i]) { if (bad < 5) printf("mismatch %d: %g vs %g\n", i, o[i], 2 * h[i]); ++bad; }
  printf("glds test: %d mismatches of %d (%s)\n", bad, n, hipGetErrorString(hipGetLastError()));
  return bad != 0;
}
