// What makes a kernel node of a replayed HIP graph expensive?  Chains of N launches of a dummy kernel, one property varied at a
// time: dynamic LDS bytes, threads per workgroup, workgroups, register footprint, kernel-argument bytes.  Prints microseconds per
// node (graph replay, HIP events).      hipcc --offload-arch=gfx950 -O2 tools/graph_node_cost.hip -o /tmp/gnc && /tmp/gnc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Big { double pad[60]; };      // 480 bytes of kernel arguments

__global__ void k_small(double* out) { extern __shared__ double lds[]; if (threadIdx.x == 0 && blockIdx.x == 0) { lds[0] = out[0]; out[1] = lds[0] + 1.0; } }
__global__ void k_bigarg(double* out, Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = out[0] + b.pad[59]; }
__global__ __launch_bounds__(256) void k_regs(double* out) {      // ~200 VGPRs + AGPRs live
  double v[96];
#pragma unroll
  for (int i = 0; i < 96; ++i) v[i] = out[i & 7] * (double)(i + threadIdx.x);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 96; ++i) v[i] = fma(v[(i + 1) % 96], v[(i + 7) % 96], v[i]);
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 96; ++i) s += v[i];
  if (s == 12345.678) out[9] = s;
}
__global__ void k_work(double* out, int iters) {      // a dependent chain: the kernel takes ~iters x 20 ns
  double s = out[0];
  for (int i = 0; i < iters; ++i) s = fma(s, 1.0000001, 1e-9);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[2] = s;
}

template <class F> static int run(const char* name, int N, hipStream_t st, F&& launch) {
  hipGraph_t g; hipGraphExec_t ge;
  for (int i = 0; i < 4; ++i) launch();
  CK(hipStreamSynchronize(st));
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < N; ++i) launch();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-58s %7.2f us per node\n", name, ms * 1e3 / 20 / N);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 0;
}

int main() {
  const int N = 200;
  hipStream_t st; CK(hipStreamCreate(&st));
  double* out; CK(hipMalloc(&out, 4096)); CK(hipMemset(out, 0, 4096));
  CK(hipFuncSetAttribute((const void*)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  char name[128];
  for (int lds : {0, 8 << 10, 32 << 10, 64 << 10, 128 << 10, 160 << 10}) {
    snprintf(name, sizeof name, "1 workgroup x 64 threads, %3d KB dynamic LDS", lds >> 10);
    if (run(name, N, st, [&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), lds, st, out); })) return 1;
  }
  for (int thr : {256, 1024}) {
    snprintf(name, sizeof name, "1 workgroup x %d threads, no LDS", thr);
    if (run(name, N, st, [&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(thr), 0, st, out); })) return 1;
  }
  for (int wg : {8, 32, 256, 2048}) {
    snprintf(name, sizeof name, "%d workgroups x 256 threads, no LDS", wg);
    if (run(name, N, st, [&] { hipLaunchKernelGGL(k_small, dim3(wg), dim3(256), 0, st, out); })) return 1;
  }
  for (int wg : {4, 32}) {
    snprintf(name, sizeof name, "%d workgroups x 256 threads, 126 KB dynamic LDS", wg);
    if (run(name, N, st, [&] { hipLaunchKernelGGL(k_small, dim3(wg), dim3(256), 126 << 10, st, out); })) return 1;
  }
  if (run("1 workgroup x 64 threads, 480 B of arguments", N, st, [&] { Big b{}; hipLaunchKernelGGL(k_bigarg, dim3(1), dim3(64), 0, st, out, b); })) return 1;
  if (run("4 workgroups x 256 threads, ~200 registers", N, st, [&] { hipLaunchKernelGGL(k_regs, dim3(4), dim3(256), 0, st, out); })) return 1;
  for (int it : {250, 1000, 2500}) {
    snprintf(name, sizeof name, "4 workgroups x 256 threads, chain of %d fma", it);
    if (run(name, N, st, [&] { hipLaunchKernelGGL(k_work, dim3(4), dim3(256), 0, st, out, it); })) return 1;
  }
  // alternating: a long kernel followed by a tiny one (does the long one's END cost more?)
  if (run("alternating: chain of 2500 fma, then a tiny kernel (per pair / 2)", N, st, [&] {
        static int t = 0; if ((t++ & 1) == 0) hipLaunchKernelGGL(k_work, dim3(4), dim3(256), 0, st, out, 2500); else hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, out); })) return 1;
  return 0;
}
