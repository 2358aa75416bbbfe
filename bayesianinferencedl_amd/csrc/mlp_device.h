// Device pieces of the learned error model shared by mlp_kernels.hip (one workgroup of 1024 threads per sample) and
// rom_onesample.hip (the network's forward pass as a spare 256-thread workgroup of the reduced model's contraction kernel):
// see mlp_kernels.hip for the model.
#pragma once
#include "finrom_internal.h"

namespace finrom {

__device__ __forceinline__ float elu_f(float z) { return z > 0.f ? z : expm1f(z); }
__device__ __forceinline__ float elu_grad_f(float z) { return z > 0.f ? 1.f : expf(z); }
// LDS hand-over between the lanes of ONE wave: LDS executes a wave's instructions in order, so all this has to stop is the compiler
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int MLP_MAX_W = 64;        // hidden width (reference: 50 / 100 -> 64 covers load_bn_model's models; checked at create)

// e_out[s][o] (double) = network output; data_shift[s][o] = data[o] - e_out (what the ROM adjoint is run against)
// (Sop, P, theta_out: when given, the sub-fin averages theta = S k of the same field (fom/forward_solve.py:466-480) are formed here
// in fp64 while k is being read anyway -- one launch less in the one-sample call chain of finrom_romml_grad)
// (NTHR threads of ONE workgroup, sample s; xs: n_in floats of LDS.  NTHR = 1024 in mlp_forward_kernel; 256 where the network rides in
// a spare workgroup of the reduced model's contraction kernel, rom_onesample.hip.)
template <int NTHR>
__device__ __forceinline__ void mlp_forward_body(const MlpDev& m, const double* __restrict__ k, int64_t s,
                                                 const double* __restrict__ data, int64_t data_stride,
                                                 float* __restrict__ tape, double* __restrict__ e_out,
                                                 double* __restrict__ data_shift, const double* __restrict__ Sop, int P,
                                                 double* __restrict__ theta_out, float* __restrict__ xs, int tid,
                                                 const double* __restrict__ mom = nullptr, double eps = 0.0,
                                                 double* __restrict__ k_out = nullptr) {
  constexpr int MLP_PARTS = NTHR / 64;                 // threads per hidden unit in the first layer
  constexpr int MLP_THREADS = NTHR;
  constexpr int CH = NTHR >= 1024 ? 32 : 128;          // first-layer loads in flight per thread (fewer threads: deeper batches)
  __shared__ float part[MLP_PARTS][MLP_MAX_W];
  __shared__ float y[MLP_MAX_W], a[MLP_MAX_W];
  __shared__ double tred[MLP_PARTS > 16 ? MLP_PARTS : 16];      // (P <= 16: finrom_romml_grad)
  const int nw = m.n_w;
  if (mom != nullptr) {                                // the leapfrog's position update in front: k + eps * mom, written to k_out
    for (int i = tid; i < m.n_in; i += MLP_THREADS) {
      const double kd = fma(eps, mom[s * m.n_in + i], k[s * m.n_in + i]);
      xs[i] = (float)kd;
      if (k_out != nullptr) k_out[s * m.n_in + i] = kd;
    }
  } else {
    for (int i = tid; i < m.n_in; i += MLP_THREADS) xs[i] = (float)k[s * m.n_in + i];
  }
  if (theta_out != nullptr) {
    // theta_p = sum_i Sop[p][i] k[i] in fp64: row p belongs to the waves p WPR .. p WPR + WPR - 1 (WPR = 16 / P: three waves per row
    // for five averages), a wave's slice dealt over its lanes in coalesced passes of 64, 13 passes requested at a time; ONE
    // value per wave to reduce across lanes.  (The first version had every thread carry all 16 partial sums through the input
    // loop and every wave reduce all of them: 16 x 6 shuffles of doubles per wave, 11.8 of the kernel's 25 us; now 4.6.)
    // (P > MLP_PARTS, e.g. nine averages on a 256-thread workgroup: one wave per row, the rows dealt over the waves in turns)
    const int wave = tid >> 6, lane = tid & 63, WPR = MLP_PARTS / P > 0 ? MLP_PARTS / P : 1, NG = MLP_PARTS / WPR;
    const int p0 = wave / WPR, part_ = wave - p0 * WPR;
    for (int p = p0; p < P && p0 < NG; p += NG) {
      const int i0 = (int)((int64_t)m.n_in * part_ / WPR), i1 = (int)((int64_t)m.n_in * (part_ + 1) / WPR);
      const double* __restrict__ srow = Sop + (int64_t)p * m.n_in;
      const double* __restrict__ krow = k + s * m.n_in;
      double v = 0.0, v2 = 0.0;
      for (int i = i0 + lane; i < i1; i += 64 * 13) {
        double sv[13], kv[13];
#pragma unroll
        for (int u = 0; u < 13; ++u) { const int ii = i + 64 * u; const bool ok = ii < i1; sv[u] = ok ? srow[ii] : 0.0; kv[u] = ok ? krow[ii] : 0.0; }
#pragma unroll
        for (int u = 0; u < 13; ++u) { if (u & 1) v2 = fma(sv[u], kv[u], v2); else v = fma(sv[u], kv[u], v); }
      }
      v += v2;
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) tred[p * WPR + part_] = v;
    }
  }
  __syncthreads();
  if (theta_out != nullptr && tid < P) {               // the row's waves in a fixed order
    const int WPR = MLP_PARTS / P > 0 ? MLP_PARTS / P : 1;
    double t = 0.0;
    for (int wv = 0; wv < WPR; ++wv) t += tred[tid * WPR + wv];
    theta_out[s * P + tid] = t;
  }
  {                                                   // y0 = W0^T x + b0: unit j = tid % 64, part p = tid / 64 of the rows
    const int j = tid & 63, p = tid >> 6;
    float acc = 0.f;
    if (j < nw) {
      const int i0 = (int)((int64_t)m.n_in * p / MLP_PARTS), i1 = (int)((int64_t)m.n_in * (p + 1) / MLP_PARTS);
      float a4[4] = {0.f, 0.f, 0.f, 0.f};              // four independent chains, CH loads in flight
      int i = i0;
      for (; i + CH <= i1; i += CH) {
        float wv[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) wv[u] = m.W0[(int64_t)(i + u) * nw + j];
#pragma unroll
        for (int u = 0; u < CH; ++u) a4[u & 3] = fmaf(xs[i + u], wv[u], a4[u & 3]);
      }
      if constexpr (CH > 16) {                         // (the tail in one clamped batch of 16 where the batches are deep)
        for (; i + 16 <= i1; i += 16) {
          float wv[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) wv[u] = m.W0[(int64_t)(i + u) * nw + j];
#pragma unroll
          for (int u = 0; u < 16; ++u) a4[u & 3] = fmaf(xs[i + u], wv[u], a4[u & 3]);
        }
      }
      for (; i < i1; ++i) a4[0] = fmaf(xs[i], m.W0[(int64_t)i * nw + j], a4[0]);
      acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    }
    part[p][j] = acc;
  }
  __syncthreads();
  if (tid < nw) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < MLP_PARTS; ++w) t += part[w][tid];
    y[tid] = m.b0[tid] + t;
  }
  __syncthreads();
  // the layers behind the first are 50 threads' work -- wave 0's: the other fifteen waves leave, and what were workgroup barriers
  // between the layers (sixteen waves to collect, twice per layer) are the wave's own program order
  if (tid >= 64) return;
  float* tp = tape + s * (int64_t)(m.n_layers + 1) * nw;
  for (int l = 0; l <= m.n_layers; ++l) {              // l == n_layers: the head
    const float* sc = m.scale + l * nw; const float* sh = m.shift + l * nw;
    if (tid < nw) { const float z = fmaf(y[tid], sc[tid], sh[tid]); tp[l * nw + tid] = z; a[tid] = elu_f(z); }
    wave_sync();
    if (l < m.n_layers) {
      const float* W = m.W + (int64_t)l * nw * nw;
      if (tid < nw) {                                 // (weights requested 16 at a time: every load waited for on its own is a
        float acc = m.b[l * nw + tid];                 //  trip to L2 on the critical path of a one-sample call)
        int i = 0;
        for (; i + 16 <= nw; i += 16) {
          float wv[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) wv[u] = W[(i + u) * nw + tid];
#pragma unroll
          for (int u = 0; u < 16; ++u) acc = fmaf(a[i + u], wv[u], acc);
        }
        for (; i < nw; ++i) acc = fmaf(a[i], W[i * nw + tid], acc);
        y[tid] += acc;
      }
    } else if (tid < m.n_out) {
      float acc = m.bh[tid];
      for (int i = 0; i < nw; ++i) acc = fmaf(a[i], m.Wh[i * m.n_out + tid], acc);
      e_out[s * m.n_out + tid] = (double)acc;
      if (data_shift != nullptr) data_shift[s * m.n_out + tid] = data[(data_stride ? s * data_stride : 0) + tid] - (double)acc;
    }
    wave_sync();
  }
}

// Backward pass through the head and the hidden layers by ONE wave (tid < 64 of the caller): on return g[0 .. n_w) in LDS holds
// g0 = d(1/2 |r|^2) / d(y0) without the final minus sign, r = data - (qoi_r + e_NN).  up: 64 floats of LDS scratch.
// (mlp_backward_kernel runs it in its wave 0; in finrom_romml_grad's one-sample form it rides as one more workgroup of the ROM's
// gradient contraction kernel -- it needs the residual, not the contraction -- and the backward kernel starts from its result.)
__device__ __forceinline__ void mlp_backward_hidden_wave(const MlpDev& m, int64_t s, const float* __restrict__ tape,
                                                         const double* __restrict__ data, int64_t data_stride,
                                                         const double* __restrict__ qoi_r, const double* __restrict__ e_nn,
                                                         float* __restrict__ g, float* __restrict__ up, int tid) {
  const int nw = m.n_w;
  const float* tp = tape + s * (int64_t)(m.n_layers + 1) * nw;
  if (tid < m.n_out) {
    const double r = data[(data_stride ? s * data_stride : 0) + tid] - (qoi_r[s * m.n_out + tid] + e_nn[s * m.n_out + tid]);
    up[tid] = (float)r;                                // dLoss/d(output) handed to vjp is +r; the minus sign comes at the end
  }
  wave_sync();
  if (tid < nw) {                                      // through the head: no skip connection
    float acc = 0.f;
    for (int o = 0; o < m.n_out; ++o) acc = fmaf(up[o], m.Wh[tid * m.n_out + o], acc);
    const float z = tp[m.n_layers * nw + tid];
    g[tid] = acc * elu_grad_f(z) * m.scale[m.n_layers * nw + tid];
  }
  wave_sync();
  for (int l = m.n_layers - 1; l >= 0; --l) {          // g <- g + (W_l g) * elu'(z_l) * s_l   (skip + branch)
    const float* W = m.W + (int64_t)l * nw * nw;
    float gnew = 0.f;
    if (tid < nw) {
      float acc = 0.f;
      int j = 0;
      for (; j + 16 <= nw; j += 16) {
        float wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = W[tid * nw + j + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = fmaf(g[j + u], wv[u], acc);
      }
      for (; j < nw; ++j) acc = fmaf(g[j], W[tid * nw + j], acc);
      gnew = g[tid] + acc * elu_grad_f(tp[l * nw + tid]) * m.scale[l * nw + tid];
    }
    wave_sync();                                       // everybody has read the old g
    if (tid < nw) g[tid] = gnew;
    wave_sync();
  }
}

}  // namespace finrom
