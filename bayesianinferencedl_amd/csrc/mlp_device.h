// Device pieces of the learned error model shared by mlp_kernels.hip (one workgroup of 1024 threads per sample) and
// rom_onesample.hip (the network's forward pass as a spare 256-thread workgroup of the reduced model's contraction kernel):
// see mlp_kernels.hip for the model.
#pragma once
#include "finrom_internal.h"

namespace finrom {

__device__ __forceinline__ float elu_f(float z) { return z > 0.f ? z : expm1f(z); }
__device__ __forceinline__ float elu_grad_f(float z) { return z > 0.f ? 1.f : expf(z); }
// LDS hand-over between the lanes of ONE wave: LDS executes a wave's instructions in order, so all this has to stop is the compiler
// (NOT a release / acquire fence pair, even at wavefront scope: hipcc turns the release into `s_waitcnt vmcnt(0)`, i.e. every
//  hand-over waited for the acknowledgement of the tape's global STORES -- 2 us from a cold line, twelve times per walk through the
//  layers: 29 of the forward walk's 30 us, found with phase clocks in round 4.  LDS keeps a wave's accesses in order; what is
//  needed is that the compiler does not move memory operations across the hand-over and that outstanding LDS operations are waited for.)
__device__ __forceinline__ void wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
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

// ---- the forward pass in two pieces (finrom_romml_grad's one-sample form, round 4) -----------------------------------------------
// As ONE spare 256-thread workgroup of the contraction kernel the forward pass took as long as the contraction itself (25 us: 400
// rows of the first layer per thread).  Now NW0 spare workgroups each take n_in / NW0 rows of the first layer and leave their
// partial sums y0_part[s][w][64]; the layers behind (50 threads' work) are a spare WAVE of the solve kernel, which is the first
// to need their result (the adjoint's right-hand side, half a kernel later).
// (xs: (r1 - r0) floats of LDS; part: [NTHR / 64][64] floats of LDS)
template <int NTHR>
__device__ __forceinline__ void mlp_first_layer_part(const MlpDev& m, const double* __restrict__ k, int64_t s, const double* __restrict__ mom,
                                                     double eps, double* __restrict__ k_out, int w, int nw0, float* __restrict__ y0_part,
                                                     float* __restrict__ xs, int tid) {
  constexpr int PARTS = NTHR / 64, CH = 32;
  __shared__ float part[PARTS][MLP_MAX_W];
  const int nw = m.n_w;
  const int r0 = (int)((int64_t)m.n_in * w / nw0), r1 = (int)((int64_t)m.n_in * (w + 1) / nw0);
  for (int i = r0 + tid; i < r1; i += NTHR) {            // the leapfrog's position update in front: k + eps * mom, written to k_out
    const int64_t idx = s * m.n_in + i;
    const double kd = mom != nullptr ? fma(eps, mom[idx], k[idx]) : k[idx];
    xs[i - r0] = (float)kd;
    if (k_out != nullptr) k_out[idx] = kd;
  }
  __syncthreads();
  {
    const int j = tid & 63, p = tid >> 6;
    float acc = 0.f;
    if (j < nw) {
      const int i0 = r0 + (int)((int64_t)(r1 - r0) * p / PARTS), i1 = r0 + (int)((int64_t)(r1 - r0) * (p + 1) / PARTS);
      float a4[4] = {0.f, 0.f, 0.f, 0.f};                // four independent chains, CH loads in flight
      int i = i0;
      for (; i + CH <= i1; i += CH) {
        float wv[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) wv[u] = m.W0[(int64_t)(i + u) * nw + j];
#pragma unroll
        for (int u = 0; u < CH; ++u) a4[u & 3] = fmaf(xs[i + u - r0], wv[u], a4[u & 3]);
      }
      for (; i < i1; ++i) a4[0] = fmaf(xs[i - r0], m.W0[(int64_t)i * nw + j], a4[0]);
      acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    }
    part[p][j] = acc;
  }
  __syncthreads();
  if (tid < 64) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < PARTS; ++q) t += part[q][tid];
    y0_part[tid] = tid < nw ? t : 0.f;
  }
}

// the layers behind the first, by ONE wave (lane = unit): y0 = b0 + the nw0 partial sums in index order; tape, e_out, data_shift as
// mlp_forward_body; dsh_lds (optional): data - e_NN for the caller's own later use.  y, a: 64 floats of LDS each.
// STAGED: wl = the hidden layers' and the head's weights in LDS -- [n_layers][n_w][n_w] then [n_w][n_out], the rest of the 64 x 64
// floats a layer's loops may touch FINITE (the caller zero-fills) -- so that the loops run unguarded over 64 inputs against a[] = 0
// beyond n_w.  (Found with phase clocks, round 4: the first version selected between the LDS copy and the global arrays with one
// pointer -- a FLAT pointer: every weight a flat load under its own exec-mask branch, the walk 29 us whatever was staged.)
template <bool STAGED>
__device__ __forceinline__ void mlp_forward_tail_wave(const MlpDev& m, int64_t s, const float* __restrict__ y0_part, int nw0,
                                                      const double* __restrict__ data, int64_t data_stride, float* __restrict__ tape,
                                                      double* __restrict__ e_out, double* __restrict__ data_shift,
                                                      double* __restrict__ dsh_lds, float* __restrict__ y, float* __restrict__ a, int tid,
                                                      const float* __restrict__ wl) {
  const int nw = m.n_w;
  // this unit's batch-norm scales / shifts and biases of EVERY layer, requested at once (three dependent trips per layer otherwise)
  constexpr int LMAX = 8;
  float scv[LMAX + 1], shv[LMAX + 1], bv[LMAX + 1];
  const bool pre = m.n_layers <= LMAX;
  if (pre) {
#pragma unroll
    for (int l = 0; l <= LMAX; ++l) {
      const bool ok = l <= m.n_layers && tid < nw;
      scv[l] = ok ? m.scale[l * nw + tid] : 0.f; shv[l] = ok ? m.shift[l * nw + tid] : 0.f;
      bv[l] = l < m.n_layers ? (tid < nw ? m.b[l * nw + tid] : 0.f) : (l == m.n_layers && tid < m.n_out ? m.bh[tid] : 0.f);
    }
  }
  {
    float pv[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) pv[w] = (w < nw0 && tid < nw) ? y0_part[w * 64 + tid] : 0.f;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += pv[w];            // (index order; slots beyond nw0 add exact zeros)
    for (int w = 8; w < nw0; ++w) t += tid < nw ? y0_part[w * 64 + tid] : 0.f;
    y[tid] = tid < nw ? m.b0[tid] + t : 0.f;
    a[tid] = 0.f;                                      // (a[n_w .. 63] stay zero: the staged loops run over 64 inputs)
  }
  wave_sync();
  float* tp = tape + s * (int64_t)(m.n_layers + 1) * nw;
#pragma unroll 1
  for (int l = 0; l <= m.n_layers; ++l) {              // l == n_layers: the head
    float scl = 0.f, shl = 0.f, bl_ = 0.f;
    if (pre) {
#pragma unroll
      for (int q = 0; q <= LMAX; ++q) if (q == l) { scl = scv[q]; shl = shv[q]; bl_ = bv[q]; }
    } else if (tid < nw) {
      scl = m.scale[l * nw + tid]; shl = m.shift[l * nw + tid];
      bl_ = l < m.n_layers ? m.b[l * nw + tid] : (tid < m.n_out ? m.bh[tid] : 0.f);
    }
    if (tid < nw) { const float z = fmaf(y[tid], scl, shl); tp[l * nw + tid] = z; a[tid] = elu_f(z); }
    wave_sync();
    const bool head = l == m.n_layers;
    const int ld = head ? m.n_out : nw;                // row length of this layer's matrix
    float acc = bl_;
    if constexpr (STAGED) {
      const float* W = wl + (int64_t)l * nw * nw + tid;      // (l == n_layers: the head follows the hidden layers in the staged array)
      // (four independent chains of 16 instead of one of 64: the walk is a lone wave's dependent arithmetic; a[] four at a time)
      typedef float f4_t __attribute__((ext_vector_type(4)));
      float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i0 = 0; i0 < MLP_MAX_W; i0 += 16) {
        float wv[16]; f4_t av[4];
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = W[(i0 + u) * ld];
#pragma unroll
        for (int u = 0; u < 4; ++u) av[u] = ((const f4_t*)a)[i0 / 4 + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) a4[u & 3] = fmaf(av[u >> 2][u & 3], wv[u], a4[u & 3]);
      }
      acc += (a4[0] + a4[1]) + (a4[2] + a4[3]);
    } else {
      const float* W = head ? m.Wh + tid : m.W + (int64_t)l * nw * nw + tid;
      if (tid < ld) {
        int i = 0;
        for (; i + 16 <= nw; i += 16) {
          float wv[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) wv[u] = W[(i + u) * ld];
#pragma unroll
          for (int u = 0; u < 16; ++u) acc = fmaf(a[i + u], wv[u], acc);
        }
        for (; i < nw; ++i) acc = fmaf(a[i], W[i * ld], acc);
      }
    }
    if (!head) {
      if (tid < nw) y[tid] += acc;
    } else if (tid < m.n_out) {
      e_out[s * m.n_out + tid] = (double)acc;
      const double dsh = data[(data_stride ? s * data_stride : 0) + tid] - (double)acc;
      if (data_shift != nullptr) data_shift[s * m.n_out + tid] = dsh;
      if (dsh_lds != nullptr) dsh_lds[tid] = dsh;
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
