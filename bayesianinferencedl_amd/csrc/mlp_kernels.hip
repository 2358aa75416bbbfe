// The learned error model on the device: the reference evaluates a Keras residual batch-normalised fully
// connected network e_NN(k): R^n -> R^n_obs (deep_learning/dl_model.py:149-176 `res_bn_fc_model`) and its input gradient
// (tf.gradients(loss, model.input), rom/averaged_affine_ROM.py:225-228, 381-395) once per MAP / HMC step, in fp32.
//     y0 = W0^T x + b0;   y_{i+1} = y_i + W_i^T elu(s_i y_i + t_i) + b_i  (i < L);   out = W_h^T elu(s_h y_L + t_h) + b_h
// (batch normalisation in inference form: s = gamma / sqrt(var + eps), t = beta - mean s; the first BN-activation-Dense triple of
// the reference's residual_unit is dead code, :150-158).
// One workgroup of 1024 threads per sample (the call pattern is S = 1 .. a few chains: latency, not throughput): x in LDS, the
// first layer split over 16 threads per hidden unit, the 50 x 50 layers one thread per output.  Forward keeps the pre-activations
// ("tape", (L + 1) x n_w floats per sample) for the backward kernel, which runs after the ROM adjoint (its upstream is the
// residual data - (qoi_r + e_NN)) and also adds the ROM part g_theta^T S, so that the final gradient is written once.
#include "mlp_device.h"

namespace finrom {

constexpr int MLP_PARTS = 16;        // threads per hidden unit in the first layer (1024 threads = 64 units x 16 parts: the layer is
                                     // 1597 x 50 dependent-load multiply-adds per sample and the call is latency-bound -- 100 rows
                                     // per thread, 32 loads in flight, instead of 400 rows with 4 parts: 28 -> ~10 us)
constexpr int MLP_THREADS = 64 * MLP_PARTS;
constexpr int MLP_SPLIT = HMC_THETA_PARTS;      // (8) workgroups per sample in the backward kernel's one-sample form ...
constexpr int MLP_SPLIT_MAX_S = 64;  // ... which serves calls of up to this many samples

__global__ __launch_bounds__(MLP_THREADS) void mlp_forward_kernel(MlpDev m, const double* __restrict__ k, int64_t S,
                                                          const double* __restrict__ data, int64_t data_stride,
                                                          float* __restrict__ tape, double* __restrict__ e_out,
                                                          double* __restrict__ data_shift, const double* __restrict__ Sop, int P,
                                                          double* __restrict__ theta_out) {
  extern __shared__ float xs[];                        // [n_in] input
  mlp_forward_body<MLP_THREADS>(m, k, (int64_t)blockIdx.x, data, data_stride, tape, e_out, data_shift, Sop, P, theta_out, xs, (int)threadIdx.x);
}

// grad[s][i] = sum_p g_theta[s][p] Sop[p][i]  -  sum_j g0[j] W0[i][j],  g0 = d(1/2 |r|^2)/d(y0) with upstream r = data - (qoi_r + e)
// loss[s] = 1/2 |r|^2 is recomputed here from the same residual (the ROM kernel's J is that of the shifted data: the same number)
// (NP > 1, calls of up to MLP_SPLIT_MAX_S samples: every one of a sample's NP workgroups walks back through the head and the hidden
// layers -- the same few microseconds, side by side -- and then takes n_in / NP rows of the first layer's transpose: 320 KB of
// weights that otherwise pass through ONE CU's L1)
template <int NP>
__global__ __launch_bounds__(MLP_THREADS) void mlp_backward_kernel(MlpDev m, int64_t S, const float* __restrict__ tape,
                                                           const double* __restrict__ data, int64_t data_stride,
                                                           const double* __restrict__ qoi_r, const double* __restrict__ e_nn,
                                                           const double* __restrict__ g_theta, const double* __restrict__ Sop,
                                                           int P, double* __restrict__ grad, const double* __restrict__ g_parts,
                                                           int n_parts, const float* __restrict__ g0_in, HmcTail ht) {
  __shared__ float g[MLP_MAX_W], up[MLP_MAX_W];
  __shared__ double gth[32];
  const int64_t s = blockIdx.x;
  const int tid = threadIdx.x, nw = m.n_w;
  if (tid < P) {
    double t = 0.0;
    if (g_parts != nullptr) {                          // the contraction's partial sums, in its own fixed order (requested twelve at a time)
      for (int w0 = 0; w0 < n_parts; w0 += 12) {
        double v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) v[u] = w0 + u < n_parts ? g_parts[(s * n_parts + w0 + u) * 32 + tid] : 0.0;
#pragma unroll
        for (int u = 0; u < 12; ++u) t += v[u];
      }
    } else if (g_theta != nullptr) t = g_theta[s * P + tid];
    gth[tid] = t;
  }
  if (tid < 64) {                                      // head and hidden layers are 50 threads' work: wave 0 walks back alone, its
    if (g0_in != nullptr) {                            // hand-overs are its own program order (not barriers of sixteen waves) --
      if (tid < nw) g[tid] = g0_in[s * MLP_MAX_W + tid];      // or somebody did it already (finrom_romml_grad's one-sample form)
    } else {
      mlp_backward_hidden_wave(m, s, tape, data, data_stride, qoi_r, e_nn, g, up, tid);
    }
  }
  __syncthreads();                                     // g is final: the other waves join for the first layer's transpose
  const int wg = NP > 1 ? (int)blockIdx.y : 0;
  const int r0 = (int)((int64_t)m.n_in * wg / NP), r1 = (int)((int64_t)m.n_in * (wg + 1) / NP);
  constexpr int TP = HMC_THETA_MAXP;                   // (the carried averages: at most this many -- registers; 1024 threads = 128 each)
  const bool carry = ht.on && ht.theta_parts != nullptr && NP == HMC_THETA_PARTS;
  double tpart[TP];
#pragma unroll
  for (int p = 0; p < TP; ++p) tpart[p] = 0.0;
  for (int i = r0 + tid; i < r1; i += MLP_THREADS) {
    // everything this row needs from memory is requested up front (one trip): its column of Sop serves the chain rule AND the next
    // step's sub-fin averages; the leapfrog's operands do not wait behind the product with W0
    const int64_t idx = s * m.n_in + i;
    double sv[TP];
    if (carry) {
#pragma unroll
      for (int p = 0; p < TP; ++p) sv[p] = p < P ? Sop[(int64_t)p * m.n_in + i] : 0.0;
    }
    double kq_i = 0.0, mean_i = 0.0, mom_i = 0.0; int bad_s = 0;
    if (ht.on) { kq_i = ht.kq[idx]; mean_i = ht.mean[idx]; mom_i = ht.mom[idx]; bad_s = ht.info[s]; }
    float acc = 0.f;
    {
      const float* __restrict__ wrow = m.W0 + (int64_t)i * nw;
      float a4[4] = {0.f, 0.f, 0.f, 0.f};
      int j = 0;
      for (; j + 16 <= nw; j += 16) {
        float wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = wrow[j + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) a4[u & 3] = fmaf(g[j + u], wv[u], a4[u & 3]);
      }
      for (; j < nw; ++j) a4[0] = fmaf(g[j], wrow[j], a4[0]);
      acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    }
    double out = -(double)acc;                         // d loss / d input = -vjp(r)
    if (carry) {
#pragma unroll
      for (int p = 0; p < TP; ++p)
        if (p < P) out = fma(gth[p], sv[p], out);
    } else {
      for (int p = 0; p < P; ++p) out = fma(gth[p], Sop[(int64_t)p * m.n_in + i], out);
    }
    if (grad != nullptr) grad[idx] = out;
    if (ht.on) {                                       // the leapfrog's momentum update behind the gradient (finrom_hmc_leapfrog)
      const double du = bad_s != 0 ? 0.0 : fma(ht.coef, out, kq_i - mean_i);
      ht.dU[idx] = du;
      const double pn = fma(-ht.eps_cpri, du, mom_i);
      ht.mom[idx] = pn;
      if (carry) {                                     // the next step's field, exactly as its kernels will form it
        const double kn = fma(ht.eps, pn, kq_i);
#pragma unroll
        for (int p = 0; p < TP; ++p)
          if (p < P) tpart[p] = fma(sv[p], kn, tpart[p]);
      }
    }
  }
  if (carry) {                                          // this workgroup's partial sums, waves in index order
    __shared__ double tw[MLP_THREADS / 64][16];
    const int lane = tid & 63, wave = tid >> 6;
    const int nwv = (r1 - r0 + 63) / 64 < MLP_THREADS / 64 ? (r1 - r0 + 63) / 64 : MLP_THREADS / 64;      // waves that hold rows (4 of 16 at n = 1597)
    if (wave < nwv) {
#pragma unroll
      for (int p = 0; p < TP; ++p)
        if (p < P) {
          double x = tpart[p];
          for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
          if (lane == 0) tw[wave][p] = x;
        }
    }
    __syncthreads();
    if (tid < P) {
      double t = 0.0;
      for (int w = 0; w < nwv; ++w) t += tw[w][tid];
      ht.theta_parts[(s * HMC_THETA_PARTS + wg) * 16 + tid] = t;
    }
  }
}

int launch_mlp_forward(const MlpDev& m, const double* k, int64_t S, const double* data, int64_t data_stride, float* tape,
                       double* e_out, double* data_shift, hipStream_t st, const double* Sop, int P, double* theta_out) {
  if (S == 0) return 0;
  if (theta_out != nullptr && (P < 1 || P > 16)) { set_error("mlp_forward: at most 16 sub-fin averages"); return FINROM_ERR_UNSUPPORTED; }
  ScopedKernelTimer t(K_MISC, st);
  hipLaunchKernelGGL(mlp_forward_kernel, dim3((unsigned)S), dim3(MLP_THREADS), (size_t)m.n_in * sizeof(float), st, m, k, S, data, data_stride,
                     tape, e_out, data_shift, Sop, P, theta_out);
  FR_HIP(hipGetLastError());
  return 0;
}

int launch_mlp_backward(const MlpDev& m, int64_t S, const float* tape, const double* data, int64_t data_stride, const double* qoi_r,
                        const double* e_nn, const double* g_theta, const double* Sop, int P, double* grad, hipStream_t st,
                        const double* g_parts, int n_parts, const float* g0_in, const HmcTail* tail) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_MISC, st);
  const HmcTail ht = tail != nullptr ? *tail : HmcTail();
  if (S <= MLP_SPLIT_MAX_S)
    hipLaunchKernelGGL(mlp_backward_kernel<MLP_SPLIT>, dim3((unsigned)S, MLP_SPLIT), dim3(MLP_THREADS), 0, st, m, S, tape, data, data_stride,
                       qoi_r, e_nn, g_theta, Sop, P, grad, g_parts, n_parts, g0_in, ht);
  else
    hipLaunchKernelGGL(mlp_backward_kernel<1>, dim3((unsigned)S), dim3(MLP_THREADS), 0, st, m, S, tape, data, data_stride, qoi_r, e_nn,
                       g_theta, Sop, P, grad, g_parts, n_parts, g0_in, ht);
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace finrom
