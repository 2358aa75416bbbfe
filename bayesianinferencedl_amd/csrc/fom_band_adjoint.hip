// The adjoint gradient on the band sweep's workspace layout (finrom_fom_gradient for batches beyond the small-batch schedule), in
// its own translation unit (fom_band.hip alone takes minutes to compile).
#include "fom_band_device.h"

namespace finrom {

namespace {

// ---- adjoint gradient on the band layout (finrom_fom_gradient: Fin.gradient, fom/forward_solve.py:293-322) --------------------------
// After the full sweep the workspace holds every column of L and w.  J = 1/2 |B_obs w - d|^2, the adjoint A v = -B_obs^T (B_obs w - d)
// is solved with the STORED factor -- one forward substitution (band_fsub: the columns stream back in elimination order) and one
// more backward sweep (band_bsweep on the region offV) -- and grad_j = sum dA_ab/dx_j v_a w_b is contracted from the two vectors
// in the workspace.  No window of the matrix is needed any more (only the right-hand-side window: NS doubles), so one wave per
// 64 samples serves every mesh, m = 16 / 20 included.  Traffic per sample ~ 2 nL + 6 n + 2 nnz(dA/dx) doubles.
template <int NS, bool POST, int NXM, int RBX = 0>
__device__ __forceinline__ void band_fsub(const BandDev& p, const Io& io, const PostTables& T, const int* __restrict__ iface, int e0,
                                          int npiv, int ntot, int L0, int offR, double (&xy)[NXM]) {
  double yw[NS];
  static_for<0, NS>([&](auto i) { yw[decltype(i)::value] = 0.0; });
  // the column of pivot v (l_1 .. l_B, 1/L_vv) and its right-hand side are requested RB pivots ahead (ring slots compile-time)
  // (RBX: an explicit, shallower ring -- this kernel has no matrix window, so several waves share a SIMD and hide each other's
  // latency: a deep ring per wave would only cost it that occupancy)
  constexpr int RB = RBX > 0 ? RBX : NS % 7 == 0 ? 7 : NS % 5 == 0 ? 5 : NS % 3 == 0 ? 3 : NS % 2 == 0 ? 2 : NS;
  constexpr int U = NS / gcd_c(NS, RB) * RB;
  double lb[RB][NS + 1];
  auto request = [&](auto rc, int nx) {
    constexpr int rs = decltype(rc)::value;
    if (nx < npiv) {
      const int base = p.offL + L0 + nx * NS;
      static_for<0, NS>([&](auto sc) { constexpr int s_ = decltype(sc)::value; lb[rs][s_] = io.template ldk<s_>(base); });
      lb[rs][NS] = io.ld(offR + e0 + nx);
    }
  };
  static_for<0, RB>([&](auto rc) { request(rc, decltype(rc)::value); });
  for (int p0 = 0; p0 < npiv; p0 += U) {
    static_for<0, U>([&](auto uc) {
      constexpr int uu = decltype(uc)::value, u = uu % NS, rs = uu % RB;
      const int v = p0 + uu;
      if (v < npiv) {
        const double zp = (yw[u] + lb[rs][NS]) * lb[rs][NS - 1];
        io.st(zp, offR + e0 + v);
        static_for<1, NS>([&](auto sc) { constexpr int s_ = decltype(sc)::value; yw[(u + s_) % NS] = fma(-lb[rs][s_ - 1], zp, yw[(u + s_) % NS]); });
        double enter = 0.0;                                // node v + NS takes the slot (an extra brings what it has collected)
        if constexpr (POST) {
          const int am = T.act[v];
          if (am != 0) {
            int k = p.offLx + T.lx_ptr[v];
            static_for<0, NXM>([&](auto sc) {
              constexpr int sl = decltype(sc)::value;
              if (am & (1 << sl)) { xy[sl] = fma(-io.ld(k), zp, xy[sl]); ++k; }
            });
          }
          if (v + NS < ntot) {
            const int ex = T.ent_extra[v + NS];
            if (ex != 0)
              static_for<0, NXM>([&](auto sc) {
                constexpr int sl = decltype(sc)::value;
                if (ex - 1 == sl) { enter = xy[sl]; xy[sl] = 0.0; }
              });
          }
        }
        yw[u] = enter;
        request(std::integral_constant<int, rs>{}, v + RB);
      }
    });
  }
  if constexpr (!POST) {                                  // what the fin leaves on its interface nodes joins the post's right-hand side
    for (int t = 0; t < ntot - npiv; ++t) {
      const int su = (npiv + t) % NS;
      double v = 0.0;
      static_for<0, NS>([&](auto uc) { constexpr int u = decltype(uc)::value; v = su == u ? yw[u] : v; });
      const int off = offR + iface[t];
      io.st(io.ld(off) + v, off);
    }
  }
}

#ifndef ADJ_RING
#define ADJ_RING 3
#endif
template <int NSF, int NSP, int NXM>
__global__ __launch_bounds__(64) void fom_band_adjoint_kernel(BandDev p, const int* __restrict__ act, const int* __restrict__ lx_ptr,
                                                              const int* __restrict__ ent_extra, const int* __restrict__ iface_elim,
                                                              const int* __restrict__ bt_ptr, const int* __restrict__ bt_obs,
                                                              const double* __restrict__ bt_w, const int* __restrict__ g_ptr,
                                                              const int* __restrict__ g_a, const int* __restrict__ g_b,
                                                              const double* __restrict__ g_w, double* __restrict__ Gw, int64_t S,
                                                              const double* __restrict__ qoi, const double* __restrict__ data,
                                                              int64_t data_stride, double* __restrict__ gradT, double* __restrict__ Jout) {
  extern __shared__ __attribute__((aligned(16))) double alds[];      // [n_obs][64] residuals | [NXM][64] the extras' solution values
  const int lane = threadIdx.x;
  const int64_t blk = blockIdx.x;
  double* __restrict__ Gs = Gw + blk * (int64_t)p.gsize * 64;
  Io io{__builtin_amdgcn_make_buffer_rsrc(Gs, 0, p.gsize * 512, 0x00020000), lane * 8};
  double* rs = alds + lane;
  double* wx = alds + (size_t)p.n_obs * 64 + lane;
  const PostTables T{act, lx_ptr, ent_extra, nullptr, nullptr, nullptr};
  int64_t s = blk * 64 + lane;
  const bool live = s < S;
  if (!live) s = S - 1;                                   // tail lanes replicate the last sample
  double jl = 0.0;
  for (int o = 0; o < p.n_obs; ++o) {
    const double r = qoi[s * p.n_obs + o] - data[(data_stride ? s * data_stride : 0) + o];
    rs[o * 64] = r;
    jl = fma(r, r, jl);
  }
  for (int i = 0; i < p.n; ++i) {                         // b = -B_obs^T r over elimination indices
    double acc = 0.0;
    for (int t = bt_ptr[i], t1 = bt_ptr[i + 1]; t < t1; ++t) acc = fma(-bt_w[t], rs[bt_obs[t] * 64], acc);
    io.st(acc, p.offV + i);
  }
  double xy[NXM];
  static_for<0, NXM>([&](auto i) { xy[decltype(i)::value] = 0.0; });
  constexpr int RBF = ADJ_RING, RBP = ADJ_RING;          // columns in flight per wave and sweep
  for (int f = 0; f < p.nfins; ++f)
    band_fsub<NSF, false, NXM, RBF>(p, io, T, iface_elim + f * p.nif, f * p.npf, p.npf, p.npf + p.nif, f * p.npf * NSF, p.offV, xy);
  band_fsub<NSP, true, NXM, RBP>(p, io, T, nullptr, p.post_e0, p.npost, p.npost, p.post_L0, p.offV, xy);
  band_bsweep<NSP, true, NXM, RBP>(p, io, wx, T, nullptr, p.post_e0, p.npost, p.npost, p.post_L0, p.offV);
  for (int f = 0; f < p.nfins; ++f)
    band_bsweep<NSF, false, NXM, RBF>(p, io, wx, T, iface_elim + f * p.nif, f * p.npf, p.npf, p.npf + p.nif, f * p.npf * NSF, p.offV);
  // grad_j = sum dA_ab/dx_j v_a w_b, loads batched by GB pairs (a failed factorisation left NaN in w: it propagates).  16 pairs =
  // 32 loads in flight: a lone wave (the one-block batches of a scalar caller) is latency-bound on exactly this loop
  constexpr int GB = 16;
  for (int j = 0; j < p.xdim; ++j) {
    double g0 = 0.0, g1 = 0.0;
    const int t0 = g_ptr[j], t1 = g_ptr[j + 1];
    for (int t = t0; t < t1; t += GB) {
      double va[GB], wb[GB];
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        const int tt = (t + u < t1) ? t + u : t1 - 1;
        va[u] = io.ld(p.offV + g_a[tt]);
        wb[u] = io.ld(p.offY + g_b[tt]);
      }
#pragma unroll
      for (int u = 0; u < GB; u += 2) {
        g0 = fma((t + u < t1) ? g_w[t + u] : 0.0, va[u] * wb[u], g0);
        g1 = fma((t + u + 1 < t1) ? g_w[t + u + 1] : 0.0, va[u + 1] * wb[u + 1], g1);
      }
    }
    gradT[(blk * p.xdim + j) * 64 + lane] = g0 + g1;
  }
  if (live) Jout[s] = 0.5 * jl;
}

// ---- general right-hand sides with the stored factor (finrom_fom_solve_rhs: the incremental solves of Fin.hessian_action,
// fom/forward_solve.py:344-368) -------------------------------------------------------------------------------------------------
// After the full sweep:  out_k = A^-1 rhs_k  for k < nrhs.  rhsT / outT: sample-blocked [blk][nrhs * n][64] in DOF order (pack_kernel /
// unpack_w_kernel layouts); the permutation to elimination order happens here (line perm[e] of the block).
template <int NSF, int NSP, int NXM>
__global__ __launch_bounds__(64) void fom_band_resolve_kernel(BandDev p, const int* __restrict__ act, const int* __restrict__ lx_ptr,
                                                              const int* __restrict__ ent_extra, const int* __restrict__ iface_elim,
                                                              const int* __restrict__ perm, double* __restrict__ Gw, int nrhs,
                                                              const double* __restrict__ rhsT, double* __restrict__ outT) {
  extern __shared__ __attribute__((aligned(16))) double alds[];      // [NXM][64] the extras' solution values
  const int lane = threadIdx.x;
  const int64_t blk = blockIdx.x;
  double* __restrict__ Gs = Gw + blk * (int64_t)p.gsize * 64;
  Io io{__builtin_amdgcn_make_buffer_rsrc(Gs, 0, p.gsize * 512, 0x00020000), lane * 8};
  double* wx = alds + lane;
  const PostTables T{act, lx_ptr, ent_extra, nullptr, nullptr, nullptr};
  const int64_t d = (int64_t)nrhs * p.n;
  for (int k = 0; k < nrhs; ++k) {
    const double* __restrict__ src = rhsT + (blk * d + (int64_t)k * p.n) * 64 + lane;
    double* __restrict__ dst = outT + (blk * d + (int64_t)k * p.n) * 64 + lane;
    for (int e = 0; e < p.n; ++e) io.st(src[(int64_t)perm[e] * 64], p.offV + e);
    double xy[NXM];
    static_for<0, NXM>([&](auto i) { xy[decltype(i)::value] = 0.0; });
    for (int f = 0; f < p.nfins; ++f)
      band_fsub<NSF, false, NXM, ADJ_RING>(p, io, T, iface_elim + f * p.nif, f * p.npf, p.npf, p.npf + p.nif, f * p.npf * NSF, p.offV, xy);
    band_fsub<NSP, true, NXM, ADJ_RING>(p, io, T, nullptr, p.post_e0, p.npost, p.npost, p.post_L0, p.offV, xy);
    band_bsweep<NSP, true, NXM, ADJ_RING>(p, io, wx, T, nullptr, p.post_e0, p.npost, p.npost, p.post_L0, p.offV);
    for (int f = 0; f < p.nfins; ++f)
      band_bsweep<NSF, false, NXM, ADJ_RING>(p, io, wx, T, iface_elim + f * p.nif, f * p.npf, p.npf, p.npf + p.nif, f * p.npf * NSF, p.offV);
    for (int e = 0; e < p.n; ++e) dst[(int64_t)perm[e] * 64] = io.ld(p.offV + e);
  }
}

template <int NSF, int NSP, int NXM>
int launch_res(const BandDev& p, double* Gw, int64_t nblk, int nrhs, const double* rhsT, double* outT, hipStream_t st) {
  hipLaunchKernelGGL((fom_band_resolve_kernel<NSF, NSP, NXM>), dim3((unsigned)nblk), dim3(64), (size_t)NXM * 64 * sizeof(double), st, p, p.act,
                     p.lx_ptr, p.ent_extra, p.iface_elim, p.perm, Gw, nrhs, rhsT, outT);
  FR_HIP(hipGetLastError());
  return 0;
}

template <int NSF, int NSP, int NXM>
int launch_adj(const BandDev& p, const BandGradDev& g, double* Gw, int64_t nblk, int64_t S, const double* qoi, const double* data,
               int64_t data_stride, double* gradT, double* J, hipStream_t st) {
  const size_t lds = (size_t)(p.n_obs + NXM) * 64 * sizeof(double);
  // (the attribute is set once per device to the CAP, not to this handle's size: a later handle with more observation rows on the
  //  same window sizes must not be refused its larger dynamic LDS; launch_fom_band_adjoint has checked lds <= the cap)
  static PerDeviceOnce once;
  if (lds > 64 * 1024)
    if (int rc = once.run([&]() -> int {
          FR_HIP(hipFuncSetAttribute((const void*)fom_band_adjoint_kernel<NSF, NSP, NXM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          return 0; })) return rc;
  hipLaunchKernelGGL((fom_band_adjoint_kernel<NSF, NSP, NXM>), dim3((unsigned)nblk), dim3(64), lds, st, p, p.act, p.lx_ptr, p.ent_extra,
                     p.iface_elim, g.bt_ptr, g.bt_obs, g.bt_w, g.g_ptr, g.g_a, g.g_b, g.g_w, Gw, S, qoi, data, data_stride, gradT, J);
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int launch_fom_band_adjoint(const BandDev& p, const BandGradDev& g, double* Gw, int64_t nblk, int64_t S, const double* qoi,
                            const double* data, int64_t data_stride, double* gradT, double* J, hipStream_t st) {
  if (nblk == 0) return 0;
  ScopedKernelTimer t(p.NSP <= 14 ? K_FOM_PATH_BAND_REG : K_FOM_PATH_BAND_LDSW, st);
  if ((size_t)(p.n_obs + 12) * 64 * sizeof(double) > 160 * 1024) { set_error("fom band adjoint: too many observations for LDS"); return FINROM_ERR_UNSUPPORTED; }
#define FR_A(A, B, X) if (p.NSF == A && p.NSP == B) return launch_adj<A, B, X>(p, g, Gw, nblk, S, qoi, data, data_stride, gradT, J, st);
  FR_A(3, 6, 4) FR_A(4, 10, 4) FR_A(5, 14, 4) FR_A(6, 18, 8) FR_A(7, 22, 8) FR_A(8, 26, 10) FR_A(9, 30, 12)
#undef FR_A
  set_error("fom band adjoint: unsupported window sizes");
  return FINROM_ERR_UNSUPPORTED;
}

int launch_fom_band_resolve(const BandDev& p, double* Gw, int64_t nblk, int nrhs, const double* rhsT, double* outT, hipStream_t st) {
  if (nblk == 0 || nrhs == 0) return 0;
  ScopedKernelTimer t(p.NSP <= 14 ? K_FOM_PATH_BAND_REG : K_FOM_PATH_BAND_LDSW, st);
#define FR_R(A, B, X) if (p.NSF == A && p.NSP == B) return launch_res<A, B, X>(p, Gw, nblk, nrhs, rhsT, outT, st);
  FR_R(3, 6, 4) FR_R(4, 10, 4) FR_R(5, 14, 4) FR_R(6, 18, 8) FR_R(7, 22, 8) FR_R(8, 26, 10) FR_R(9, 30, 12)
#undef FR_R
  set_error("fom band resolve: unsupported window sizes");
  return FINROM_ERR_UNSUPPORTED;
}

}  // namespace finrom
