// Device pieces shared by the translation units of the frontal band sweep (fom_band.hip: factorisation + forward solve;
// fom_band_adjoint.hip: the adjoint gradient on the same workspace layout): compile-time loops, the workspace as a buffer
// resource, the extras' LDS layout, the backward sweep.  See fom_band.hip for the algorithm.
#pragma once
#include "finrom_core.h"
#include <type_traits>

namespace finrom {

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
constexpr __device__ __host__ int ent_s(int e) { int s = 1; while (e >= s) { e -= s; ++s; } return s; }      // row of packed entry e (rows from 1)
constexpr __device__ __host__ int tri(int a, int b) { return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// the columns of L are written once and read once: non-temporal (aux bit 1 = nt on gfx94x/95x) keeps them from displacing the
// projection kernel's tables in L2 when the two kernels run side by side
#ifndef BAND_STREAM_AUX
#define BAND_STREAM_AUX 2
#endif

struct Io {                       // the wave's slice of the workspace as a buffer resource: SGPR offsets, no address arithmetic
  __amdgpu_buffer_rsrc_t r; int lane8;
  __device__ __forceinline__ double ld(int elem) const {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, lane8, elem * 512, 0));
  }
  template <int K> __device__ __forceinline__ double ldk(int elem) const {      // element elem + K, K folded into the offset field
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, lane8 + (K % 8) * 512, (elem + K / 8 * 8) * 512, BAND_STREAM_AUX));
  }
  __device__ __forceinline__ void st(double v, int elem) const {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, lane8, elem * 512, 0);
  }
  template <int K> __device__ __forceinline__ void stk(double v, int elem) const {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, lane8 + (K % 8) * 512, (elem + K / 8 * 8) * 512, BAND_STREAM_AUX);
  }
};

// LDS state of the extras, per lane (index * 64 + lane): X[NXM][NS] | XD[NXM] | XY[NXM] | XX[pairs a > b] | WX[NXM]
template <int NS, int NXM> struct XL {
  static constexpr int X = 0, XD = NXM * NS, XY = XD + NXM, XX = XY + NXM, WX = XX + NXM * (NXM - 1) / 2, SIZE = WX + NXM;
  static constexpr __device__ int xx(int a, int b) { return XX + (a > b ? a * (a - 1) / 2 + b : b * (b - 1) / 2 + a); }
};

struct PostTables {
  const int* act; const int* lx_ptr; const int* ent_extra; const int* ecp_ptr; const int* ecp_slot; const int* ecp_off;
};

constexpr int gcd_c(int a, int b) { return b == 0 ? a : gcd_c(b, a % b); }

// (offR: the region the right-hand side is read from and the solution written to -- p.offY for the forward solve, p.offV for the
// adjoint solve of the gradient; wx: LDS, [NXM][64], the extras' solution values: XL::WX inside the forward kernels' state)
template <int NS, bool POST, int NXM, int RBX = 0>
__device__ __forceinline__ void band_bsweep(const BandDev& p, const Io& io, double* __restrict__ wx, const PostTables& T,
                                            const int* __restrict__ iface, int e0, int npiv, int ntot, int L0, int offR) {
  double ww[NS];
  static_for<0, NS>([&](auto i) { ww[decltype(i)::value] = 0.0; });
  if constexpr (!POST) {                                 // the fin's trailing nodes are post nodes whose w is known
    for (int t = 0; t < ntot - npiv; ++t) {
      const double v = io.ld(offR + iface[t]);
      const int su = (npiv + t) % NS;
      static_for<0, NS>([&](auto uc) { constexpr int u = decltype(uc)::value; ww[u] = su == u ? v : ww[u]; });
    }
  } else {
    static_for<0, NXM>([&](auto sc) { wx[decltype(sc)::value * 64] = 0.0; });
  }
  // Software-pipelined like the forward sweep: the column of pivot v (NS values) and y_v were requested RB steps ago into ring
  // slot v mod RB (RB divides NS: compile-time slots); after using them the step requests pivot v - RB into the same slot.
  // RB columns in flight per wave (7 x 7.5 KB in the post at m = 12) is what keeps HBM busy with one wave per SIMD.
  // (RBX: an explicit ring depth that need not divide NS -- the loop is then unrolled lcm(NS, RB) times)
  constexpr int RB = RBX > 0 ? RBX : NS % 7 == 0 ? 7 : NS % 5 == 0 ? 5 : NS % 3 == 0 ? 3 : NS % 2 == 0 ? 2 : NS;
  constexpr int U = NS / gcd_c(NS, RB) * RB;
  // PFX (with RBX, the one-wave-per-CU callers): the pivot's table entries and its couplings to the extras travel with the
  // column -- requested RB steps ahead, their own scalars one step before that -- instead of being fetched where they are used
  constexpr bool PFX = POST && RBX > 0;
  double lb[RB][NS + 1];
  double lxv[PFX ? RB : 1][NXM];
  int am_r[PFX ? RB : 1], ex_r[PFX ? RB : 1];
  const int vtop = npiv - 1 + RB;
  int am_q = 0, k_q = 0, ex_q = 0;                       // tables of pivot v - RB while step v runs (PFX)
  auto tables = [&](int q) { if (q >= 0 && q < npiv) { am_q = T.act[q]; k_q = p.offLx + T.lx_ptr[q]; ex_q = T.ent_extra[q]; } else { am_q = 0; ex_q = 0; } };
  if constexpr (PFX) tables(vtop - RB);                  // = npiv - 1, the pivot whose column the first executed step requests
  for (int p0 = vtop / U * U; p0 >= 0; p0 -= U) {
    static_for<0, U>([&](auto rc) {
      constexpr int uu = U - 1 - decltype(rc)::value;
      constexpr int u = uu % NS;
      constexpr int rs = uu % RB;
      const int v = p0 + uu;
      if (v <= vtop) {
        if (v < npiv) {
          double acc = lb[rs][NS];
          static_for<1, NS>([&](auto sc) { constexpr int s = decltype(sc)::value; acc = fma(-lb[rs][s - 1], ww[(u + s) % NS], acc); });
          if constexpr (PFX) {
            const int am = am_r[rs];
            if (am != 0)
              static_for<0, NXM>([&](auto sc) {
                constexpr int sl = decltype(sc)::value;
                if (am & (1 << sl)) acc = fma(-lxv[rs][sl], wx[sl * 64], acc);
              });
          } else if constexpr (POST) {
            const int am = T.act[v];
            if (am != 0) {
              int k = p.offLx + T.lx_ptr[v];
              static_for<0, NXM>([&](auto sc) {
                constexpr int sl = decltype(sc)::value;
                if (am & (1 << sl)) { acc = fma(-io.ld(k), wx[sl * 64], acc); ++k; }
              });
            }
          }
          const double wv = acc * lb[rs][NS - 1];
          io.st(wv, offR + e0 + v);
          ww[u] = wv;
          if constexpr (POST) {
            const int ex = PFX ? ex_r[rs] : T.ent_extra[v];
            if (ex != 0) wx[(ex - 1) * 64] = wv;   // this node is an extra of earlier pivots
          }
        }
        const int nx = v - RB;
        if (nx >= 0 && nx < npiv) {
          const int base = p.offL + L0 + nx * NS;
          static_for<0, NS>([&](auto sc) { constexpr int s = decltype(sc)::value; lb[rs][s] = io.template ldk<s>(base); });   // l_1..l_B, 1/L_jj
          lb[rs][NS] = io.ld(offR + e0 + nx);
          if constexpr (PFX) {
            am_r[rs] = am_q; ex_r[rs] = ex_q;
            if (am_q != 0) {
              int k = k_q;
              static_for<0, NXM>([&](auto sc) {
                constexpr int sl = decltype(sc)::value;
                if (am_q & (1 << sl)) { lxv[rs][sl] = io.ld(k); ++k; }
              });
            }
          }
        }
        if constexpr (PFX) tables(nx - 1);
      }
    });
  }
}

}  // namespace

}  // namespace finrom
