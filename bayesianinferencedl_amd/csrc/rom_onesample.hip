// The one-sample call pattern of the reduced model (batches of <= 64 samples: MAP / HMC, BASELINE configs[4]): LATENCY, not
// throughput.  AffineROMFin.forward_nine_param_reduced + qoi_reduced (rom/averaged_affine_ROM.py:278-333) and, with `grad`, the
// adjoint solve of grad_reduced (:335-356) for a handful of samples whose inputs depend on the previous call's outputs.
//
// Round 2's kernel (rom_proj_splitk_kernel: the sample's k-steps over the four waves of ONE workgroup, wave 0 then factors and
// substitutes by shuffle steps) took 178 us at r = 81; by phase clocks 76 us were the contraction, 23 us the factorisation and
// 2 x 35 us the two pairs of substitutions (2 r dependent pivot steps each, an fp64 division and LDS-crossbar shuffles per step).
// Two kernels here:
//   rom_small_proj_kernel   the sample's k-steps over NC workgroups x 4 waves (one wave per SIMD of NC CUs); a workgroup adds its
//                           four partial block triangles through LDS in a fixed order and writes ONE partial to global scratch;
//   rom_small_solve_kernel  one workgroup per sample: its four waves add the NC partials (fixed order: deterministic), then
//                           wave 0 factors and solves in MFMA form, LEFT-looking, with the tiles in LDS -- only one block row
//                           of tiles is ever in registers, so hipcc's allocator has nothing to spill and the MFMA builtin (with
//                           the compiler's own hazard handling) is good enough: this is one wave, not a pipe to keep full.
//     forward   per block row kb: T(kb, tj) = A(kb, tj) - sum_{j < kb} U(j, kb)^T U(j, tj); the diagonal tile by 16 shuffle steps
//               on [T_kk | I] (-> M_kb = U_kk^-T, carried transposed = as an A operand); U(kb, tj) = M_kb T(kb, tj) and the extra
//               column Z_kb = M_kb (G_kb - sum_j U(j, kb)^T Z_j), G = [B_r | (B_obs Phi)^T]: 4 MFMAs per tile;
//     middle    qoi_r = Z[:, 1:]^T Z[:, 0];  J;  U^-T (B_obs Phi)^T (data - qoi_r) = Z[:, 1:] (data - qoi_r) -- a combination of
//               columns the forward sweep already produced: no second forward substitution;
//     backward  ONE block substitution for both right-hand sides: X_kb = M_kb^T (R_kb - sum_{j > kb} U(kb, j) X_j); M_kb in its
//               C/D layout IS the A operand of M_kb^T R, U(kb, j) is read from LDS transposed.  w_r = X[:, 0], v_r = X[:, 1].
#include "rom_proj_device.h"
#include "mlp_device.h"

namespace finrom {

namespace {

template <int NB> constexpr int onesample_nt() { return NB * (NB + 1) / 2; }

// ---- contraction: partial block triangle of workgroup (s, j) ---------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256, 1) void rom_small_proj_kernel(RomDev p, const double* __restrict__ theta, int64_t S, int NC,
                                                                double* __restrict__ part, const int* __restrict__ kpat, MlpFuse fm) {
  constexpr int NT = onesample_nt<NB>();
  extern __shared__ __attribute__((aligned(16))) double red_lds[];      // 3 partial triangles: NT x 256 doubles each
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t s = blockIdx.x;
  const int j = blockIdx.y;
  // workgroups NC .. of a sample (finrom_romml_grad only): the learned error model's FIRST LAYER, which this contraction does not
  // depend on -- as a kernel of its own the forward pass was 19 us at the head of the one-sample call; as ONE spare workgroup here
  // (round 3) it took as long as the contraction (25 us); now nw0 workgroups take a share of the rows each and the layers behind
  // are a spare wave of the solve kernel (mlp_device.h)
  if (fm.on && j >= NC) {                               // (workgroups NC .. NC + nw0 - 1: their rows of the network's first layer)
    mlp_first_layer_part<256>(fm.m, fm.k, s, fm.mom, fm.eps, fm.k_out, j - NC, fm.nw0, fm.y0_part + (s * fm.nw0 + (j - NC)) * 64,
                              (float*)red_lds, (int)threadIdx.x);
    return;
  }
  const int q = lane >> 4, c = lane & 15;
  if (fm.on) {
    // theta_p = sum_i S[p][i] k[i] for this sample, by this workgroup itself (every workgroup of the sample: same sums, same order):
    // a wave takes a quarter of the columns for ALL rows, every load of its slice requested at once; the four partial sums in a fixed
    // order.  The result goes to the workgroup's own 16 doubles of scratch, from where the main loop's scalar loads read it
    // (stores acknowledged by L2, then the scalar cache invalidated).
    __shared__ double th_part[4][16];
    const int n = fm.m.n_in, P = fm.P;
    double* __restrict__ tscr = fm.theta_scr + (s * NC + j) * 16;
    if (fm.theta_parts != nullptr) {                     // (a leapfrog step behind the first: the previous step left the sums)
      if ((int)threadIdx.x < P) {
        const int pp = threadIdx.x;
        const double* tp = fm.theta_parts + s * (HMC_THETA_PARTS * 16) + pp;
        double v[HMC_THETA_PARTS];
#pragma unroll
        for (int w = 0; w < HMC_THETA_PARTS; ++w) v[w] = tp[w * 16];
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < HMC_THETA_PARTS; ++w) t += v[w];
        tscr[pp] = t;
        if (j == 0) fm.theta_out[s * P + pp] = t;
      }
    } else {
    const double* __restrict__ krow = fm.k + s * (int64_t)n;
    const double* __restrict__ mrow = fm.mom != nullptr ? fm.mom + s * (int64_t)n : nullptr;      // (leapfrog: the field is k + eps * mom)
    const int i0 = (int)((int64_t)n * wave / 4), i1 = (int)((int64_t)n * (wave + 1) / 4);
    double tacc[16];
#pragma unroll
    for (int pp = 0; pp < 16; ++pp) tacc[pp] = 0.0;
    constexpr int TU = 7;                                // 7 x 64 columns per pass: one pass covers a quarter of 1597
    for (int ib = i0 + lane; ib < i1; ib += 64 * TU) {
      double kv[TU], sv[16][TU];
#pragma unroll
      for (int u = 0; u < TU; ++u) kv[u] = ib + 64 * u < i1 ? krow[ib + 64 * u] : 0.0;
      if (mrow != nullptr) {
#pragma unroll
        for (int u = 0; u < TU; ++u) kv[u] = ib + 64 * u < i1 ? fma(fm.eps, mrow[ib + 64 * u], kv[u]) : 0.0;
      }
#pragma unroll
      for (int pp = 0; pp < 16; ++pp)
        if (pp < P) {
#pragma unroll
          for (int u = 0; u < TU; ++u) sv[pp][u] = ib + 64 * u < i1 ? fm.Sop[(int64_t)pp * n + ib + 64 * u] : 0.0;
        }
#pragma unroll
      for (int pp = 0; pp < 16; ++pp)
        if (pp < P) {
#pragma unroll
          for (int u = 0; u < TU; ++u) tacc[pp] = fma(sv[pp][u], kv[u], tacc[pp]);
        }
    }
#pragma unroll
    for (int pp = 0; pp < 16; ++pp)
      if (pp < P) {
        double x = tacc[pp];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (lane == 0) th_part[wave][pp] = x;
      }
    __syncthreads();
    if ((int)threadIdx.x < P) {
      const int pp = threadIdx.x;
      const double t = ((th_part[0][pp] + th_part[1][pp]) + th_part[2][pp]) + th_part[3][pp];
      tscr[pp] = t;
      if (j == 0) fm.theta_out[s * P + pp] = t;
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();
    theta = tscr - s * p.P;                              // (the pointer arithmetic below adds s P back)
  }
  const unsigned long long ta = (unsigned long long)(theta + s * p.P);
  const double* theta_s = (const double*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ta >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ta));
  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
  // k-steps [k0, k0 + cnt) of the pattern-uniform tables; parts start at even k-steps (the loop is unrolled twice)
  const int parts = 4 * NC, kpart = j * 4 + wave;
  const int per = ((p.nku + parts - 1) / parts + 1) / 2 * 2, k0 = kpart * per;
  const int cnt = k0 >= p.nku ? 0 : (p.nku - k0 < per ? p.nku - k0 : per);
  if (cnt > 0) proj_main_uniform<NB>(p, kpat + 8 * k0, theta_s, q, c, acc, cnt);
  mfma_drain(acc);
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) red_lds[((wave - 1) * NT + t) * 256 + g * 64 + lane] = acc[t][g];
  }
  __syncthreads();
  if (wave > 0) return;
  for (int w = 1; w < 4; ++w)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[t][g] += red_lds[((w - 1) * NT + t) * 256 + g * 64 + lane];
  // tile t, natural layout: element (row q + 4 g, column c) at [t][(q + 4 g) * 16 + c]
  double* __restrict__ dst = part + ((s * NC + j) * (int64_t)NT) * 256;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) dst[t * 256 + (q + 4 * g) * 16 + c] = acc[t][g];
}

// ---- factorisation + forward / adjoint solves -------------------------------------------------------------------------------------
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4d mma(double a, double b, v4d cacc) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, cacc, 0, 0, 0); }

constexpr int MLP_STAGE_FLOATS = 16 * 256 * 4;      // what 256 threads stage with 16 four-float loads each: 64 KB (5 x 50 x 50 + 50 x 9 = 12 950 floats fit)
template <int NB> constexpr int small_solve_lds(bool mlp = false) { return onesample_nt<NB>() * 256 + 2 * NB * 256 + 16 * NB + 32 + 32 + 32 + 64 + 2 + (mlp ? MLP_STAGE_FLOATS / 2 : 0); }

template <int NB>
__global__ __launch_bounds__(320, 1) void rom_small_solve_kernel(RomDev p, const double* __restrict__ theta, int64_t S, int NC,
                                                                 const double* __restrict__ part, int grad, RomGradArgs ga,
                                                                 double* __restrict__ w_r, double* __restrict__ qoi_r,
                                                                 int* __restrict__ info, MlpFuse fm) {
  constexpr int NT = onesample_nt<NB>();
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* __restrict__ tl = lds;                      // [NT][256] A tiles, overwritten by U tiles; natural layout (row * 16 + col)
  double* __restrict__ mbuf = tl + NT * 256;          // [NB][256] M_kb, natural layout
  double* __restrict__ zl = mbuf + NB * 256;          // [NB][256] the extra tile column Z, natural layout
  double* __restrict__ bl = zl + NB * 256;            // [16 NB] B_r
  double* __restrict__ th = bl + 16 * NB;             // [32] 1, theta_1..P, 0
  double* __restrict__ dsh = th + 32;                 // [32] data - e_NN of this sample (the spare wave's hand-over, fm.on)
  float* __restrict__ mlp_y = (float*)(dsh + 32);     // [64] + [64] floats: the spare wave's layer vectors
  int* __restrict__ mlp_flag = (int*)(mlp_y + 128);   // arrivals of the four staging waves (+ 3 ints of padding)
  float* __restrict__ mlp_wl = mlp_y + 128 + 4;       // [MLP_STAGE_FLOATS] the error model's hidden + head weights (fm.on)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t s = blockIdx.x;
  const int q = lane >> 4, c = lane & 15;
  // the NC partial triangles, added in a fixed order: tile t by wave t % 4
#ifdef FINROM_SOLVE_CLOCKS
  long long ck[8]; ck[0] = wall_clock64(); ck[6] = 0;
#endif
  const double* __restrict__ src = part + (s * NC * (int64_t)NT) * 256;
  // (a wave's tiles three at a time, eight partials of each requested together: the sums are a few trips to L2, and one tile per
  // trip -- the first version -- made this phase 12 of the kernel's 50 us)
  {
    constexpr int TB = 3, JB = 8;
    // (wave 4 exists in finrom_romml_grad's one-sample form only: the error model's layers behind the first, whose result -- the
    //  adjoint's data, data - e_NN -- wave 0 needs at the middle of the kernel; it meets the others at every barrier)
    // Its weights (52 KB, cold: every batch of them fetched by a lone wave was a trip beyond L2, the walk 30 us) are STAGED in LDS
    // by waves 0 .. 3 -- sixteen 16-byte loads per lane, all in flight at once, in front of their own partial-sum loads -- and wave 4
    // waits for the four arrivals on an LDS counter (the hardware barrier would tie the others to it; the counter is zeroed behind
    // a barrier at the kernel's first instruction, when nobody has anything to wait for)
    const int n_stage = fm.on ? fm.m.n_layers * fm.m.n_w * fm.m.n_w + fm.m.n_w * fm.m.n_out : 0;
    const bool staged = fm.on && n_stage <= MLP_STAGE_FLOATS && ((fm.m.n_layers * fm.m.n_w * fm.m.n_w) & 3) == 0;
    if (fm.on) {
      if (threadIdx.x == 0) *mlp_flag = 0;
      __syncthreads();
      if (staged && wave < 4) {
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const int nh4 = (fm.m.n_layers * fm.m.n_w * fm.m.n_w) >> 2, nt4 = (n_stage + 3) >> 2;
        f4_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int c4 = (int)threadIdx.x + 256 * u;
          f4_t x = (f4_t){0.f, 0.f, 0.f, 0.f};
          if (c4 < nh4) x = ((const f4_t*)fm.m.W)[c4];
          else if (c4 < nt4) {                             // the head (its start need not be 16-byte aligned in the staged array)
            const int e = 4 * (c4 - nh4), ne = fm.m.n_w * fm.m.n_out;
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = e + t < ne ? fm.m.Wh[e + t] : 0.f;
          }
          v[u] = x;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int c4 = (int)threadIdx.x + 256 * u;
          ((f4_t*)mlp_wl)[c4] = v[u];                      // (beyond the weights: zeros -- the unguarded loops may read there)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(mlp_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (wave == 4) {
#ifdef FINROM_SOLVE_CLOCKS
      const long long w4a = wall_clock64();
#endif
      if (staged) {
        while (__hip_atomic_load(mlp_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
#ifdef FINROM_SOLVE_CLOCKS
      const long long w4b = wall_clock64();
#endif
      if (staged) mlp_forward_tail_wave<true>(fm.m, s, fm.y0_part + s * fm.nw0 * 64, fm.nw0, fm.data, fm.data_stride, fm.tape, fm.e_out,
                                              fm.data_shift, dsh, mlp_y, mlp_y + 64, lane, mlp_wl);
      else mlp_forward_tail_wave<false>(fm.m, s, fm.y0_part + s * fm.nw0 * 64, fm.nw0, fm.data, fm.data_stride, fm.tape, fm.e_out,
                                        fm.data_shift, dsh, mlp_y, mlp_y + 64, lane, nullptr);
#ifdef FINROM_SOLVE_CLOCKS
      if (lane == 0) { dsh[30] = (double)(w4b - w4a); dsh[31] = (double)(wall_clock64() - w4b); }
#endif
    }
    for (int t0 = wave; t0 < NT && wave < 4; t0 += 4 * TB) {
      double a[TB][4];
#pragma unroll
      for (int u = 0; u < TB; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) a[u][g] = 0.0;
      for (int j0 = 0; j0 < NC; j0 += JB) {
        double v[TB][JB][4];
#pragma unroll
        for (int u = 0; u < TB; ++u)
#pragma unroll
          for (int jj = 0; jj < JB; ++jj)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int t = t0 + 4 * u, j = j0 + jj;
              v[u][jj][g] = (t < NT && j < NC) ? src[(j * NT + t) * 256 + g * 64 + lane] : 0.0;
            }
#pragma unroll
        for (int u = 0; u < TB; ++u)
#pragma unroll
          for (int jj = 0; jj < JB; ++jj)                // (partials in index order: the same sums as one at a time)
#pragma unroll
            for (int g = 0; g < 4; ++g) a[u][g] += v[u][jj][g];
      }
#pragma unroll
      for (int u = 0; u < TB; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (t0 + 4 * u < NT) tl[(t0 + 4 * u) * 256 + g * 64 + lane] = a[u][g];
    }
  }
  if (wave == 0) {
    if (lane == 0) { th[0] = 1.0; th[p.P + 1] = 0.0; }
    if (lane < p.P) th[lane + 1] = theta[s * p.P + lane];
  }
  // (no barrier here: B_r is wave 0's work on wave 0's own th[]; the tiles of the other waves are first read behind the barrier
  //  below -- which is also where the error model's spare wave has to have arrived: it has partial sums + B_r, ~9 us, to finish)
  wave_sync();
#ifdef FINROM_SOLVE_CLOCKS
  ck[1] = wall_clock64();
#endif

  // B_r = psi^T F (rom :297): F is non-zero on the root nodes only; their rows of psi are rebuilt here (wave 0)
  if (wave == 0) {
    double bacc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) bacc[b] = 0.0;
    for (int ks = 0; ks < p.rhs_nk; ++ks) {
      double v[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) v[b] = 0.0;
      for (int t = 0; t < p.rhs_nt; ++t) {
        const int row = (ks * p.rhs_nt + t) * 4 + q;
        const double thp = th[p.rhs_pidx[row]];
        const double* rsrc = p.rhs_tv + (int64_t)row * p.rp + c;
#pragma unroll
        for (int b = 0; b < NB; ++b) v[b] = fma(thp, rsrc[16 * b], v[b]);
      }
      const double fk = p.rhs_f[ks * 4 + q];
#pragma unroll
      for (int b = 0; b < NB; ++b) bacc[b] = fma(v[b], fk, bacc[b]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      double x = bacc[b];
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);
      if (q == 0) bl[16 * b + c] = x;
    }
  }
  __syncthreads();
#ifdef FINROM_SOLVE_CLOCKS
  ck[2] = wall_clock64();
#endif

  auto tile_at = [&](int ti, int tj) -> double* { return tl + (ti * NB - (ti * (ti - 1)) / 2 + (tj - ti)) * 256; };
  // column NB = the extra tile column G = [B_r | (B_obs Phi)^T] -> Z: its tile of block row j lives at zl + 256 j
  auto col_tile = [&](int ti, int tj) -> double* { return tj < NB ? tile_at(ti, tj) : zl + ti * 256; };
  const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.obs_phi), 0, p.n_obs * p.r * 8, 0x00020000);
  const int ovoff = ((c - 1) * p.r + q) * 8;            // row q of observation c - 1 (c == 0 and c > n_obs: out of range, reads 0)
  const int nat = q * 16 + c, trn = c * 16 + q;         // natural / transposed offset of element (q, c) inside a tile: + 64 g / + 4 g
  int bad = 0;
  // ---- forward (round 4: over the four waves) ---------------------------------------------------------------------------------
  // Round 3 walked the block rows on wave 0 alone: ~400 dependent-issue MFMAs of 64 cycles each on one SIMD while three waves had
  // left.  Now, per block row kb: wave 0 owns the DIAGONAL tile (left-looking sum, then diag_tile_inverse -> M_kb to LDS); the
  // tiles right of it and the extra column are dealt over waves 1 .. 3 (at most two each), which form their left-looking sums
  // T(kb, tj) = A(kb, tj) - sum_{j < kb} U(j, kb)^T U(j, tj) meanwhile, wait for M_kb at ONE barrier, turn their tiles into
  // U(kb, tj) = M_kb T(kb, tj) (Z_kb for the extra column) and hand them over through LDS at a second barrier.  Same sums in the
  // same order as before (an accumulator's terms j = 0 .. kb - 1, k-steps g = 0 .. 3): bit-identical.
  sfor<0, NB>([&](auto kc) {
    constexpr int kb = decltype(kc)::value;
    constexpr int NITEM = NB - kb;                       // tiles tj = kb + 1 .. NB - 1 and the extra column tj = NB
    v4d T[2];
    int tjs[2];
    if (wave == 0) {
      v4d D;
      const double* at = tile_at(kb, kb);
#pragma unroll
      for (int g = 0; g < 4; ++g) D[g] = at[nat + 64 * g];
      // padding rows / columns (>= r) of psi^T psi are zero: unit diagonal (kb >= 2: the look-ahead has done it)
      if constexpr (kb < 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (q + 4 * g == c && 16 * kb + c >= p.r) D[g] = 1.0;
      }
      if constexpr (kb > 0) {
        // (kb >= 2: the terms j < kb - 1 were subtracted a block row ago by the look-ahead below, in place; the newest row's is left)
        constexpr int j0 = kb >= 2 ? kb - 1 : 0;
        double ua[kb][4];
        sfor<j0, kb>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const double* ujk = tile_at(j, kb);
#pragma unroll
          for (int g = 0; g < 4; ++g) ua[j][g] = ujk[nat + 64 * g];
        });
        sfor<j0, kb>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
#pragma unroll
          for (int g = 0; g < 4; ++g) D = mma(-ua[j][g], ua[j][g], D);
        });
      }
#ifdef FINROM_SOLVE_CLOCKS
      const long long cd0 = wall_clock64();
#endif
      const v4d Mc = diag_tile_inverse(D, q, c, lane, bad);      // (4 x 4-blocked factorisation on the matrix cores, rom_proj_device.h)
      // M_kb in its natural layout: the backward sweep's operand, and (read transposed) the A operand of M T
#pragma unroll
      for (int g = 0; g < 4; ++g) mbuf[kb * 256 + nat + 64 * g] = Mc[g];
#ifdef FINROM_SOLVE_CLOCKS
      ck[6] += wall_clock64() - cd0;
#endif
    } else if (wave <= 3) {
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int item = (wave - 1) + 3 * sl;            // 0 .. NITEM - 1
        const int tj = kb + 1 + item;                    // (tj == NB: the extra column)
        tjs[sl] = item < NITEM ? tj : -1;
        if (item >= NITEM) continue;
        if (tj < NB) {
          const double* at = tile_at(kb, tj);
#pragma unroll
          for (int g = 0; g < 4; ++g) T[sl][g] = at[nat + 64 * g];
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 16 * kb + q + 4 * g;
            double v = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ores, ovoff, (16 * kb + 4 * g) * 8, 0));
            if (row >= p.r) v = 0.0;
            if (c == 0) v = bl[row];
            T[sl][g] = v;
          }
        }
        if constexpr (kb > 0) {
          // every operand of the tile is requested from LDS before the first MFMA
          double ua[kb][4], ub[kb][4];
          sfor<0, kb>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const double* ujk = tile_at(j, kb);
            const double* ujt = col_tile(j, tj);
#pragma unroll
            for (int g = 0; g < 4; ++g) { ua[j][g] = -ujk[nat + 64 * g]; ub[j][g] = ujt[nat + 64 * g]; }
          });
          sfor<0, kb>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) T[sl] = mma(ua[j][g], ub[j][g], T[sl]);
          });
        }
      }
      // LOOK-AHEAD: the next diagonal tile's left-looking sum over the block rows that are already final (j < kb), in place --
      // wave 0 then only subtracts block row kb's term before it factors the tile (the chain of 4 kb dependent MFMAs in front of
      // every diag_tile_inverse was on the sweep's critical path while three waves waited for M_kb)
      if constexpr (kb >= 1 && kb + 1 < NB) {
        if (wave == 1 + NITEM % 3) {
          double* dt = tile_at(kb + 1, kb + 1);
          v4d Dn;
#pragma unroll
          for (int g = 0; g < 4; ++g) Dn[g] = dt[nat + 64 * g];
#pragma unroll
          for (int g = 0; g < 4; ++g)
            if (q + 4 * g == c && 16 * (kb + 1) + c >= p.r) Dn[g] = 1.0;
          double ua[kb][4];
          sfor<0, kb>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const double* ujk = tile_at(j, kb + 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) ua[j][g] = ujk[nat + 64 * g];
          });
          sfor<0, kb>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) Dn = mma(-ua[j][g], ua[j][g], Dn);
          });
#pragma unroll
          for (int g = 0; g < 4; ++g) dt[nat + 64 * g] = Dn[g];
        }
      }
    }
    __syncthreads();                                     // M_kb is in LDS
    if (wave > 0 && wave <= 3) {
      double Am[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) Am[g] = mbuf[kb * 256 + trn + 4 * g];      // Am[g] at lane (q, c) = M[c][q + 4 g]
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        if (tjs[sl] < 0) continue;
        v4d n = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < 4; ++g) n = mma(Am[g], T[sl][g], n);
        double* ut = col_tile(kb, tjs[sl]);
#pragma unroll
        for (int g = 0; g < 4; ++g) ut[nat + 64 * g] = n[g];
      }
    }
    __syncthreads();                                     // block row kb of U and Z_kb are in LDS
  });
  if (wave > 0) return;
  v4d z[NB];
  sfor<0, NB>([&](auto kc) {
    constexpr int kb = decltype(kc)::value;
#pragma unroll
    for (int g = 0; g < 4; ++g) z[kb][g] = zl[kb * 256 + nat + 64 * g];
  });
#ifdef FINROM_SOLVE_CLOCKS
  ck[3] = wall_clock64();
#endif
  // ---- middle ----------------------------------------------------------------------------------------------------------------
  double part_q = 0.0;
  sfor<0, NB>([&](auto kc) {
    constexpr int kb = decltype(kc)::value;
#pragma unroll
    for (int g = 0; g < 4; ++g) part_q = fma(z[kb][g], __shfl(z[kb][g], q * 16), part_q);
  });
  part_q += __shfl_xor(part_q, 16);
  part_q += __shfl_xor(part_q, 32);                     // lane column c = o + 1: qoi_r[o]
  const double nanv = __builtin_nan("");
  if (info != nullptr && lane == 0) {
    if (grad && ga.info_store) info[s] = bad ? 2 : 0;
    else if (bad) atomicOr(&info[s], 2);
  }
  if (qoi_r != nullptr && lane >= 1 && lane <= p.n_obs) qoi_r[s * p.n_obs + lane - 1] = bad ? nanv : part_q;
  if (grad) {
    const bool ocol = c >= 1 && c <= p.n_obs;
    // (fm.on: the data come from the spare wave through LDS.  Two loads under a uniform branch, NOT one pointer that selects between
    //  the two address spaces: that would be a flat load)
    double dval = 0.0;
    if (fm.on) { if (ocol) dval = dsh[c - 1]; }
    else { if (ocol) dval = ga.data[(ga.data_stride ? s * ga.data_stride : 0) + c - 1]; }
    const double res = ocol ? dval - part_q : 0.0;
    double r2 = (q == 0) ? res * res : 0.0;
    for (int off = 8; off > 0; off >>= 1) r2 += __shfl_xor(r2, off);
    if (lane == 0) ga.J[s] = bad ? nanv : 0.5 * r2;
    // y2[row] = sum_o Z[row][o + 1] resid[o] -> column 1 of the tile column (column 0 keeps Z[:, 0])
    sfor<0, NB>([&](auto kc) {
      constexpr int kb = decltype(kc)::value;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        double t = z[kb][g] * res;
        for (int off = 8; off > 0; off >>= 1) t += __shfl_xor(t, off);
        if (c == 1) z[kb][g] = t;
      }
    });
  }
#ifdef FINROM_SOLVE_CLOCKS
  ck[4] = wall_clock64();
#endif
  // ---- backward: X_kb = M_kb^T (R_kb - sum_{j > kb} U(kb, j) X_j); X overwrites Z block by block ----------------------------------
  sfor<0, NB>([&](auto kc) {
    constexpr int kb = NB - 1 - decltype(kc)::value;
    v4d rr = z[kb];
    sfor<kb + 1, NB>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const double* ukj = tile_at(kb, j);
#pragma unroll
      for (int g = 0; g < 4; ++g) rr = mma(-ukj[trn + 4 * g], z[j][g], rr);      // A operand: U(kb, j)[i = c][k = q + 4 g]
    });
    v4d n = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 4; ++g) n = mma(mbuf[kb * 256 + nat + 64 * g], rr[g], n);   // A operand of M^T R: M[k = q + 4 g][i = c]
    z[kb] = n;
  });
#ifdef FINROM_SOLVE_CLOCKS
  ck[5] = wall_clock64();
  if (lane == 0 && qoi_r != nullptr) { for (int i = 0; i < 5; ++i) qoi_r[s * p.n_obs + i] = (double)(ck[i + 1] - ck[i]); qoi_r[s * p.n_obs + 5] = (double)ck[6]; }
  __builtin_amdgcn_s_waitcnt(0);
  if (lane == 10 && qoi_r != nullptr && fm.on) { qoi_r[s * p.n_obs + 6] = dsh[30]; qoi_r[s * p.n_obs + 7] = dsh[31]; }
#endif
  const int R = p.rp;
  if (c <= 1) {                                         // column 0 of X is w_r, column 1 v_r
    double* dst = grad ? ga.vw + s * (int64_t)(2 * R) + (c == 0 ? R : 0) : nullptr;
    sfor<0, NB>([&](auto kc) {
      constexpr int kb = decltype(kc)::value;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = 16 * kb + q + 4 * g;
        const double v = row < p.r ? (bad ? nanv : z[kb][g]) : 0.0;
        if (dst != nullptr) dst[row] = v;
        if (c == 0 && w_r != nullptr && row < p.r) w_r[s * p.r + row] = v;
      }
    });
  }
}

template <int NB>
int launch_small(const RomDev& p, const double* theta, int64_t S, int NC, double* part, int grad, const RomGradArgs& ga, double* w_r,
                 double* qoi_r, int* info, hipStream_t st, const MlpFuse* fuse) {
  constexpr int lds_a = 3 * onesample_nt<NB>() * 256 * (int)sizeof(double), lds_b = small_solve_lds<NB>() * (int)sizeof(double);
  constexpr int lds_b_mlp = small_solve_lds<NB>(true) * (int)sizeof(double);
  // the spare workgroup keeps the network's input (n_in floats) in the same dynamic LDS: a small basis' three partial triangles
  // can be smaller than that
  constexpr int lds_cap = lds_a > 96 * 1024 ? lds_a : 96 * 1024;
  MlpFuse fm{};
  if (fuse != nullptr && fuse->on) {
    if ((size_t)fuse->m.n_in * sizeof(float) > (size_t)lds_cap) { set_error("rom_onesample: error model input does not fit LDS"); return FINROM_ERR_UNSUPPORTED; }
    fm = *fuse;
  }
  const int lds_launch = fm.on && fm.m.n_in * (int)sizeof(float) > lds_a ? (fm.m.n_in * (int)sizeof(float) + 15) / 16 * 16 : lds_a;
  static PerDeviceOnce once;
  if (int rc = once.run([&]() -> int {
        FR_HIP(hipFuncSetAttribute((const void*)rom_small_proj_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_cap));
        FR_HIP(hipFuncSetAttribute((const void*)rom_small_solve_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b_mlp));
        return 0; })) return rc;
  {
    ScopedKernelTimer t(K_ROM_PROJ, st);
    hipLaunchKernelGGL(rom_small_proj_kernel<NB>, dim3((unsigned)S, (unsigned)(NC + (fm.on ? fm.nw0 : 0))), dim3(256), lds_launch, st, p, theta, S, NC, part,
                       p.kmeta, fm);
    FR_HIP(hipGetLastError());
  }
  ScopedKernelTimer t(K_ROM_SOLVE, st);
  if (fm.on && !grad) { set_error("rom_onesample: the error model rides with the adjoint solve only"); return FINROM_ERR_ARG; }
  hipLaunchKernelGGL(rom_small_solve_kernel<NB>, dim3((unsigned)S), dim3(fm.on ? 320 : 256), fm.on ? lds_b_mlp : lds_b, st, p, theta, S, NC, part, grad, ga, w_r, qoi_r, info, fm);
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// workgroups per sample of the contraction: as many as keep a part at >= 8 k-steps per wave, at most 8 (NC x 4 SIMDs of NC CUs).
// NOT a function of the batch size: a sample's sums are split the same way alone and inside a batch of 64 (bit-identical results)
int rom_onesample_parts(const RomDev& p, int64_t /*S*/) {
  static const int forced = getenv("FINROM_ONESAMPLE_NC") != nullptr ? atoi(getenv("FINROM_ONESAMPLE_NC")) : 0;
  if (forced > 0) return forced > 16 ? 16 : forced;
  int nc = p.nku / 32;
  if (nc > 8) nc = 8;
  return nc < 1 ? 1 : nc;
}
size_t rom_onesample_scratch_bytes(const RomDev& p, int64_t S) {
  return (size_t)S * rom_onesample_parts(p, S) * (p.NB * (p.NB + 1) / 2) * 256 * sizeof(double);
}
bool rom_onesample_applies(const RomDev& p, int64_t S) {
  static const bool off = getenv("FINROM_NO_ONESAMPLE") != nullptr;
  return !off && S <= ROM_SPLITK_MAX_S && p.NB >= 1 && p.NB <= 6 && p.nku >= 64 && p.n_obs <= 15;
}

// grad = 0: w_r (optional) and qoi_r; grad = 1 (ga.data, ga.vw, ga.J set): also v_r; v_r | w_r go to ga.vw for the contraction kernel
int launch_rom_onesample(const RomDev& p, const double* theta, int64_t S, double* part, int grad, const RomGradArgs& ga, double* w_r,
                         double* qoi_r, int* info, hipStream_t st, const MlpFuse* fuse) {
  if (S == 0) return 0;
  const int NC = rom_onesample_parts(p, S);
  switch (p.NB) {
#define FR_ONE(N) case N: return launch_small<N>(p, theta, S, NC, part, grad, ga, w_r, qoi_r, info, st, fuse);
    FR_ONE(1) FR_ONE(2) FR_ONE(3) FR_ONE(4) FR_ONE(5) FR_ONE(6)
#undef FR_ONE
    default: set_error("rom_onesample: basis size"); return FINROM_ERR_UNSUPPORTED;
  }
}

}  // namespace finrom
