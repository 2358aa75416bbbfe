// ROM side of the hot path: batched least-squares Petrov-Galerkin reduced solve.
//
// Replaces AffineROMFin.forward_nine_param_reduced + .qoi_reduced
// (rom/averaged_affine_ROM.py:278-310, 323-333) for a batch of parameter vectors:
//     psi = (sum_p theta_p A_p + Bi M) Phi          :289-295  (assemble + AIJ x Dense)
//     A_r = psi^T psi ,  B_r = psi^T F              :296-297  (Dense^T x Dense, gemv)
//     w_r = solve(A_r, B_r)                         :304      (LAPACK dgesv in the reference)
//     qoi_r = (B_obs Phi) w_r                       :323-333
//
// rom_proj_kernel: one wave per sample.  psi is never materialised: each k-step builds a
// 4 x (16*NB) slab of psi in registers from the row-sparse tables Psi_p = A_p Phi (the
// reference's precomputed dA_dsigmak_phi, :215-220) and feeds it to v_mfma_f64_16x16x4_f64
// as BOTH operands (A = slab^T, B = slab hold the same lane values), accumulating only the
// upper block triangle of A_r (NB(NB+1)/2 tiles of 16x16).  Bound: fp64 MFMA issue.
#include "finrom_internal.h"

#include <cstdlib>
#include "rom_proj_device.h"
#include "mlp_device.h"

namespace finrom {

// ---------------------------------------------------------------------------------------
// Blocked Cholesky for bases that need more than one projection wave (96 < r <= 208): one wave per sample,
// in place on the packed A_r the projection kernel wrote.  Left-looking by block rows: the block row kb
// (<= 13 tiles, in registers) first receives the updates -U(kp,kb)^T U(kp,tj) of all previous block rows
// (tiles streamed back from memory, 4 MFMAs per tile), then is factored exactly like in chol_tiles and
// written back as rows of U (= columns of L in the packed layout the substitution kernel reads).
// ---------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(64, NB <= 8 ? 3 : 2) void rom_chol_blocked_kernel(RomDev p, double* __restrict__ Arp, int64_t S,
                                                              int* __restrict__ info) {
  const int lane = threadIdx.x, q = lane >> 4, c = lane & 15;
  const int64_t s = blockIdx.x;
  constexpr int R = 16 * NB;
  double* A = Arp + s * (int64_t)(R * (R + 1) / 2);
  auto at = [&](int row, int col) -> double& { return A[row * R - (row * (row - 1)) / 2 + col - row]; };   // row <= col
  int bad = 0;
#pragma unroll 1
  for (int kb = 0; kb < NB; ++kb) {
    d4 rowt[NB];
#pragma unroll
    for (int tj = 0; tj < NB; ++tj)
      if (tj >= kb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int rr = 16 * kb + q + 4 * g, cc = 16 * tj + c;
          double v = (cc >= rr) ? at(rr, cc) : at(cc, rr);          // the diagonal tile is mirrored from its upper half
          if (rr == cc && rr >= p.r) v = 1.0;                        // padding rows: unit diagonal
          rowt[tj][g] = v;
        }
      }
#pragma unroll 1
    for (int kp = 0; kp < kb; ++kp) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {       // operands of one k-step at a time: keeps the kernel free of AGPR spills
        const double neg = -at(16 * kp + q + 4 * g, 16 * kb + c);
        double ut[NB];
#pragma unroll
        for (int tj = 0; tj < NB; ++tj)
          if (tj >= kb) ut[tj] = at(16 * kp + q + 4 * g, 16 * tj + c);
#pragma unroll
        for (int tj = 0; tj < NB; ++tj)
          if (tj >= kb)
            asm volatile("s_nop 3\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(rowt[tj]) : "v"(neg), "v"(ut[tj]));
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) {          // drain: same-accumulator MFMAs follow in the next k-step
          if (tj == 0) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(rowt[tj]));
          else asm volatile("" : "+v"(rowt[tj]));
        }
      }
    }
    // factor the block row in registers (right-looking inside the 16 x 16(NB-kb) strip); the register index
    // gs of the pivot row is unrolled, its lane group qs is a run-time loop (4 code copies instead of 16)
#pragma unroll
    for (int gs = 0; gs < 4; ++gs)
#pragma unroll 1
    for (int qs = 0; qs < 4; ++qs) {
      const int st = 4 * gs + qs;
      double dtile[4];
#pragma unroll
      for (int tj = 0; tj < NB; ++tj)
        if (tj == kb) {
#pragma unroll
          for (int g = 0; g < 4; ++g) dtile[g] = rowt[tj][g];
        }
      const double piv = read_lane_f64(dtile[gs], qs * 16 + st);      // wave-uniform source lane
      bad |= !(piv > 0.0);
      double rinv = __builtin_amdgcn_rsq(piv);
#pragma unroll
      for (int it = 0; it < 2; ++it) rinv = rinv * fma(-0.5 * piv * rinv, rinv, 1.5);
      const double sc = (q == qs) ? rinv : 1.0;
      double m[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const double v = __shfl(dtile[gs] * sc, qs * 16 + ((q + 4 * g) & 15));
        m[g] = (q + 4 * g > st) ? v : 0.0;
      }
#pragma unroll
      for (int tj = 0; tj < NB; ++tj)
        if (tj >= kb) {
          rowt[tj][gs] *= sc;
          const double rv = __shfl(rowt[tj][gs], qs * 16 + c);
#pragma unroll
          for (int g = 0; g < 4; ++g) rowt[tj][g] = fma(-m[g], rv, rowt[tj][g]);
        }
    }
#pragma unroll
    for (int tj = 0; tj < NB; ++tj)
      if (tj >= kb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int rr = 16 * kb + q + 4 * g, cc = 16 * tj + c;
          if (cc >= rr) at(rr, cc) = rowt[tj][g];
        }
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next block rows read these tiles back
    __syncthreads();
  }
  if (bad && info != nullptr && lane == 0) atomicOr(&info[s], 2);
}

int launch_rom_chol_blocked(const RomDev& p, double* Ar, int64_t S, int* info, hipStream_t st) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_ROM_SOLVE, st);
  switch (p.NB) {
#define FR_CB(N) case N: hipLaunchKernelGGL(rom_chol_blocked_kernel<N>, dim3((unsigned)S), dim3(64), 0, st, p, Ar, S, info); break;
    FR_CB(7) FR_CB(8) FR_CB(9) FR_CB(10) FR_CB(11) FR_CB(12) FR_CB(13)
#undef FR_CB
    default: set_error("rom_chol_blocked: unsupported basis size"); return FINROM_ERR_UNSUPPORTED;
  }
  FR_HIP(hipGetLastError());
  return 0;
}

// rom_proj_kernel<NB, NW> (rom_proj_device.h); the eight-wave instantiations (r > 144) live in rom_proj_wide.hip so that the two
// halves compile side by side

// ---------------------------------------------------------------------------------------
// LDS-staged projection (bases up to r = 96, one wave per sample, 8 samples per workgroup).
// The Psi tables are cut into chunks of whole k-steps (<= 24 KiB, constant term count); a chunk is
// copied global -> LDS ONCE per workgroup with global_load_lds_dwordx4 (no VGPR round trip) while the
// previous chunk is being consumed, so the 8 waves share one table fetch and the fetch has a whole
// chunk of MFMA time (>10 us) to land -- which keeps this kernel MFMA-bound even while the FOM kernel
// saturates the CU's vector-memory pipeline on the other stream.
// LDS image of a chunk = its global bytes: [nks*NT*4 rows][rp] doubles, then nks*NT*4 theta indices.
// With rp = 80 the 4 row-groups of a ds_read_b64 wave access fall on disjoint banks (640 B row pitch).
// ---------------------------------------------------------------------------------------
constexpr int ROM_CHUNK_BYTES = 24 * 1024;

template <int NB>
__device__ __forceinline__ void lds_chunk_compute(const double* __restrict__ buf, int nks, int nt, int rp, const double* thw,
                                                  int q, int c, d4 (&acc)[NB * (NB + 1) / 2]) {
  // ONE copy of the MFMA group (runtime term count): several unrolled copies make hipcc spill the
  // inline-asm accumulators around every copy
  const int* pidx = reinterpret_cast<const int*>(buf + (size_t)nks * nt * 4 * rp);
  const int nrow = nks * nt;
  double v[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) v[b] = 0.0;
  int t = 0;
#pragma unroll 1
  for (int rowi = 0; rowi < nrow; ++rowi) {
    const int row = rowi * 4 + q;
    const double thp = thw[pidx[row]];
    const double* src = buf + (size_t)row * rp + c;
#pragma unroll
    for (int b = 0; b < NB; ++b) v[b] = fma(thp, src[16 * b], v[b]);
    if (++t == nt) {               // the slab of this k-step is complete
      mfma_tiles<NB, 1, 0>(v, acc);
      t = 0;
#pragma unroll
      for (int b = 0; b < NB; ++b) v[b] = 0.0;
    }
  }
}

template <int NB, int WPB>      // WPB waves (= samples) per workgroup share each staged chunk
__global__ __launch_bounds__(64 * WPB, WPB == 4 ? 3 : 1) void rom_proj_lds_kernel(RomDev p, const int* __restrict__ ch_nt,
                                                              const int* __restrict__ ch_nks,
                                                              const int* __restrict__ ch_off,
                                                              const int* __restrict__ ch_bytes,
                                                              const double* __restrict__ tvc,
                                                              const double* __restrict__ theta, int64_t S,
                                                              double* __restrict__ Ar, double* __restrict__ Br, int factor,
                                                              int* __restrict__ info) {
  constexpr int NT = NB * (NB + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [2][ROM_CHUNK_BYTES] + th[8][32]
  double* th = reinterpret_cast<double*>(smem + 2 * ROM_CHUNK_BYTES);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t s = (int64_t)blockIdx.x * WPB + wave;
  const bool live = s < S;
  trace_begin(p.trace, blockIdx.x);
  const long long probe_t0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_setprio(3);
  double* thw = th + wave * 32;
  if (lane == 0) thw[0] = 1.0;
  if (live && lane < p.P) thw[lane + 1] = theta[s * p.P + lane];
  const int q = lane >> 4, c = lane & 15;
  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};

  auto stage = [&](int g, int slot) {     // async copy of chunk g into LDS buffer `slot`, 1 KiB per wave-instruction
    const char* src = reinterpret_cast<const char*>(tvc) + (size_t)ch_off[g] * 8;
    char* dst = smem + slot * ROM_CHUNK_BYTES;
    const int nb = ch_bytes[g];
    for (int off = wave * 1024; off < nb; off += WPB * 1024)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + off), 16, 0, 0);
  };
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int g = 0; g < p.n_chunks; ++g) {
    if (g + 1 < p.n_chunks) stage(g + 1, (g + 1) & 1);
    const double* buf = reinterpret_cast<const double*>(smem + (g & 1) * ROM_CHUNK_BYTES);
    const int nks = ch_nks[g];
    lds_chunk_compute<NB>(buf, nks, ch_nt[g], p.rp, thw, q, c, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk g+1 has landed (this wave's share)
    __syncthreads();                                    // ... and everybody is done reading chunk g
  }
  if (!live) return;
  if (p.clock_probe && threadIdx.x == 0 && (blockIdx.x % 2500) == 7) {
    const long long dt = __builtin_amdgcn_s_memtime() - probe_t0, dr = __builtin_amdgcn_s_memrealtime() - probe_r0;
    printf("[proj probe] block %d: %lld shader ticks, %lld x10ns -> clock %.0f MHz, main loop %.1f us\n", (int)blockIdx.x, dt, dr,
           (double)dt / (double)dr * 100.0, dr * 0.01);
  }

  mfma_drain(acc);
  if (factor) {
    const int bad = chol_tiles<NB>(acc, q, c, p.r);
    if (bad && info != nullptr && lane == 0) atomicOr(&info[s], 2);
  }
  const int R = p.rp;
  double* A = Ar + s * (int64_t)(R * (R + 1) / 2);
  int idx = 0;
#pragma unroll
  for (int ti = 0; ti < NB; ++ti)
#pragma unroll
    for (int tj = ti; tj < NB; ++tj) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = 16 * ti + q + 4 * g, col = 16 * tj + c;
        if (ti != tj || col >= row) A[row * R - (row * (row - 1)) / 2 + col - row] = acc[idx][g];
      }
      ++idx;
    }
  // B_r = psi^T F over the root rows (rebuilt from the global table: a handful of k-steps, VALU only)
  double bacc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) bacc[b] = 0.0;
  for (int ks = 0; ks < p.rhs_nk; ++ks) {
    double v[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) v[b] = 0.0;
    for (int t = 0; t < p.rhs_nt; ++t) {
      const int row = (ks * p.rhs_nt + t) * 4 + q;
      const double thp = thw[p.rhs_pidx[row]];
      const double* src = p.rhs_tv + (int64_t)row * p.rp + c;
#pragma unroll
      for (int b = 0; b < NB; ++b) v[b] = fma(thp, src[16 * b], v[b]);
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
    if (q == 0) Br[s * p.rp + 16 * b + c] = x;
  }
  trace_end(p.trace, blockIdx.x);
}

template <int NB>
static int launch_proj_lds(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                           hipStream_t st) {
  const size_t lds = 2 * ROM_CHUNK_BYTES + 8 * 32 * sizeof(double);
  static const int wpb = getenv("FINROM_PROJ_LDS") ? atoi(getenv("FINROM_PROJ_LDS")) : 4;
  if (wpb == 8)
    hipLaunchKernelGGL((rom_proj_lds_kernel<NB, 8>), dim3((unsigned)((S + 7) / 8)), dim3(512), lds, st, p, p.ch_nt, p.ch_nks,
                       p.ch_off, p.ch_bytes, p.tvc, theta, S, Ar, Br, factor, info);
  else
    hipLaunchKernelGGL((rom_proj_lds_kernel<NB, 4>), dim3((unsigned)((S + 3) / 4)), dim3(256), lds, st, p, p.ch_nt, p.ch_nks,
                       p.ch_off, p.ch_bytes, p.tvc, theta, S, Ar, Br, factor, info);
  FR_HIP(hipGetLastError());
  return 0;
}

int launch_rom_proj(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                    hipStream_t st, double* w_r, double* qoi_r, int* cu_ticket) {
  // factor: 0 = write A_r, 1 = write its Cholesky factor (NB <= 6), 2 = also solve and write only w_r / qoi_r (NB <= 5)
  if (S == 0) return 0;
  ScopedKernelTimer t(K_ROM_PROJ, st);
  // The LDS-staged variant shares each table fetch among the 4 waves of a workgroup; measured on MI355X it
  // is 8 % slower stand-alone (barriers) and no faster beside the FOM kernel, so it is opt-in.
  if (p.n_chunks > 0 && getenv("FINROM_PROJ_LDS") != nullptr) {
    switch (p.NB) {
      case 1: return launch_proj_lds<1>(p, theta, S, Ar, Br, factor, info, st);
      case 2: return launch_proj_lds<2>(p, theta, S, Ar, Br, factor, info, st);
      case 3: return launch_proj_lds<3>(p, theta, S, Ar, Br, factor, info, st);
      case 4: return launch_proj_lds<4>(p, theta, S, Ar, Br, factor, info, st);
      case 5: return launch_proj_lds<5>(p, theta, S, Ar, Br, factor, info, st);
      default: break;
    }
  }
  // one-sample call patterns (MAP / HMC): a lone wave per sample is pure MFMA latency; split its k-steps over four waves
  if (rom_splitk_applies(p, S)) return launch_rom_proj_splitk(p, theta, S, Ar, Br, factor, info, st, w_r, qoi_r, RomGradArgs());
  dim3 block(256);
#define FR_CASE(N, W)                                                                              \
  case N: { constexpr int wpb = W > 4 ? W : 4; constexpr int spb = wpb / W;                        \
            hipLaunchKernelGGL((rom_proj_kernel<N, W>), dim3((unsigned)((S + spb - 1) / spb)),     \
                               dim3(64 * wpb), 0, st, p, theta, S, Ar, Br, factor, info, w_r, qoi_r, p.kmeta); } break;
  // bases whose last block is at most half full: 11 % (r = 120) / 7 % (r = 200) fewer MFMAs per k-step (HalfCover, rom_proj_device.h)
  const bool no_half = getenv("FINROM_PROJ_NO_HALF") != nullptr;      // (read per call: the test of the two forms flips it)
  if (p.NB >= 7 && p.r <= 16 * p.NB - 8 && !no_half) return launch_rom_proj_half(p, theta, S, Ar, Br, factor, info, st, w_r, qoi_r);
  switch (p.NB) {
    case 1: case 2: case 3: case 4: case 5:      // own translation unit (-O2)
      return launch_rom_proj_single(p, theta, S, Ar, Br, factor, info, st, w_r, qoi_r, cu_ticket);
    FR_CASE(6, 1)      // (r = 81..96 through the four-wave kernel with the fused solve: 28.6 vs 27.2 ms per 100k -- 21 tiles do not split evenly)
    FR_CASE(7, 4) FR_CASE(8, 4) FR_CASE(9, 4)
    case 10: case 11: case 12: case 13: return launch_rom_proj_wide(p, theta, S, Ar, Br, factor, info, st, w_r, qoi_r);
    //   // r > 144: 8 waves per sample so that a wave's tiles (and its share of the fused epilogue's extra columns) fit 256 VGPRs
    default:
      set_error("rom_proj: basis size " + std::to_string(p.r) + " > 208 not supported");
      return FINROM_ERR_UNSUPPORTED;
  }
#undef FR_CASE
  FR_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// reduced solve: one WAVE per sample (workgroup = 64 threads), lane = row of A_r.
// A_r arrives as the packed lower triangle, stored by columns (column k holds rows k..R-1
// contiguously, so a wave reading column k over its rows is conflict-free in LDS).
// Left-looking Cholesky in panels of 4 columns: per k the wave does 1 vector + 4 broadcast
// LDS reads for 4 (8 with the second row set) multiply-adds; pivots travel by lane shuffles.
// A_r is SPD; the reference calls LAPACK dgesv (rom :304) -- same solution.
// Rows r..R-1 are padding (zero rows of psi^T psi): they get a unit diagonal.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int col_start(int k, int R) { return k * R - (k * (k - 1)) / 2; }

template <bool IN_LDS, int NSET, bool FACTORED, bool GRAD = false>
__global__ __launch_bounds__(64) void rom_solve_kernel(RomDev p, const double* __restrict__ Arp,
                                                       const double* __restrict__ Br, int64_t S,
                                                       double* __restrict__ w_r, double* __restrict__ qoi_r,
                                                       double* __restrict__ Ar_out, double* __restrict__ Br_out,
                                                       int* __restrict__ info, RomGradArgs ga = RomGradArgs()) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int r = p.r, R = p.rp, np = R * (R + 1) / 2;
  // The factor lives in LDS when it fits (R <= 176); for larger bases (r = 200) it is factored
  // in place in the global scratch the projection kernel wrote.  Lane l owns rows l, l+64, ...
  // (NSET = ceil(R / 64) row sets).
  const int lane = threadIdx.x;
  const int64_t s = blockIdx.x;
  double* As = const_cast<double*>(Arp) + s * (int64_t)np;
  double* Lm;                                    // [np] packed lower triangle, column-major
  double* invd;                                  // [R]
  if constexpr (IN_LDS) { Lm = sm; invd = sm + np; } else { Lm = As; invd = sm; }
  double* xs = invd + R;                         // [R]
  if constexpr (IN_LDS) {   // np = R(R+1)/2 is even for R a multiple of 16: copy as 16-byte vectors, 8 loads in flight
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* src = reinterpret_cast<const d2*>(As);
    d2* dst = reinterpret_cast<d2*>(Lm);
    const int nv = np / 2;
    int t = lane;
    for (; t + 7 * 64 < nv; t += 8 * 64) {
      d2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[t + u * 64];
#pragma unroll
      for (int u = 0; u < 8; ++u) dst[t + u * 64] = v[u];
    }
    for (; t < nv; t += 64) dst[t] = src[t];
  }
  __syncthreads();
  if constexpr (!FACTORED)
    for (int i = r + lane; i < R; i += 64) Lm[col_start(i, R)] = 1.0;
  int row[NSET];
  bool ok[NSET];
  double b[NSET];
#pragma unroll
  for (int u = 0; u < NSET; ++u) {
    row[u] = lane + 64 * u;
    ok[u] = row[u] < R;
    b[u] = ok[u] ? Br[s * R + row[u]] : 0.0;
  }
  if (Ar_out != nullptr)
    for (int t = lane; t < r * r; t += 64) {
      const int i = t / r, j = t - i * r;
      const int hi = i > j ? i : j, lo = i > j ? j : i;
      Ar_out[s * (int64_t)r * r + t] = Lm[col_start(lo, R) + hi - lo];
    }
  if (Br_out != nullptr) {
#pragma unroll
    for (int u = 0; u < NSET; ++u)
      if (row[u] < r) Br_out[s * r + row[u]] = b[u];
  }
  __syncthreads();

  // value held by the lane that owns row `pr` (pr is wave-uniform)
  auto bcast = [&](const double (&x)[NSET], int pr) -> double {
    double out = 0.0;
#pragma unroll
    for (int u = 0; u < NSET; ++u)
      if ((pr >> 6) == u) out = read_lane_f64(x[u], pr & 63);      // wave-uniform source lane: v_readlane, no LDS round trip
    return out;
  };

  int bad = 0;
  if constexpr (FACTORED) {      // the projection kernel already factored A_r: Lm holds L = U^T
    for (int i = lane; i < R; i += 64) invd[i] = 1.0 / Lm[col_start(i, R)];
    __syncthreads();
  } else
  for (int j0 = 0; j0 < R; j0 += 4) {
    double a[4][NSET];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int pc = j0 + c;
#pragma unroll
      for (int u = 0; u < NSET; ++u) a[c][u] = (ok[u] && row[u] >= pc) ? Lm[col_start(pc, R) + row[u] - pc] : 0.0;
    }
#pragma unroll 2
    for (int k = 0; k < j0; ++k) {
      const int ck = col_start(k, R) - k;
      double l[NSET];
#pragma unroll
      for (int u = 0; u < NSET; ++u) l[u] = (ok[u] && row[u] >= j0) ? Lm[ck + row[u]] : 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double bc = Lm[ck + j0 + c];
#pragma unroll
        for (int u = 0; u < NSET; ++u) a[c][u] = fma(-l[u], bc, a[c][u]);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int pc = j0 + c;
      const double piv = bcast(a[c], pc);
      bad |= !(piv > 0.0);
      const double d = sqrt(piv), inv = 1.0 / d;
      double l[NSET];
      const int cp = col_start(pc, R) - pc;
#pragma unroll
      for (int u = 0; u < NSET; ++u) {
        l[u] = (row[u] > pc) ? a[c][u] * inv : ((row[u] == pc) ? d : 0.0);
        if (ok[u] && row[u] >= pc) Lm[cp + row[u]] = l[u];
      }
      if (lane == 0) invd[pc] = inv;
#pragma unroll
      for (int c2 = c + 1; c2 < 4; ++c2) {
        const double lp = bcast(l, j0 + c2);
#pragma unroll
        for (int u = 0; u < NSET; ++u) a[c2][u] = fma(-l[u], lp, a[c2][u]);
      }
    }
    __syncthreads();
  }
  // b <- A_r^{-1} b through the factor:  L y = b, then L^T x = y  (lane = row; pivots broadcast by shuffle)
  auto substitute = [&]() {
    for (int k = 0; k < R; ++k) {
      const double yk = bcast(b, k) * invd[k];
      const int ck = col_start(k, R) - k;
#pragma unroll
      for (int u = 0; u < NSET; ++u) {
        if (row[u] == k) b[u] = yk;
        else if (ok[u] && row[u] > k) b[u] = fma(-Lm[ck + row[u]], yk, b[u]);
      }
    }
    for (int k = R - 1; k >= 0; --k) {
      const double xk = bcast(b, k) * invd[k];
#pragma unroll
      for (int u = 0; u < NSET; ++u) {
        if (row[u] == k) b[u] = xk;
        else if (row[u] < k) b[u] = fma(-Lm[col_start(row[u], R) + k - row[u]], xk, b[u]);
      }
    }
  };
  substitute();
  const double nanv = __builtin_nan("");
#pragma unroll
  for (int u = 0; u < NSET; ++u)
    if (ok[u]) xs[row[u]] = b[u];
  __syncthreads();
  if (w_r != nullptr) {
#pragma unroll
    for (int u = 0; u < NSET; ++u)
      if (row[u] < r) w_r[s * r + row[u]] = bad ? nanv : b[u];
  }
  for (int o = lane; o < p.n_obs; o += 64) {
    double qv = 0.0;
    for (int t = 0; t < r; ++t) qv = fma(p.obs_phi[o * r + t], xs[t], qv);
    qoi_r[s * p.n_obs + o] = bad ? nanv : qv;
  }
  if (info != nullptr && lane == 0 && bad) atomicOr(&info[s], 2);

  if constexpr (GRAD) {
    // Adjoint gradient of J = 1/2 |data - B_obs Phi w_r|^2 with respect to the affine parameters, psi treated
    // as theta-independent exactly as the reference does (rom/averaged_affine_ROM.py:335-356):
    //   v_r = A_r^{-T} (B_obs Phi)^T (data - obs),   g_i = (psi v_r)^T (A_i Phi w_r) = sum_p theta_p v_r^T G_pi w_r,
    // G_pi = (A_p Phi)^T (A_i Phi) precomputed on the host (only region pairs that share nodes are non-zero).
    double* rs = xs + R;                           // [n_obs] residual, then v_r in vs
    double* vs = rs + 64;
    const double* dat = ga.data + (ga.data_stride ? s * ga.data_stride : 0);
    double jl = 0.0;
    for (int o = lane; o < p.n_obs; o += 64) {
      double qv = 0.0;
      for (int t = 0; t < r; ++t) qv = fma(p.obs_phi[o * r + t], xs[t], qv);
      const double res = dat[o] - qv;
      rs[o] = res;
      jl = fma(res, res, jl);
    }
    for (int off = 32; off > 0; off >>= 1) jl += __shfl_xor(jl, off);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
      double acc = 0.0;
      if (row[u] < r)
        for (int o = 0; o < p.n_obs; ++o) acc = fma(p.obs_phi[o * r + row[u]], rs[o], acc);
      b[u] = acc;
    }
    substitute();
#pragma unroll
    for (int u = 0; u < NSET; ++u)
      if (ok[u]) vs[row[u]] = b[u];
    __syncthreads();
    double g = 0.0;                                // lane i accumulates g_i
    for (int pi = 0; pi < ga.npairs; ++pi) {
      const double* Gt = ga.Gt + (int64_t)pi * r * r;     // stored column by column: Gt[c * r + row]
      double part = 0.0;
#pragma unroll
      for (int u = 0; u < NSET; ++u) {
        if (row[u] < r) {
          double t = 0.0;
          for (int c2 = 0; c2 < r; ++c2) t = fma(Gt[c2 * r + row[u]], xs[c2], t);
          part = fma(b[u], t, part);
        }
      }
      for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
      const int pp = ga.pair_p[pi];
      const double thp = pp == 0 ? 1.0 : ga.theta[s * p.P + pp - 1];
      if (lane == ga.pair_i[pi]) g = fma(thp, part, g);
    }
    if (lane < p.P) ga.g[s * p.P + lane] = bad ? nanv : g;
    if (lane == 0) ga.J[s] = bad ? nanv : 0.5 * jl;
  }
}

// ---------------------------------------------------------------------------------------
// Blocked substitutions for bases wider than 80 (the factor comes from the projection kernel's in-register Cholesky for
// r <= 96, from the blocked Cholesky kernel beyond) with the factor left in global memory: wave = sample, lane l owns
// rows l, l + 64, ...  Columns go in blocks of 16: the 16 x 16 diagonal tile is solved in registers (16 dependent steps,
// pivots passed by v_readlane), everything else is a throughput phase -- forward, the rows below the block subtract
// L[i][block] . y_block (16 independent, lane-coalesced column loads per row set); backward, the rows above the block
// subtract L[block][i] . x_block (column i is contiguous: one 128-byte segment per lane).  8 - 13 sequential block steps instead
// of 2 r pivot steps with an LDS round trip each, no LDS (so as many waves per CU as registers allow), the factor read twice.
// ---------------------------------------------------------------------------------------
// b <- (L L^T)^{-1} b for the rows this lane owns (row[u] = lane + 64 u)
template <int NSET>
__device__ __forceinline__ void subst_blocked(const double* __restrict__ L, const int (&row)[NSET], double (&b)[NSET], int R, int lane) {
  // ---- forward: L y = b ----------------------------------------------------------------
  for (int jb = 0; jb < R; jb += 16) {
    const int u0 = jb >> 6, l0 = jb & 63, li = lane - l0;      // lanes l0 .. l0 + 15 of row set u0 own the block's rows
    const bool mine = li >= 0 && li < 16;
    double t[16];                                               // row li of the diagonal tile: L[jb + li][jb + j], j <= li
#pragma unroll
    for (int j = 0; j < 16; ++j) t[j] = (mine && j <= li) ? L[col_start(jb + j, R) + li - j] : 0.0;
    double y[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      double bj = 0.0;
#pragma unroll
      for (int u = 0; u < NSET; ++u) if (u == u0) bj = b[u];
      const double yj = read_lane_f64(bj, l0 + j) / read_lane_f64(t[j], l0 + j);
      y[j] = yj;
#pragma unroll
      for (int u = 0; u < NSET; ++u)
        if (u == u0) b[u] = (mine && li == j) ? yj : ((mine && li > j) ? fma(-t[j], yj, b[u]) : b[u]);
    }
    // rows below the block
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
      const int i = row[u];
      if (i >= jb + 16 && i < R) {
        double lv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) lv[j] = L[col_start(jb + j, R) + i - (jb + j)];
        double acc = b[u];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = fma(-lv[j], y[j], acc);
        b[u] = acc;
      }
    }
  }
  // ---- backward: L^T x = y ---------------------------------------------------------------
  for (int jb = R - 16; jb >= 0; jb -= 16) {
    const int u0 = jb >> 6, l0 = jb & 63, li = lane - l0;
    const bool mine = li >= 0 && li < 16;
    double t[16];                                               // column li of the tile: L[jb + j][jb + li], j >= li (contiguous)
#pragma unroll
    for (int j = 0; j < 16; ++j) t[j] = (mine && j >= li) ? L[col_start(jb + li, R) + j - li] : 0.0;
    double x[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = 15 - jj;
      double bj = 0.0;
#pragma unroll
      for (int u = 0; u < NSET; ++u) if (u == u0) bj = b[u];
      const double xj = read_lane_f64(bj, l0 + j) / read_lane_f64(t[j], l0 + j);     // lane l0 + j holds L[jb+j][jb+j] in t[j]
      x[j] = xj;
#pragma unroll
      for (int u = 0; u < NSET; ++u)
        if (u == u0) b[u] = (mine && li == j) ? xj : ((mine && li < j) ? fma(-t[j], xj, b[u]) : b[u]);
    }
    // rows above the block: L[jb + j][i], j = 0..15, are 16 consecutive entries of column i
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
      const int i = row[u];
      if (i < jb) {
        const double* __restrict__ src = L + col_start(i, R) + jb - i;
        double lv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) lv[j] = src[j];
        double acc = b[u];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = fma(-lv[j], x[j], acc);
        b[u] = acc;
      }
    }
  }
}

template <int NSET, bool GRAD>
__global__ __launch_bounds__(64) void rom_subst_blocked_kernel(RomDev p, const double* __restrict__ Lp, const double* __restrict__ Br,
                                                               int64_t S, double* __restrict__ w_r, double* __restrict__ qoi_r,
                                                               RomGradArgs ga) {
  const int r = p.r, R = p.rp, np = R * (R + 1) / 2;
  const int lane = threadIdx.x;
  const int64_t s = blockIdx.x;
  const double* __restrict__ L = Lp + s * (int64_t)np;
  __shared__ double xs[256];
  __shared__ double rs[64];
  int row[NSET];
  double b[NSET];
#pragma unroll
  for (int u = 0; u < NSET; ++u) {
    row[u] = lane + 64 * u;
    b[u] = row[u] < R ? Br[s * R + row[u]] : 0.0;
  }
  subst_blocked<NSET>(L, row, b, R, lane);
#pragma unroll
  for (int u = 0; u < NSET; ++u) {
    if (row[u] < R) xs[row[u]] = b[u];
    if (w_r != nullptr && row[u] < r) w_r[s * r + row[u]] = b[u];
  }
  __syncthreads();
  double jl = 0.0;
  for (int o = lane; o < p.n_obs; o += 64) {
    double qv = 0.0;
    for (int tt = 0; tt < r; ++tt) qv = fma(p.obs_phi[o * r + tt], xs[tt], qv);
    if (qoi_r != nullptr) qoi_r[s * p.n_obs + o] = qv;
    if constexpr (GRAD) {
      const double res = ga.data[(ga.data_stride ? s * ga.data_stride : 0) + o] - qv;
      rs[o] = res;
      jl = fma(res, res, jl);
    }
  }
  if constexpr (GRAD) {
    // adjoint gradient, the same steps as the GRAD stage of rom_solve_kernel (rom/averaged_affine_ROM.py:335-356):
    //   v_r = A_r^{-T} (B_obs Phi)^T (data - obs),  g_i = sum_p theta_p v_r^T G_pi w_r
    for (int off = 32; off > 0; off >>= 1) jl += __shfl_xor(jl, off);
    __syncthreads();
    double v[NSET];
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
      double acc = 0.0;
      if (row[u] < r)
        for (int o = 0; o < p.n_obs; ++o) acc = fma(p.obs_phi[o * r + row[u]], rs[o], acc);
      v[u] = acc;
    }
    subst_blocked<NSET>(L, row, v, R, lane);
    if (lane == 0) ga.J[s] = 0.5 * jl;
    if (ga.vw != nullptr) {                        // the contraction runs batched on the matrix cores (rom_grad_contract_kernel)
      double* dst = ga.vw + s * (int64_t)(2 * R);
#pragma unroll
      for (int u = 0; u < NSET; ++u)
        if (row[u] < R) { dst[row[u]] = row[u] < r ? v[u] : 0.0; dst[R + row[u]] = row[u] < r ? xs[row[u]] : 0.0; }
      return;
    }
    double g = 0.0;                                // lane i accumulates g_i
    for (int pi = 0; pi < ga.npairs; ++pi) {
      const double* Gt = ga.Gt + (int64_t)pi * r * r;     // stored column by column: Gt[c * r + row]
      double part = 0.0;
#pragma unroll
      for (int u = 0; u < NSET; ++u) {
        if (row[u] < r) {
          double t2 = 0.0;
          for (int c2 = 0; c2 < r; ++c2) t2 = fma(Gt[c2 * r + row[u]], xs[c2], t2);
          part = fma(v[u], t2, part);
        }
      }
      for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
      const int pp = ga.pair_p[pi];
      const double thp = pp == 0 ? 1.0 : ga.theta[s * p.P + pp - 1];
      if (lane == ga.pair_i[pi]) g = fma(thp, part, g);
    }
    if (lane < p.P) ga.g[s * p.P + lane] = g;
  }
}

// g[s][i] = sum over the listed pairs (p, i) of theta_p[s] * v_s^T G_pi w_s for 16 samples per wave: T = V G_pi on the fp64
// matrix cores (A = V tile [16 samples x 4], B = G [4 x 16 columns]), then the row sums of T o W.  One wave reads every block
// G_pi once per 16 samples (per sample in the substitution kernel's own loop: 1.7 MB each at r = 80, the L2 -> L1 traffic that
// bounded it).  V and W tiles sit in LDS.
__global__ __launch_bounds__(64) void rom_grad_contract_kernel(RomDev p, int64_t S, RomGradArgs ga) {
  extern __shared__ __attribute__((aligned(16))) double cs[];
  const int r = p.r, R = p.rp;
  const int lane = threadIdx.x, q = lane >> 4, c = lane & 15;
  const int64_t s0 = (int64_t)blockIdx.x * 16;
  double* Vs = cs;                   // [16][R]
  double* Ws = cs + 16 * R;          // [16][R]
  double* gs = cs + 32 * R;          // [16][32]
  for (int t = lane; t < 16 * R; t += 64) {
    const int sl = t / R, col = t - sl * R;
    const bool ok = s0 + sl < S;
    Vs[t] = ok ? ga.vw[(s0 + sl) * (int64_t)(2 * R) + col] : 0.0;
    Ws[t] = ok ? ga.vw[(s0 + sl) * (int64_t)(2 * R) + R + col] : 0.0;
  }
  for (int t = lane; t < 16 * 32; t += 64) gs[t] = 0.0;
  __syncthreads();
  const int nct = (r + 15) / 16;
  for (int pi = 0; pi < ga.npairs; ++pi) {
    const double* __restrict__ G = ga.Gt + (int64_t)pi * r * r;      // G[k][j] = G[j * r + k]
    double part[4] = {0.0, 0.0, 0.0, 0.0};
    for (int ct = 0; ct < nct; ++ct) {
      const int col = 16 * ct + c;
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
      for (int k0 = 0; k0 < r; k0 += 4) {
        const int k = k0 + q;
        const double a = k < r ? Vs[c * R + k] : 0.0;                  // A[i = c][k = q]: sample c of the tile
        const double b = (k < r && col < r) ? G[(int64_t)col * r + k] : 0.0;   // B[k = q][j = c]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) part[g] = fma(acc[g], Ws[(q + 4 * g) * R + col], part[g]);   // D[row = q + 4g][col = c]; W is 0 beyond r
    }
    const int pp = ga.pair_p[pi], ii = ga.pair_i[pi];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double x = part[g];
      x += __shfl_xor(x, 8); x += __shfl_xor(x, 4); x += __shfl_xor(x, 2); x += __shfl_xor(x, 1);
      const int sl = q + 4 * g;
      if (c == 0 && s0 + sl < S) {
        const double thp = pp == 0 ? 1.0 : ga.theta[(s0 + sl) * p.P + pp - 1];
        gs[sl * 32 + ii] = fma(thp, x, gs[sl * 32 + ii]);
      }
    }
  }
  __syncthreads();
  for (int t = lane; t < 16 * p.P; t += 64) {
    const int sl = t / p.P, i = t - sl * p.P;
    if (s0 + sl < S) ga.g[(s0 + sl) * p.P + i] = gs[sl * 32 + i];
  }
}

// The same contraction for a handful of samples (one-sample call patterns): latency, not throughput -- ROM_GRAD_SMALL_NG
// workgroups per sample, each walks its share of the blocks G_pi as flat arrays (64 consecutive entries per step, 16 steps in
// flight) with v_r, w_r in LDS; the last workgroup to arrive (ticket) adds the partial sums in a fixed order.
__global__ __launch_bounds__(256) void rom_grad_contract_small_kernel(RomDev p, int64_t S, RomGradArgs ga, MlpBackFuse bf) {
  extern __shared__ __attribute__((aligned(16))) double cs[];
  __shared__ double wsum[4];
  __shared__ double gacc[32];
  const int r = p.r, R = p.rp, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, grp = blockIdx.y, NG = gridDim.y - (bf.on ? 1 : 0);
  const int64_t s = blockIdx.x;
  if (bf.on && grp == NG) {                              // the spare workgroup: the error model's walk back to its first layer (wave 0)
    __shared__ float bg[64], bup[64];
    if (tid < 64) {
      mlp_backward_hidden_wave(bf.m, s, bf.tape, bf.data, bf.data_stride, bf.qoi_r, bf.e_nn, bg, bup, tid);
      if (tid < bf.m.n_w) bf.g0_out[s * 64 + tid] = bg[tid];
    }
    return;
  }
  double* vs = cs; double* ws = cs + R;
  for (int t = tid; t < 2 * R; t += 256) cs[t] = ga.vw[s * (int64_t)(2 * R) + t];
  if (tid < 32) gacc[tid] = 0.0;
  __syncthreads();
  // (four waves per workgroup: a block G_pi is r^2 = 6561 dependent-free loads at r = 81 -- 26 per thread, 16 in flight, instead
  // of 103 per lane with one wave: the call is latency-bound)
  for (int pi = grp; pi < ga.npairs; pi += NG) {
    const double* __restrict__ G = ga.Gt + (int64_t)pi * r * r;      // G[c r + row]
    const int n = r * r;
    double part = 0.0;
    int c2 = 0, row = tid;                               // entry base + tid = c2 r + row
    while (row >= r) { row -= r; ++c2; }
    for (int base = 0; base < n; base += 256 * 16) {
      double gv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int idx = base + u * 256 + tid; gv[u] = idx < n ? G[idx] : 0.0; }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (base + u * 256 + tid < n) part = fma(gv[u] * vs[row], ws[c2], part);
        row += 256;
        while (row >= r) { row -= r; ++c2; }
      }
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) wsum[wv] = part;
    __syncthreads();
    if (tid == 0) {
      const int pp = ga.pair_p[pi];
      const double thp = pp == 0 ? 1.0 : ga.theta[s * p.P + pp - 1];
      gacc[ga.pair_i[pi]] = fma(thp, (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]), gacc[ga.pair_i[pi]]);
    }
    __syncthreads();
  }
  if (tid < 32) ga.gpart[(s * NG + grp) * 32 + tid] = gacc[tid];
  if (ga.defer_sum) return;                              // the next kernel in the stream adds the NG partial sums (no fence, no ticket)
  __threadfence();
  __syncthreads();
  __shared__ int last_s;
  if (tid == 0) last_s = atomicAdd(&ga.ticket[s], 1) == NG - 1;
  __syncthreads();
  if (!last_s) return;
  __threadfence();                                       // the other workgroups' partial sums are visible
  if (tid < p.P) {
    double t = 0.0;
    for (int w = 0; w < NG; ++w) t += ga.gpart[(s * NG + w) * 32 + tid];
    ga.g[s * p.P + tid] = t;
  }
  if (tid == 0) ga.ticket[s] = 0;                        // ready for the next call
}

int launch_rom_grad_contract_small(const RomDev& p, int64_t S, const RomGradArgs& ga, hipStream_t st, const MlpBackFuse* bf) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_ROM_SOLVE, st);
  MlpBackFuse b{};
  if (bf != nullptr && bf->on && ga.defer_sum) b = *bf;      // (only where the backward kernel follows in the same stream)
  hipLaunchKernelGGL(rom_grad_contract_small_kernel, dim3((unsigned)S, ROM_GRAD_SMALL_NG + (b.on ? 1 : 0)), dim3(256),
                     (size_t)2 * p.rp * sizeof(double), st, p, S, ga, b);
  FR_HIP(hipGetLastError());
  return 0;
}

int launch_rom_grad_contract(const RomDev& p, int64_t S, const RomGradArgs& ga, hipStream_t st) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_ROM_SOLVE, st);
  const size_t lds = ((size_t)32 * p.rp + 16 * 32) * sizeof(double);
  hipLaunchKernelGGL(rom_grad_contract_kernel, dim3((unsigned)((S + 15) / 16)), dim3(64), lds, st, p, S, ga);
  FR_HIP(hipGetLastError());
  return 0;
}

template <bool IN_LDS, int NSET, bool FACTORED>
static int launch_solve_t(const RomDev& p, size_t lds, const double* Ar, const double* Br, int64_t S, double* w_r,
                          double* qoi_r, double* Ar_out, double* Br_out, int* info, hipStream_t st) {
  static PerDeviceOnce once;                      // (per template instantiation, per device)
  if (lds > 64 * 1024)
    if (int rc = once.run([&]() -> int { FR_HIP(hipFuncSetAttribute((const void*)rom_solve_kernel<IN_LDS, NSET, FACTORED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64)); return 0; })) return rc;
  hipLaunchKernelGGL((rom_solve_kernel<IN_LDS, NSET, FACTORED>), dim3((unsigned)S), dim3(64), lds, st, p, Ar, Br, S, w_r, qoi_r, Ar_out, Br_out, info);
  FR_HIP(hipGetLastError());
  return 0;
}

template <bool IN_LDS, int NSET>
static int launch_grad_t(const RomDev& p, size_t lds, const double* Ar, const double* Br, int64_t S, double* w_r, double* qoi_r,
                         int* info, const RomGradArgs& ga, hipStream_t st) {
  static PerDeviceOnce once;
  if (lds > 64 * 1024)
    if (int rc = once.run([&]() -> int { FR_HIP(hipFuncSetAttribute((const void*)rom_solve_kernel<IN_LDS, NSET, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64)); return 0; })) return rc;
  hipLaunchKernelGGL((rom_solve_kernel<IN_LDS, NSET, true, true>), dim3((unsigned)S), dim3(64), lds, st, p, Ar, Br, S, w_r, qoi_r,
                     (double*)nullptr, (double*)nullptr, info, ga);
  FR_HIP(hipGetLastError());
  return 0;
}

int launch_rom_grad(const RomDev& p, const double* Ar, const double* Br, int64_t S, double* w_r, double* qoi_r,
                    int* info, const RomGradArgs& ga, hipStream_t st) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_ROM_SOLVE, st);
  const int nset = (p.rp + 63) / 64;
  static const bool old_subst = getenv("FINROM_OLD_SUBST") != nullptr;
  if (!old_subst && p.n_obs <= 64) {      // blocked substitutions with the factor in global memory, both solves + the gradient
    const dim3 grid((unsigned)S), block(64);
    switch (nset) {
      case 1: hipLaunchKernelGGL((rom_subst_blocked_kernel<1, true>), grid, block, 0, st, p, Ar, Br, S, w_r, qoi_r, ga); break;
      case 2: hipLaunchKernelGGL((rom_subst_blocked_kernel<2, true>), grid, block, 0, st, p, Ar, Br, S, w_r, qoi_r, ga); break;
      case 3: hipLaunchKernelGGL((rom_subst_blocked_kernel<3, true>), grid, block, 0, st, p, Ar, Br, S, w_r, qoi_r, ga); break;
      default: hipLaunchKernelGGL((rom_subst_blocked_kernel<4, true>), grid, block, 0, st, p, Ar, Br, S, w_r, qoi_r, ga); break;
    }
    FR_HIP(hipGetLastError());
    return 0;
  }
  const size_t lds = ((p.solve_in_lds ? (size_t)p.rp * (p.rp + 1) / 2 : 0) + 3 * (size_t)p.rp + 64) * sizeof(double);
  if (!p.solve_in_lds) return launch_grad_t<false, 4>(p, lds, Ar, Br, S, w_r, qoi_r, info, ga, st);   // r > 176: factor in global memory
  if (nset == 1) return launch_grad_t<true, 1>(p, lds, Ar, Br, S, w_r, qoi_r, info, ga, st);
  if (nset == 2) return launch_grad_t<true, 2>(p, lds, Ar, Br, S, w_r, qoi_r, info, ga, st);
  return launch_grad_t<true, 3>(p, lds, Ar, Br, S, w_r, qoi_r, info, ga, st);
}

int launch_rom_solve(const RomDev& p, const double* Ar, const double* Br, int64_t S, double* w_r, double* qoi_r,
                     double* Ar_out, double* Br_out, int* info, int factored, hipStream_t st) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_ROM_SOLVE, st);
  const bool in_lds = p.solve_in_lds;
  const size_t lds = ((in_lds ? (size_t)p.rp * (p.rp + 1) / 2 : 0) + 2 * (size_t)p.rp) * sizeof(double);
  const int nset = (p.rp + 63) / 64;
#define FR_SOLVE(L, N, F) return launch_solve_t<L, N, F>(p, lds, Ar, Br, S, w_r, qoi_r, Ar_out, Br_out, info, st)
  static const bool old_subst = getenv("FINROM_OLD_SUBST") != nullptr;
  if (factored && p.NB >= 6 && Ar_out == nullptr && Br_out == nullptr && !old_subst) {     // r > 80: blocked substitutions
    switch (nset) {
      case 2: hipLaunchKernelGGL((rom_subst_blocked_kernel<2, false>), dim3((unsigned)S), dim3(64), 0, st, p, Ar, Br, S, w_r, qoi_r, RomGradArgs()); break;
      case 3: hipLaunchKernelGGL((rom_subst_blocked_kernel<3, false>), dim3((unsigned)S), dim3(64), 0, st, p, Ar, Br, S, w_r, qoi_r, RomGradArgs()); break;
      default: hipLaunchKernelGGL((rom_subst_blocked_kernel<4, false>), dim3((unsigned)S), dim3(64), 0, st, p, Ar, Br, S, w_r, qoi_r, RomGradArgs()); break;
    }
    FR_HIP(hipGetLastError());
    return 0;
  }
  if (factored) {
    if (!in_lds) FR_SOLVE(false, 4, true);
    if (nset == 1) FR_SOLVE(true, 1, true);
    if (nset == 2) FR_SOLVE(true, 2, true);
    FR_SOLVE(true, 3, true);
  }
  if (p.solve_in_lds) {
    if (nset == 1) FR_SOLVE(true, 1, false);
    if (nset == 2) FR_SOLVE(true, 2, false);
    FR_SOLVE(true, 3, false);
  }
  FR_SOLVE(false, 4, false);
#undef FR_SOLVE
}

}  // namespace finrom
