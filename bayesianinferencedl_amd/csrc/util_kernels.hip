// Small helper kernels of the hot path: sub-fin averages, Gaussian-field sampler, difference.
#include "finrom_core.h"
#include <algorithm>
#include <type_traits>

namespace finrom {

typedef double d4 __attribute__((ext_vector_type(4)));

// theta[s][p] = sum_j Sop[p][j] * k[s][j]   (fom/forward_solve.py:466-480, rom :404-418:
// nine whole-mesh scalar assembles per sample in the reference; here one wave per sample)
constexpr int MAXP = 16;
// (round 4: SPW = 4 samples per wave share every value of Sop they load -- the operator is 9 x n doubles out of L2 per SAMPLE
// otherwise, 5.9 GB of L2 -> CU traffic per 20 000 fields at n = 4101 against 0.66 GB of fields: 0.50 -> ~0.2 ms.  Each sample's
// sums run in the same order as before: bit-identical.)
template <int MAXP, int SUBFIN_SPW>      // (<10, 4> for the fin's nine averages: ~200 VGPRs, two waves per SIMD; <16, 2> beyond)
__global__ __launch_bounds__(256) void subfin_avg_kernel(const double* __restrict__ Sop, int P, int n,
                                                         const double* __restrict__ k, int64_t S,
                                                         double* __restrict__ theta) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t s0 = ((int64_t)blockIdx.x * 4 + wave) * SUBFIN_SPW;
  if (s0 >= S) return;
  double acc[SUBFIN_SPW][MAXP];
#pragma unroll
  for (int q = 0; q < SUBFIN_SPW; ++q)
#pragma unroll
    for (int p = 0; p < MAXP; ++p) acc[q][p] = 0.0;
  // (four passes of 64 columns requested together -- one value of k per sample and P of S per pass -- and added in the order of the
  // columns: the first version waited for every pass on its own, 25 dependent trips to memory for a 1597-node field)
  constexpr int UP = 4;
  for (int j0 = lane; j0 < n; j0 += 64 * UP) {
    double kv[SUBFIN_SPW][UP], sv[MAXP][UP];
#pragma unroll
    for (int q = 0; q < SUBFIN_SPW; ++q)
#pragma unroll
      for (int u = 0; u < UP; ++u) kv[q][u] = (j0 + 64 * u < n && s0 + q < S) ? k[(s0 + q) * n + j0 + 64 * u] : 0.0;
#pragma unroll
    for (int p = 0; p < MAXP; ++p)
      if (p < P) {
#pragma unroll
        for (int u = 0; u < UP; ++u) sv[p][u] = j0 + 64 * u < n ? Sop[(int64_t)p * n + j0 + 64 * u] : 0.0;
      }
#pragma unroll
    for (int q = 0; q < SUBFIN_SPW; ++q)
#pragma unroll
      for (int u = 0; u < UP; ++u)
#pragma unroll
        for (int p = 0; p < MAXP; ++p)
          if (p < P && j0 + 64 * u < n) acc[q][p] = fma(sv[p][u], kv[q][u], acc[q][p]);
  }
#pragma unroll
  for (int q = 0; q < SUBFIN_SPW; ++q)
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      if (p < P && s0 + q < S) {                         // (wave-uniform: the unused sums are not reduced)
        double x = acc[q][p];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (lane == 0) theta[(s0 + q) * P + p] = x;
      }
    }
}

// The same product for parameter vectors (n = 5 / 9 conductivities per sample, the dataset loop's five- and nine-parameter forms):
// a wave per sample would spend its time reducing 64 lanes of which n are busy (0.2 ms per 100k samples at the head of the ROM
// half's critical path) -- here a thread owns a sample, S (P x n doubles) sits in LDS, sums run over j in index order.
constexpr int SUBFIN_SMALL_N = 16;
__global__ __launch_bounds__(256) void subfin_avg_small_kernel(const double* __restrict__ Sop, int P, int n,
                                                               const double* __restrict__ k, int64_t S,
                                                               double* __restrict__ theta) {
  __shared__ double sl[MAXP * SUBFIN_SMALL_N];
  for (int t = threadIdx.x; t < P * n; t += 256) sl[t] = Sop[t];
  __syncthreads();
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  double kv[SUBFIN_SMALL_N];
#pragma unroll
  for (int j = 0; j < SUBFIN_SMALL_N; ++j) kv[j] = j < n ? k[s * n + j] : 0.0;
  for (int p = 0; p < P; ++p) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < SUBFIN_SMALL_N; ++j)
      if (j < n) acc = fma(sl[p * n + j], kv[j], acc);
    theta[s * P + p] = acc;
  }
}

int launch_subfin_avg(const double* Sop, int P, int n, const double* k, int64_t S, double* theta, hipStream_t st) {
  if (S == 0) return 0;
  if (P > MAXP) { set_error("subfin_avg: P > 16"); return FINROM_ERR_UNSUPPORTED; }
  const auto wgs = [&](int spw) { return dim3((unsigned)((S + 4 * spw - 1) / (4 * spw))); };
  ScopedKernelTimer t(K_AVG, st);
  if (n <= SUBFIN_SMALL_N)
    hipLaunchKernelGGL(subfin_avg_small_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, Sop, P, n, k, S, theta);
  else
    if (P <= 10) hipLaunchKernelGGL((subfin_avg_kernel<10, 4>), wgs(4), dim3(256), 0, st, Sop, P, n, k, S, theta);
    else hipLaunchKernelGGL((subfin_avg_kernel<16, 2>), wgs(2), dim3(256), 0, st, Sop, P, n, k, S, theta);
  FR_HIP(hipGetLastError());
  return 0;
}

// k[s][j] = exp(0.5 * sum_{i<=j} xi[s][i] U[i][j])   (generate_fin_dataset.py:87-88; U upper)
// Small batches (S < SAMPLER_GEMM_MIN_S): fp64 MFMA GEMM, 64 x 64 output tile per workgroup, K-chunks of 16 through LDS; only the
// K-range i < j0+64 of the upper-triangular factor is visited.
__global__ __launch_bounds__(256) void sampler_small_kernel(const double* __restrict__ U, int n,
                                                            const double* __restrict__ xi, int64_t S,
                                                            double* __restrict__ kout) {
  __shared__ double Xs[64][17];
  __shared__ double Us[16][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t s0 = (int64_t)blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;
  const int q = lane >> 4, c = lane & 15;
  d4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
  const int kend = min(n, j0 + 64);
  for (int k0 = 0; k0 < kend; k0 += 16) {
    {
      const int kk = tid & 15, rr = tid >> 4;           // 16 x 16 threads -> 64 rows in 4 passes
#pragma unroll
      for (int pss = 0; pss < 4; ++pss) {
        const int row = rr + 16 * pss;
        const int64_t s = s0 + row;
        Xs[row][kk] = (s < S && k0 + kk < n) ? xi[s * n + k0 + kk] : 0.0;
      }
      const int cc = tid & 63, kr = tid >> 6;           // 64 cols x 4 rows -> 16 rows in 4 passes
#pragma unroll
      for (int pss = 0; pss < 4; ++pss) {
        const int kx = kr + 4 * pss;
        Us[kx][cc] = (k0 + kx < n && j0 + cc < n) ? U[(int64_t)(k0 + kx) * n + j0 + cc] : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const double a = Xs[16 * wave + c][4 * ks + q];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Us[4 * ks + q][16 * t + c], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t s = s0 + 16 * wave + q + 4 * g;
      const int j = j0 + 16 * t + c;
      if (s < S && j < n) kout[s * n + j] = exp(0.5 * acc[t][g]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The throughput form (round 4): C = Xi U as a blocked fp64-MFMA GEMM with the exp fused into the epilogue.
//  * Workgroup = 8 waves = one 256 (samples) x 128 (columns) output tile; wave (wm, wn) owns 64 x 64 of it = 4 x 4 MFMA tiles,
//    64 accumulator doubles per lane in ARCHITECTURAL VGPRs (inline-asm MFMA: the builtin's AGPR accumulators run at half rate,
//    rom_proj_device.h), two waves per SIMD.
//  * K in chunks of 16 through LDS, double-buffered, ONE `s_waitcnt lgkmcnt(0); s_barrier` per chunk; the operands of chunk c + 2
//    are requested into registers while chunk c computes (no wait on them at the barrier).  LDS holds the chunk in OPERAND
//    ORDER -- [k-step][16-row (col) tile][q][c] -- so a wave's operand fetch is one contiguous 512-B ds_read_b64: conflict-free.
//    Per k-step a wave issues 16 MFMAs on 8 operand doubles (the 64 x 64 tile's intensity: 2 MFMAs per LDS double).
//  * Workgroup tile traffic: (256 + 128) x K doubles per 256 x 128 x K multiply-adds -- 15 GB of L2 -> LDS traffic per 20 000
//    samples at n = 4101 (round 3's 64 x 64 tiles: 41 GB, of which 30 GB reached HBM).  What keeps most of THAT in the XCD's 4 MiB
//    L2: blocks b and b + 8 share an XCD (round-robin dispatch -- speed only, never correctness), and the 32 workgroups an XCD
//    runs together are one SUPER-TILE of 4 sample tiles x 8 adjacent column tiles (1024 x 1024 outputs) that walk K in lockstep
//    -- all 32 run to the super-tile's common K end (the top column tile's: U is zero below its diagonal, so the shorter tiles
//    add exact zeros; ~17 % more MFMAs than the triangle needs, paid so that the tiles stay within the L2 window of each other):
//    an Xi strip is then fetched from HBM once for 8 workgroups, a U strip once for 4.  Column groups are formed from the TOP
//    (the lone leftover is column tile 0, K = 128) and super-tiles are dealt to the XCDs in descending K.
// Same sums in the same order as sampler_small_kernel (k ascending in steps of four through the same instruction): bit-identical.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int SG_TS = 80;                         // doubles between two operand tiles in LDS (64 + 16: rows 16 apart land on other banks when staged)
// WM = wave rows of a workgroup: WM = 4 -> 256 x 128 tile, 8 waves, one workgroup per CU (120 KB of LDS); WM = 2 -> 128 x 128 tile,
// 4 waves, TWO workgroups per CU (2 x 80 KB): the two waves of a SIMD then belong to different workgroups, and one workgroup's
// per-chunk barrier and LDS hand-over are covered by the other's MFMAs.
template <int WM> struct SGeo {
  static constexpr int TM = 64 * WM, THREADS = 128 * WM;
  // doubles per k-step block of A: its row tiles + 64 B, so that the four k-steps a staging instruction writes together (4 lanes
  // per row: coalesced 128-B row reads) land on four different 16-bank groups -- without it every A write was a 4-way bank conflict
  // and the burst of them held up the operand reads (WM = 2 has no room for it beside a second workgroup)
  static constexpr int A_KS = 4 * WM * SG_TS + (WM == 4 ? 8 : 0);
  static constexpr int A_BUF = 4 * A_KS;                    // [k-step][row tile] per buffer
  static constexpr int B_BUF = 4 * 8 * SG_TS;               // [k-step][col tile]
  static constexpr size_t LDS_BYTES = (size_t)2 * (A_BUF + B_BUF) * sizeof(double);
  static constexpr int B_ITEMS = 1024 / THREADS;            // 2-column items (one 16-byte load / LDS write each) of the U chunk per thread
  static constexpr int SM_PER_SUPER = 1024 / TM;            // sample tiles of a super-tile (1024 samples x 8 column tiles per XCD)
};
constexpr int64_t SAMPLER_GEMM_MIN_S = 4096;      // below: the 64 x 64 kernel fills the chip better

#define SG_MFMA(ACC, A, B) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))

template <int WM, int NT>      // NT = this wave's live column tiles (4; fewer only in the last column tile of a factor whose n is not a multiple of 128)
__device__ __forceinline__ void sampler_tile(const double* __restrict__ U, int n, const double* __restrict__ xi, int64_t S,
                                             double* __restrict__ kout, int64_t s0, int j0, int nch, double* lds, int xflags) {
  typedef SGeo<WM> G;
  constexpr int SG_A_BUF = G::A_BUF, SG_B_BUF = G::B_BUF;
  double* Abuf = lds;
  double* Bbuf = lds + 2 * SG_A_BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int q = lane >> 4, c = lane & 15;
  // staging: A = two items of 4 consecutive k for one row (4 lanes cover a row's 128 B), B = one item of 4 consecutive columns
  const int a_row = tid >> 2, a_kq = tid & 3;
  // (a wave's 64 lanes take 128 consecutive columns of ONE row of U: 1 KB contiguous from memory, and in LDS eight lanes fill a
  //  column tile's 128-B row while the next eight land 640 B = 32 banks further: conflict-free 16-byte writes)
  const int b_k = tid >> 6, b_j = (tid & 63) * 2;            // (+ THREADS / 64 rows per further item)
  // Operand fetches are BUFFER loads (two 16-byte loads per item): rows beyond the batch / beyond the factor fall outside the
  // resource and come back as zeros without a branch; the chunk offset rides in an SGPR.  (k >= n inside a row reads the head of
  // the next row -- selected away below; columns j >= n of U read the next row of U and only feed output columns that are never
  // stored.)
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  typedef double d2_t __attribute__((ext_vector_type(2)));
  const int64_t rows_here = S - s0 < G::TM ? S - s0 : G::TM;
  const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(xi + s0 * n), 0, (int)(rows_here * n * 8), 0x00020000);
  const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(U + j0), 0, (int)(((int64_t)n * n - j0) * 8), 0x00020000);
  const int a_voff0 = (a_row * n + 4 * a_kq) * 8, a_voff1 = ((a_row + G::TM / 2) * n + 4 * a_kq) * 8;
  const int b_voff = (b_k * n + b_j) * 8;
  double ra[2][2][4], rb[2][G::B_ITEMS][2];              // two chunks in flight (register sets 0 / 1)
  auto gload = [&](int ch, auto setc) {
    constexpr int st = decltype(setc)::value;
    const int k0 = ch * 16;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int vo = it ? a_voff1 : a_voff0;
      const d2_t lo = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(ares, vo, k0 * 8, 0));
      const d2_t hi = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(ares, vo + 16, k0 * 8, 0));
      ra[st][it][0] = lo[0]; ra[st][it][1] = lo[1]; ra[st][it][2] = hi[0]; ra[st][it][3] = hi[1];
    }
#pragma unroll
    for (int it = 0; it < G::B_ITEMS; ++it) {
      const int so = (k0 + it * (G::THREADS / 64)) * n * 8;
      const d2_t lo = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(bres, b_voff, so, 0));
      rb[st][it][0] = lo[0]; rb[st][it][1] = lo[1];
    }
  };
  auto lwrite = [&](int buf, int ch, auto setc) {                     // (ch: the chunk register set `setc` holds)
    constexpr int st = decltype(setc)::value;
    const int kk = ch * 16 + 4 * a_kq;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = a_row + (G::TM / 2) * it;
      double* dst = Abuf + buf * SG_A_BUF + a_kq * G::A_KS + (row >> 4) * SG_TS + (row & 15);
      // (k >= n, the factor's last chunk only: the load wrapped into the next row -- zero it HERE, not where the load was issued: a
      //  select there would wait for the load it should let fly for a whole chunk; and only in that chunk: 16 selects per chunk were
      //  a tenth of the kernel -- every vector instruction costs the matrix pipe ~11 cycles)
      if (ch * 16 + 16 > n) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e * 16] = kk + e < n ? ra[st][it][e] : 0.0;      // e = q: the k inside the k-step
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e * 16] = ra[st][it][e];
      }
    }
#pragma unroll
    for (int it = 0; it < G::B_ITEMS; ++it) {
      const int bk = b_k + it * (G::THREADS / 64);
      double* dstb = Bbuf + buf * SG_B_BUF + ((bk >> 2) * 8 + (b_j >> 4)) * SG_TS + (bk & 3) * 16 + (b_j & 15);
      dstb[0] = rb[st][it][0]; dstb[1] = rb[st][it][1];
    }
  };
  auto exchange = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  d4 acc[4][NT > 0 ? NT : 1];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < (NT > 0 ? NT : 1); ++t) acc[i][t] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* Ard = Abuf + (4 * wm) * SG_TS + lane;
  const double* Brd = Bbuf + (4 * wn) * SG_TS + lane;

  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1> S1;
  gload(0, S0{});
  lwrite(0, 0, S0{});
  exchange();
  if (nch > 1) gload(1, S1{});                           // chunk c lives in register set c & 1 ...
  if (nch > 2) gload(2, S0{});                           // ... and TWO chunks are in flight: one chunk of MFMAs (1.7-3.4 us) did not
                                                         //     cover the fetch's latency under load
  // (two chunks per trip, the buffer a compile-time constant: every LDS address of the loop is base + immediate)
  auto chunk = [&](auto curc, int ch) {
    constexpr int cur = decltype(curc)::value;
    const double* Ac = Ard + cur * SG_A_BUF;
    const double* Bc = Brd + cur * SG_B_BUF;
    double av[2][4], bv[2][NT > 0 ? NT : 1];
    if constexpr (NT > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) av[0][i] = Ac[i * SG_TS];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[0][t] = Bc[t * SG_TS];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if constexpr (NT > 0) {
        if (ks < 3) {                                    // the next k-step's operands are on their way while this one's MFMAs issue
#pragma unroll
          for (int i = 0; i < 4; ++i) av[(ks + 1) & 1][i] = Ac[(ks + 1) * G::A_KS + i * SG_TS];
#pragma unroll
          for (int t = 0; t < NT; ++t) bv[(ks + 1) & 1][t] = Bc[((ks + 1) * 8 + t) * SG_TS];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < NT; ++t) SG_MFMA(acc[i][t], av[ks & 1][i], bv[ks & 1][t]);
      }
      if (ks == 1 && ch + 1 < nch && !(xflags & 2)) {    // chunk ch + 1 (requested an iteration ago) goes to the other buffer ...
        lwrite(cur ^ 1, ch + 1, std::integral_constant<int, cur ^ 1>{});
        if (ch + 3 < nch) gload(ch + 3, std::integral_constant<int, cur ^ 1>{});      // ... and its register set takes chunk ch + 3
      }
    }
    if (!(xflags & 1)) exchange();                       // (xflags: timing experiments, results are garbage -- FINROM_SAMPLER_XFLAGS)
  };
  int ch = 0;
  for (; ch + 1 < nch; ch += 2) {
    chunk(std::integral_constant<int, 0>{}, ch);
    chunk(std::integral_constant<int, 1>{}, ch + 1);
  }
  if (ch < nch) chunk(std::integral_constant<int, 0>{}, ch);
  if constexpr (NT > 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (i == 0 && t == 0) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc[i][t]));
        else asm volatile("" : "+v"(acc[i][t]));
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int64_t s = s0 + 64 * wm + 16 * i + q + 4 * g;
          const int j = j0 + 64 * wn + 16 * t + c;
          if (s < S && j < n) kout[s * n + j] = exp(0.5 * acc[i][t][g]);
        }
  }
}

template <int WM>
__global__ __launch_bounds__(128 * WM) void sampler_gemm_kernel(const double* __restrict__ U, int n, const double* __restrict__ xi,
                                                              int64_t S, double* __restrict__ kout, int n_sgroups, int n_cgroups,
                                                              int top_tile, int pad_k, int pad_min_k, int xflags) {
  extern __shared__ __attribute__((aligned(16))) double sg_lds[];
  typedef SGeo<WM> G;
  constexpr int WGS = 8 * G::SM_PER_SUPER;               // workgroups of a super-tile = what one XCD runs together (32 CUs)
  // blocks b and b + 8 share an XCD: XCD-local sequence number -> (super-tile, slot); the super-tiles (descending K) are dealt to
  // the XCDs boustrophedon -- 0..7, 7..0, ... -- so that no XCD always gets the longest of its round
  const int b = blockIdx.x, xcd = b & 7, seq = b >> 3;
  const int round = seq / WGS, slot = seq - round * WGS;
  const int super = round * 8 + ((round & 1) ? 7 - xcd : xcd);
  if (super >= n_sgroups * n_cgroups) return;
  const int cg = super / n_sgroups, sgp = super - cg * n_sgroups;        // descending K: column group 0 is the top one
  const int sm = slot >> 3, jn = slot & 7;
  const int top = top_tile - 8 * cg, jt = top - 7 + jn;
  if (jt < 0) return;
  const int64_t s0 = ((int64_t)sgp * G::SM_PER_SUPER + sm) * G::TM;
  if (s0 >= S) return;
  const int j0 = jt * 128;
  // lockstep (the whole super-tile to the top tile's K end) where the padding is cheap: column groups whose K end is >= 2048 -- below,
  // the 3.5 tiles of average padding are 20-40 % of the group's work and its strips are short anyway (6.81 -> 6.70 ms, +0.1 GB)
  const int kend = min(n, ((cg < pad_k && (top + 1) * 128 >= pad_min_k ? top : jt) + 1) * 128);
  const int nch = (kend + 15) / 16;
  const int live = (n - j0 - 64 * ((int)(threadIdx.x >> 6) & 1) + 15) / 16;     // this wave's column tiles that hold columns < n
  if (live >= 4) sampler_tile<WM, 4>(U, n, xi, S, kout, s0, j0, nch, sg_lds, xflags);
  else if (live == 3) sampler_tile<WM, 3>(U, n, xi, S, kout, s0, j0, nch, sg_lds, xflags);
  else if (live == 2) sampler_tile<WM, 2>(U, n, xi, S, kout, s0, j0, nch, sg_lds, xflags);
  else if (live == 1) sampler_tile<WM, 1>(U, n, xi, S, kout, s0, j0, nch, sg_lds, xflags);
  else sampler_tile<WM, 0>(U, n, xi, S, kout, s0, j0, nch, sg_lds, xflags);               // (stages and meets the barriers, no MFMAs)
}

template <int WM>
static int launch_sampler_gemm(const double* U, int n, const double* xi, int64_t S, double* k, hipStream_t st, bool pad) {
  typedef SGeo<WM> G;
  static PerDeviceOnce once;
  if (int rc = once.run([&]() -> int {
        FR_HIP(hipFuncSetAttribute((const void*)sampler_gemm_kernel<WM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
        return 0; })) return rc;
  const int ntn = (n + 127) / 128, n_cgroups = (ntn + 7) / 8;
  const int64_t n_sgroups = (S + 1023) / 1024, n_super = n_sgroups * n_cgroups;
  if (n_sgroups > (1 << 20)) { set_error("sampler: batch too large for one launch"); return FINROM_ERR_UNSUPPORTED; }
  const unsigned grid = (unsigned)((n_super + 7) / 8 * 8 * 8 * G::SM_PER_SUPER);
  hipLaunchKernelGGL(sampler_gemm_kernel<WM>, dim3(grid), dim3(G::THREADS), G::LDS_BYTES, st, U, n, xi, S, k, (int)n_sgroups, n_cgroups,
                     ntn - 1, pad ? (getenv("FINROM_SAMPLER_PAD_GROUPS") ? atoi(getenv("FINROM_SAMPLER_PAD_GROUPS")) : 1 << 20) : 0,
                     getenv("FINROM_SAMPLER_PAD_GROUPS") ? 0 : 2048,      // (A/B: the first N column groups in lockstep whatever their K)
                     getenv("FINROM_SAMPLER_XFLAGS") ? atoi(getenv("FINROM_SAMPLER_XFLAGS")) : 0);
  FR_HIP(hipGetLastError());
  return 0;
}

int launch_sampler(const double* U, int n, const double* xi, int64_t S, double* k, hipStream_t st) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_SAMPLER, st);
  const char* env_min = getenv("FINROM_SAMPLER_GEMM_MIN");                   // (tests: force one kernel or the other)
  const int64_t min_s = env_min ? atoll(env_min) : SAMPLER_GEMM_MIN_S;
  if (S < min_s) {
    dim3 grid((unsigned)((S + 63) / 64), (unsigned)((n + 63) / 64));
    hipLaunchKernelGGL(sampler_small_kernel, grid, dim3(256), 0, st, U, n, xi, S, k);
    FR_HIP(hipGetLastError());
    return 0;
  }
  const bool no_pad = getenv("FINROM_SAMPLER_NO_PAD") != nullptr;            // (experiment: every tile stops at its own K end)
  const char* env_wm = getenv("FINROM_SAMPLER_WM");                          // (A/B: 2 = 128-row tiles, two workgroups per CU)
  if (env_wm && atoi(env_wm) == 2) return launch_sampler_gemm<2>(U, n, xi, S, k, st, !no_pad);
  return launch_sampler_gemm<4>(U, n, xi, S, k, st, !no_pad);
}

// ---------------------------------------------------------------------------------------
// xi ~ N(0, I) on the device, keyed by the GLOBAL sample index (deep_learning/generate_fin_dataset.py:87 draws them with
// np.random.randn): Philox4x32-10, counter = (sample index lo, hi, pair index, 0), key = seed; the four output words make
// two 53-bit uniforms and Box-Muller turns them into the pair xi[2 jb], xi[2 jb + 1].  A sample's draw depends on nothing but
// (seed, its global index): any sharding of a dataset over GPUs produces the same fields.
// oracle/fin_oracle.py::philox_normal is the NumPy restatement the tests compare with.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__global__ __launch_bounds__(256) void philox_normal_kernel(unsigned long long seed, int64_t first, int64_t S, int n,
                                                            double* __restrict__ xi) {
  const int npair = (n + 1) / 2;
  const int64_t total = S * npair;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t sl = i / npair;
    const int jb = (int)(i - sl * npair);
    const unsigned long long g = (unsigned long long)(first + sl);
    unsigned o[4];
    philox4x32_10((unsigned)g, (unsigned)(g >> 32), (unsigned)jb, 0u, (unsigned)seed, (unsigned)(seed >> 32), o);
    const unsigned long long a = ((unsigned long long)o[1] << 32) | o[0], b = ((unsigned long long)o[3] << 32) | o[2];
    const double u1 = (double)((a >> 11) + 1) * 0x1.0p-53;       // (0, 1]
    const double u2 = (double)(b >> 11) * 0x1.0p-53;             // [0, 1)
    const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586476925286766559 * u2;
    xi[sl * n + 2 * jb] = rad * cos(ang);
    if (2 * jb + 1 < n) xi[sl * n + 2 * jb + 1] = rad * sin(ang);
  }
}

int launch_philox_normal(unsigned long long seed, int64_t first, int64_t S, int n, double* xi, hipStream_t st) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_MISC, st);
  const int64_t total = S * ((n + 1) / 2);
  const int64_t blocks = std::min<int64_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)blocks), dim3(256), 0, st, seed, first, S, n, xi);
  FR_HIP(hipGetLastError());
  return 0;
}

__global__ void sub_kernel(const double* __restrict__ a, const double* __restrict__ b, int64_t count,
                           double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = a[i] - b[i];
}

int launch_sub(const double* a, const double* b, int64_t count, double* out, hipStream_t st) {
  if (count == 0) return 0;
  ScopedKernelTimer t(K_MISC, st);
  int64_t blocks = (count + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sub_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, count, out);
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace finrom
