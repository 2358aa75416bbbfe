// Device code of the projection kernels (psi build + psi^T psi on fp64 MFMA, in-register Cholesky and substitutions),
// shared by rom_kernels.hip and rom_proj_single.hip: the r = 80 instantiation lives in its own translation unit because it
// is compiled at -O2 (register budget, see _build.py), everything else at -O3.
#pragma once
#include "finrom_internal.h"
#include <type_traits>

namespace finrom {

typedef double d4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

// One k-step = 4 rows of psi.  Rows are sorted by their number of terms, so the k-steps form a
// few PHASES with a constant term count nt (<= 4): slot t of k-step ks is 4 padded r-vectors at
//   tv[((slot0 + (ks - ks0) * nt + t) * 4 + q) * rp + col],   theta index pidx[(slot) * 4 + q],
// i.e. every address is a function of the loop counter (no dependent index loads), and the raw
// table values of k-step ks+1 are fetched into registers while the MFMAs of k-step ks issue.
// (These phase tables serve psi^T F on the root rows and the FINROM_CLOCK_PROBE=3 A/B loop; the main loops run on the
// pattern-uniform tables RomDev::tvu / kmeta, see proj_main_uniform / proj_main_uniform_mw below.)
constexpr int ROM_MAX_NT = 4;

template <int NB>
__device__ __forceinline__ void load_kstep(const double* __restrict__ tv, const int* __restrict__ pidx, int slot, int nt,
                                           int rp, int q, int c, double (&raw)[ROM_MAX_NT][NB], int (&pi)[ROM_MAX_NT]) {
  // theta indices of ALL four possible terms first (slots beyond nt belong to the following k-steps or to the padding: valid
  // indices, their theta is read and not used): the LDS reads of theta can then be issued together, ahead of the table values
#pragma unroll
  for (int t = 0; t < ROM_MAX_NT; ++t) pi[t] = pidx[(slot + t) * 4 + q];
#pragma unroll
  for (int t = 0; t < ROM_MAX_NT; ++t) {
    if (t < nt) {                                  // wave-uniform
      const int row = (slot + t) * 4 + q;
      const double* src = tv + (int64_t)row * rp + c;
#pragma unroll
      for (int b = 0; b < NB; ++b) raw[t][b] = src[16 * b];
    }
  }
}

// The accumulators MUST live in architectural VGPRs: measured on gfx950 (tools/mfma_f64_variants.hip),
// v_mfma_f64_16x16x4_f64 issues every 64 cycles (77 TFLOP/s chip-wide) with VGPR accumulators but only
// every ~131 cycles (38 TFLOP/s) with AGPR accumulators, which is what hipcc picks for the builtin in a
// kernel of this size.  Hence inline asm with "+v" constraints; hipcc pads nothing around inline asm, so
// the VALU->MFMA operand hazard is covered by the s_nop in front of each k-step's MFMA group.
template <int NB, int NW, int W>
__device__ __forceinline__ void mfma_tiles(const double (&v)[NB], d4 (&acc)[(NB * (NB + 1) / 2 + NW - 1) / NW]) {
  // Bases wider than 128 spill accumulators; spill code next to inline asm is not hazard-safe
  // (the compiler cannot see that the asm is an MFMA), so those sizes use the compiler-managed builtin.
  constexpr bool kAsm = true;      // every instantiated (NB, NW) keeps its tiles in architectural VGPRs
  int idx = 0, mine = 0;
#pragma unroll
  for (int ti = 0; ti < NB; ++ti)
#pragma unroll
    for (int tj = ti; tj < NB; ++tj) {
      if (idx % NW == W) {
        // the wait states for "VALU wrote an operand -> MFMA reads it" sit INSIDE the asm statement: a
        // separate s_nop statement can be scheduled away from the MFMA it is meant to protect
        // (only the FIRST MFMA of the group can follow the VALU writes of the slab closely; the microbenchmark
        // tools/mfma_f64_mix.hip shows the pipe taking one MFMA per 64 cycles only when nothing sits between two of them)
        if constexpr (kAsm) {
          if (mine == 0) asm volatile("s_nop 7\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[mine]) : "v"(v[ti]), "v"(v[tj]));
          else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[mine]) : "v"(v[ti]), "v"(v[tj]));
        }
        else acc[mine] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[ti], v[tj], acc[mine], 0, 0, 0);
        ++mine;
      }
      ++idx;
    }
}

// Wait for the in-flight inline-asm MFMAs before the compiler-scheduled code reads their results: the
// wait carries every accumulator as an in/out operand, so no reader can be scheduled above it.
template <int N>
__device__ __forceinline__ void mfma_drain(d4 (&acc)[N]) {
#pragma unroll
  for (int t = 0; t < N; ++t) {
    if (t == 0) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc[t]));
    else asm volatile("" : "+v"(acc[t]));
  }
}

// ---------------------------------------------------------------------------------------
// In-register blocked Cholesky A_r = U^T U on the MFMA accumulator tiles (one wave, NW == 1).
// Tile (ti <= tj) holds rows 16ti.., columns 16tj.. in the C/D layout (lane: q = row&3 group, c = column;
// register g: row q + 4g).  Block row kb: 16 right-looking steps (pivot, scale, rank-1 update) with lane
// shuffles inside the block row; the update of the trailing tiles is 4 MFMAs per tile whose operands ARE
// the freshly computed registers of U (A[i][k] = U[k][i] lives exactly where the C/D layout put it).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double read_lane_f64(double v, int lane) {
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <int NB>
__device__ __forceinline__ int tile_index(int ti, int tj) { return ti * NB - (ti * (ti - 1)) / 2 + (tj - ti); }

template <int NB>
__device__ __forceinline__ int chol_tiles(d4 (&acc)[NB * (NB + 1) / 2], int q, int c, int r) {
  int bad = 0;
  // padding rows/columns (>= r) of psi^T psi are zero: give them a unit diagonal
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (q + 4 * g == c && 16 * t + c >= r) acc[tile_index<NB>(t, t)][g] = 1.0;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    const int dg = tile_index<NB>(kb, kb);
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int qs = st & 3, gs = st >> 2;
      const double piv = read_lane_f64(acc[dg][gs], qs * 16 + st);      // wave-uniform source lane: v_readlane, no LDS round trip
      bad |= !(piv > 0.0);
      double rinv = __builtin_amdgcn_rsq(piv);                 // 1/sqrt: hardware estimate + Newton steps
#pragma unroll
      for (int it = 0; it < 2; ++it) rinv = rinv * fma(-0.5 * piv * rinv, rinv, 1.5);
      const double sc = (q == qs) ? rinv : 1.0;
#pragma unroll
      for (int tj = kb; tj < NB; ++tj) acc[dg + (tj - kb)][gs] *= sc;     // row st of U is final
      double m[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const double v = __shfl(acc[dg][gs], qs * 16 + ((q + 4 * g) & 15));
        m[g] = (q + 4 * g > st) ? v : 0.0;                                // only rows below the pivot row
      }
#pragma unroll
      for (int tj = kb; tj < NB; ++tj) {
        const double rv = __shfl(acc[dg + (tj - kb)][gs], qs * 16 + c);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[dg + (tj - kb)][g] = fma(-m[g], rv, acc[dg + (tj - kb)][g]);
      }
    }
    // trailing update: T(ti,tj) -= U(kb,ti)^T U(kb,tj)
    if (kb + 1 < NB) {
      // operands are copied out of the accumulator tuples into plain 64-bit registers first (a sub-register
      // of a 256-bit inline-asm tuple is not guaranteed to be a legal, even-aligned MFMA source); the minus sign
      // of the update rides on the instruction's source-A negate modifier
      double pos[NB][4];
#pragma unroll
      for (int ti = kb + 1; ti < NB; ++ti)
#pragma unroll
        for (int g = 0; g < 4; ++g) pos[ti][g] = acc[dg + (ti - kb)][g];
      // k-step g outermost: consecutive MFMAs hit different tiles; the drain between k-steps covers the
      // MFMA -> same-accumulator MFMA hazard that hipcc cannot pad around inline asm
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int ti = kb + 1; ti < NB; ++ti)
#pragma unroll
          for (int tj = ti; tj < NB; ++tj)
            asm volatile("s_nop 3\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0 neg:[1,0,0]"
                         : "+v"(acc[tile_index<NB>(ti, tj)]) : "v"(pos[ti][g]), "v"(pos[tj][g]));
        mfma_drain(acc);
      }
    }
  }
  return bad;
}

// ---------------------------------------------------------------------------------------
// In-register substitutions U^T y = b, U x = y on the factored tiles (one wave, NW == 1), so that the default forward
// path needs neither the packed factor in memory nor a second kernel.  Vectors live as v[t] = value of index 16 t + c
// in lane column c, replicated over the four row groups q.  Row k = 16 ti + kr of U sits in row group kr & 3,
// register kr >> 2.  Forward, right-looking: every row group keeps the updates it produced in its own partial sums;
// the total for index k is collected with v_readlane.  Backward, row-oriented: the dot product of row k with the x
// already known is reduced over the 16 lanes of its row group by xor shuffles.  Block row ti and register g are
// unrolled, the row group qk of the step is a run-time loop (20 code copies at r = 80 instead of 80).
// ---------------------------------------------------------------------------------------

template <int NB>
__device__ __forceinline__ void solve_tiles(const d4 (&acc)[NB * (NB + 1) / 2], const double (&b)[NB],
                                            int q, int c, double (&x)[NB]) {
  // registers are tight (this kernel shares SIMDs with the FOM interpreter): b is folded into row group 0's partial sums,
  // and y is written into x, which the backward sweep overwrites entry by entry (an entry is read as y_k exactly once,
  // right before it becomes x_k; the dot products only touch entries that already are x)
  double part[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) { part[t] = (q == 0) ? b[t] : 0.0; x[t] = 0.0; }
#pragma unroll
  for (int ti = 0; ti < NB; ++ti)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll 1
      for (int qk = 0; qk < 4; ++qk) {
        const int kr = 4 * g + qk;
        double sum = read_lane_f64(part[ti], kr);
#pragma unroll
        for (int qq = 1; qq < 4; ++qq) sum += read_lane_f64(part[ti], qq * 16 + kr);
        const double yk = sum / read_lane_f64(acc[tile_index<NB>(ti, ti)][g], qk * 16 + kr);      // U_kk: row group qk, column kr
        if (c == kr) x[ti] = yk;
        const double m = (q == qk) ? yk : 0.0;
#pragma unroll
        for (int tj = ti; tj < NB; ++tj) part[tj] = fma(-acc[tile_index<NB>(ti, tj)][g], m, part[tj]);
      }
#pragma unroll
  for (int tr = 0; tr < NB; ++tr)
#pragma unroll
    for (int gr = 0; gr < 4; ++gr)
#pragma unroll 1
      for (int qk = 3; qk >= 0; --qk) {
        const int ti = NB - 1 - tr, g = 3 - gr;
        const int kr = 4 * g + qk;
        double d = (c > kr) ? acc[tile_index<NB>(ti, ti)][g] * x[ti] : 0.0;      // columns right of the diagonal only
#pragma unroll
        for (int tj = ti + 1; tj < NB; ++tj) d = fma(acc[tile_index<NB>(ti, tj)][g], x[tj], d);
        d += __shfl_xor(d, 8); d += __shfl_xor(d, 4); d += __shfl_xor(d, 2); d += __shfl_xor(d, 1);
        const double xk = (read_lane_f64(x[ti], kr) - read_lane_f64(d, qk * 16)) / read_lane_f64(acc[tile_index<NB>(ti, ti)][g], qk * 16 + kr);
        if (c == kr) x[ti] = xk;
      }
}

// ---------------------------------------------------------------------------------------
// Main loop of the single-wave kernels (NW == 1) on the pattern-uniform tables (RomDev::tvu / kpat).
// What limits this kernel is not memory (a run whose table fetches all hit one line is as fast) but VECTOR INSTRUCTIONS next
// to the matrix pipe: tools/mfma_f64_mix.hip -- one v_fma_f64 per v_mfma_f64_16x16x4_f64 costs 17 % of the MFMA rate at two
// waves per SIMD, an integer VALU op ~6 %, whichever wave issues it; LDS reads are free and loads nearly so.  The loop this
// one replaces issued 2.4 VALU instructions per MFMA (per-lane theta indices -> LDS addresses, 64-bit table addresses, one
// multiply-add per table value) and kept the pipe 69 % busy.  Here the four rows of a k-step share their theta indices, so
//   theta      comes from SGPRs (scalar loads of the sample's parameters, one k-step ahead),
//   addresses  are buffer loads: table descriptor + SGPR row offset + one constant per-lane offset,
// and the only vector arithmetic left is NB multiply(-add)s per TERM (1.4 terms per row at the fin): ~0.5 per MFMA.
// ---------------------------------------------------------------------------------------
// theta indices / values as wave-uniform values in SGPRs: the tables and the sample's parameters are read through CONSTANT
// address-space pointers (they are not written during the launch), which is what makes hipcc emit scalar loads and place
// the waits at the uses; through plain global pointers it never did.
typedef const int __attribute__((address_space(4)))* c_int_p;
typedef const double __attribute__((address_space(4)))* c_f64_p;
typedef int i4 __attribute__((ext_vector_type(4)));

template <int NB> constexpr int tile_ti(int i) { int ti = 0; while (i >= NB - ti) { i -= NB - ti; ++ti; } return ti; }
template <int NB> constexpr int tile_tj(int i) { int ti = 0; while (i >= NB - ti) { i -= NB - ti; ++ti; } return ti + i; }

// Interleaved form (the default): the burst of non-MFMA work of a k-step does not overlap with the partner wave's MFMAs as
// one would hope -- tools/mfma_f64_burst.hip: 15 MFMAs + a burst of 7 VALU, 40 SALU and 8 branches reach 75 % of the MFMA rate
// at two waves per SIMD, whatever the start offset or priority of the second wave -- so it is hidden behind this wave's OWN
// MFMAs: a 64-cycle MFMA leaves the issuing wave free, and scalar instructions, scalar / buffer loads placed between two MFMAs
// cost the pipe (almost) nothing (tools/mfma_f64_mix.hip).  One k-step = the MFMAs on slab vc, and in the gaps:
//   gap 0      the next slab vn = sum_t theta_t raw_t             (all vector arithmetic in ONE gap: the first VALU
//              instruction behind an MFMA costs ~15 cycles of pipe time, each further one ~4)
//   gaps 1..4  raw_t <- table rows of the k-step after next        (buffer loads; one k-step to land)
//   gap 5      scalar loads: thetas + record of the k-steps ahead  (constant address space; consumed in the next iteration)
// Two slab buffers, loop unrolled twice; per-k-step records (RomDev::kmeta) make every address a function of the loop
// counter.  Term counts are run-time, wave-uniform: scalar branches between the MFMAs.
template <int NB>
__device__ __forceinline__ void proj_main_uniform(const RomDev& p, const int* __restrict__ kmeta_g,
                                                  const double* __restrict__ theta_g, int q, int c,
                                                  d4 (&acc)[NB * (NB + 1) / 2], int nku) {
  constexpr int NTL = NB * (NB + 1) / 2;
  typedef const i4 __attribute__((address_space(4)))* c_i4_p;
  const c_i4_p kmeta = (c_i4_p)(unsigned long long)kmeta_g;             // record ks = kmeta[2 ks] (slot, nt, -, -), kmeta[2 ks + 1] (theta indices)
  const c_f64_p theta_s = (c_f64_p)(unsigned long long)theta_g;
  const __amdgpu_buffer_rsrc_t tres = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.tvu), 0, p.tvu_bytes, 0x00020000);
  const int voff = (q * p.rp + c) * 8;
  // bytes per slot (4 rows).  FINROM_CLOCK_PROBE=2 (timing experiment, results are garbage): every table fetch reads slot 0 --
  // same instructions, no L2 / fabric traffic: what the kernel loses beside the FOM sweep with and without its table stream
  const int rowb = p.clock_probe == 2 ? 0 : 4 * p.rp * 8;
  auto ldt = [&](int slot, auto bc) -> double {
    constexpr int b = decltype(bc)::value;
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(tres, voff + 128 * b, slot * rowb, 0));
  };
  auto theta_of = [&](int pp) -> double { return pp == 0 ? 1.0 : theta_s[pp - 1]; };     // index 0 = the constant 1
  auto load_raw = [&](double (&raw)[ROM_MAX_NT][NB], auto tc, int slot, int nt) {
    constexpr int t = decltype(tc)::value;
    if (t < nt) {
      asm volatile("" ::: "memory");                     // (keeps the branch a branch)
      sfor<0, NB>([&](auto bc) { raw[t][decltype(bc)::value] = ldt(slot + t, bc); });
    }
  };
  auto build = [&](double (&v)[NB], const double (&raw)[ROM_MAX_NT][NB], const double (&th)[ROM_MAX_NT], int nt) {
    sfor<0, NB>([&](auto bc) { constexpr int b = decltype(bc)::value; v[b] = th[0] * raw[0][b]; });
    sfor<1, ROM_MAX_NT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if (t < nt) {
        asm volatile("" ::: "memory");
        sfor<0, NB>([&](auto bc) { constexpr int b = decltype(bc)::value; v[b] = fma(th[t], raw[t][b], v[b]); });
      }
    });
  };
  double raw[ROM_MAX_NT][NB], th1[ROM_MAX_NT], va[NB], vb[NB];
  sfor<0, ROM_MAX_NT>([&](auto tc) { sfor<0, NB>([&](auto bc) { raw[decltype(tc)::value][decltype(bc)::value] = 0.0; }); });
  // prologue: slab of k-step 0, raw <- k-step 1, scalars of k-steps 1 and 2
  i4 m0 = kmeta[0], k0 = kmeta[1];
  sfor<0, ROM_MAX_NT>([&](auto tc) { load_raw(raw, tc, m0[0], m0[1]); });
  sfor<0, ROM_MAX_NT>([&](auto tc) { th1[decltype(tc)::value] = theta_of(k0[decltype(tc)::value]); });
  build(va, raw, th1, m0[1]);
  i4 m1 = kmeta[2], k1 = kmeta[3];
  sfor<0, ROM_MAX_NT>([&](auto tc) { load_raw(raw, tc, m1[0], m1[1]); });
  sfor<0, ROM_MAX_NT>([&](auto tc) { th1[decltype(tc)::value] = theta_of(k1[decltype(tc)::value]); });
  int nt1 = m1[1];
  i4 m2 = kmeta[4], k2 = kmeta[5];
  asm volatile("s_nop 7" ::: "memory");                  // va was written by VALU just now: VALU -> MFMA operand wait states

  auto step = [&](double (&vc)[NB], double (&vn)[NB], int ks) {
    double th2[ROM_MAX_NT];
    i4 m3, k3;
    sfor<0, NTL>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int ti = tile_ti<NB>(i), tj = tile_tj<NB>(i);
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(vc[ti]), "v"(vc[tj]));
      constexpr int last = NTL - 1;
      if constexpr (i == 0) build(vn, raw, th1, nt1);
      if constexpr (i == (1 < last ? 1 : last)) load_raw(raw, std::integral_constant<int, 0>{}, m2[0], m2[1]);
      if constexpr (i == (2 < last ? 2 : last)) load_raw(raw, std::integral_constant<int, 1>{}, m2[0], m2[1]);
      if constexpr (i == (3 < last ? 3 : last)) load_raw(raw, std::integral_constant<int, 2>{}, m2[0], m2[1]);
      if constexpr (i == (4 < last ? 4 : last)) load_raw(raw, std::integral_constant<int, 3>{}, m2[0], m2[1]);
      if constexpr (i == (5 < last ? 5 : last)) {
        sfor<0, ROM_MAX_NT>([&](auto tc) { th2[decltype(tc)::value] = theta_of(k2[decltype(tc)::value]); });
        m3 = kmeta[2 * (ks + 3)]; k3 = kmeta[2 * (ks + 3) + 1];
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    sfor<0, ROM_MAX_NT>([&](auto tc) { th1[decltype(tc)::value] = th2[decltype(tc)::value]; });
    nt1 = m2[1]; m2 = m3; k2 = k3;
  };
#pragma unroll 1
  for (int ks = 0; ks < nku; ks += 2) {
    step(va, vb, ks);
    if (ks + 1 < nku) step(vb, va, ks + 1);
  }
}

// ---------------------------------------------------------------------------------------
// The same loop on the GROUPED tables (RomDev::tvg / kmg, built in finrom_rom_create): most k-steps then need no vector
// arithmetic at all.  A row of psi in the interior of sub-domain d is theta_d T; accumulated DIVIDED by theta_d the slab is the
// table's rows as they are, and the buffer loads deliver the MFMA operands directly; rows on the Robin boundary of d add one
// multiply-add per block (T_d + (1 / theta_d) T_0).  Where the group changes the accumulators are multiplied by the ratio of the
// two scales squared (NT x 4 multiplies behind a drain: ten times per sample at nine sub-domains), after the last k-step by the
// last scale squared; the scalars come from the sample's row of RomDev::ext (rom_ext_kernel), read with scalar loads like theta
// above.  Measured at r = 80, 100k samples, beside the FOM sweep: 21.16 -> 20.04 ms per launch (the multiplies merely skipped in
// the loop above, garbage results: 20.94 -> 20.38).
// Three slab buffers rotate (loop unrolled three times; the host pads the list to a multiple of three): k-step ks multiplies x0,
// k-step ks + 1 is completed in place in x1 (gap 0), the first-term rows of k-step ks + 2 land in x2, its further terms in e.
template <int N>
__device__ __forceinline__ void mfma_fence(d4 (&acc)[N]) {      // VALU wrote the accumulators: nothing that reads them may move above, and the MFMA that follows keeps its wait states
#pragma unroll
  for (int t = 0; t < N; ++t) {
    if (t == 0) asm volatile("s_nop 7" : "+v"(acc[t]));
    else asm volatile("" : "+v"(acc[t]));
  }
}

template <int NB>
__device__ __forceinline__ void proj_main_grouped(const RomDev& p, const int* __restrict__ kmg_g, const double* __restrict__ ext_g,
                                                  int q, int c, d4 (&acc)[NB * (NB + 1) / 2]) {
  constexpr int NTL = NB * (NB + 1) / 2;
  typedef const i4 __attribute__((address_space(4)))* c_i4_p;
  const c_i4_p kmg = (c_i4_p)(unsigned long long)kmg_g;
  const c_f64_p ext_s = (c_f64_p)(unsigned long long)ext_g;
  const __amdgpu_buffer_rsrc_t tres = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.tvg), 0, p.tvg_bytes, 0x00020000);
  const int voff = (q * p.rp + c) * 8;
  const int rowb = 4 * p.rp * 8;
  auto ldt = [&](int slot, auto bc) -> double {
    constexpr int b = decltype(bc)::value;
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(tres, voff + 128 * b, slot * rowb, 0));
  };
  auto load_first = [&](double (&x)[NB], int slot) { sfor<0, NB>([&](auto bc) { x[decltype(bc)::value] = ldt(slot, bc); }); };
  auto load_extra = [&](double (&e)[ROM_MAX_NT - 1][NB], auto tc, int slot, int nt) {
    constexpr int t = decltype(tc)::value;
    if (t < nt) {
      asm volatile("" ::: "memory");                     // (keeps the branch a branch)
      sfor<0, NB>([&](auto bc) { e[t - 1][decltype(bc)::value] = ldt(slot + t, bc); });
    }
  };
  auto finish = [&](double (&x)[NB], const double (&e)[ROM_MAX_NT - 1][NB], const double (&cf)[ROM_MAX_NT], int nt, int flags) {
    if (!(flags & 1)) {
      asm volatile("" ::: "memory");
      sfor<0, NB>([&](auto bc) { constexpr int b = decltype(bc)::value; x[b] = cf[0] * x[b]; });
    }
    sfor<1, ROM_MAX_NT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if (t < nt) {
        asm volatile("" ::: "memory");
        sfor<0, NB>([&](auto bc) { constexpr int b = decltype(bc)::value; x[b] = fma(cf[t], e[t - 1][b], x[b]); });
      }
    });
  };
  auto rescale = [&](double f) {
    mfma_drain(acc);
    sfor<0, NTL>([&](auto ic) { constexpr int i = decltype(ic)::value; acc[i] = acc[i] * f; });
    mfma_fence(acc);
  };
  double xa[NB], xb[NB], xc[NB], e[ROM_MAX_NT - 1][NB], cf1[ROM_MAX_NT], f1;
  sfor<0, ROM_MAX_NT - 1>([&](auto tc) { sfor<0, NB>([&](auto bc) { e[decltype(tc)::value][decltype(bc)::value] = 0.0; }); });
  // prologue: k-step 0 complete in xa, the rows of k-step 1 on their way into xb / e, the scalars of k-steps 1 and 2
  i4 m0 = kmg[0], k0 = kmg[1];
  load_first(xa, m0[0]);
  sfor<1, ROM_MAX_NT>([&](auto tc) { load_extra(e, tc, m0[0], m0[1]); });
  sfor<0, ROM_MAX_NT>([&](auto tc) { cf1[decltype(tc)::value] = ext_s[k0[decltype(tc)::value]]; });
  finish(xa, e, cf1, m0[1], m0[2]);
  i4 m1 = kmg[2], k1 = kmg[3];
  load_first(xb, m1[0]);
  sfor<1, ROM_MAX_NT>([&](auto tc) { load_extra(e, tc, m1[0], m1[1]); });
  sfor<0, ROM_MAX_NT>([&](auto tc) { cf1[decltype(tc)::value] = ext_s[k1[decltype(tc)::value]]; });
  f1 = ext_s[m1[3]];
  int nt1 = m1[1], fl1 = m1[2];
  i4 m2 = kmg[4], k2 = kmg[5];
  asm volatile("s_nop 7" ::: "memory");                  // xa may have been written by VALU just now: VALU -> MFMA operand wait states

  auto step = [&](double (&x0)[NB], double (&x1)[NB], double (&x2)[NB], int ks) {
    double cf2[ROM_MAX_NT], f2;
    i4 m3, k3;
    sfor<0, NTL>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int ti = tile_ti<NB>(i), tj = tile_tj<NB>(i);
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x0[ti]), "v"(x0[tj]));
      constexpr int last = NTL - 1;
      if constexpr (i == 0) finish(x1, e, cf1, nt1, fl1);
      if constexpr (i == (1 < last ? 1 : last)) load_first(x2, m2[0]);
      if constexpr (i == (2 < last ? 2 : last)) load_extra(e, std::integral_constant<int, 1>{}, m2[0], m2[1]);
      if constexpr (i == (3 < last ? 3 : last)) load_extra(e, std::integral_constant<int, 2>{}, m2[0], m2[1]);
      if constexpr (i == (4 < last ? 4 : last)) load_extra(e, std::integral_constant<int, 3>{}, m2[0], m2[1]);
      if constexpr (i == (5 < last ? 5 : last)) {
        sfor<0, ROM_MAX_NT>([&](auto tc) { cf2[decltype(tc)::value] = ext_s[k2[decltype(tc)::value]]; });
        f2 = ext_s[m2[3]];
        m3 = kmg[2 * (ks + 3)]; k3 = kmg[2 * (ks + 3) + 1];
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (fl1 & 2) rescale(f1);                            // k-step ks + 1 opens a group: its scale from here on
    sfor<0, ROM_MAX_NT>([&](auto tc) { cf1[decltype(tc)::value] = cf2[decltype(tc)::value]; });
    f1 = f2; nt1 = m2[1]; fl1 = m2[2]; m2 = m3; k2 = k3;
  };
#pragma unroll 1
  for (int ks = 0; ks < p.nkg; ks += 3) {
    step(xa, xb, xc, ks);
    step(xb, xc, xa, ks + 1);
    step(xc, xa, xb, ks + 2);
  }
  if (p.ext_final) rescale(ext_s[p.ext_final]);
}

// ---------------------------------------------------------------------------------------
// Main loop of the multi-wave kernels (NW waves = one workgroup = one sample, r > 96) on the same pattern-uniform tables.
// Wave W owns the blocks b = W, W + NW, .. of the slab (builds them: 1/NW of the table loads and multiply-adds) and the
// tiles idx = W, W + NW, .. of the block triangle; the slab travels through LDS.  Same interleaving as above -- everything
// that is not an MFMA sits in the gaps between this wave's own MFMAs -- with the exchange software-pipelined over THREE LDS
// slab buffers so that one barrier per k-step is enough and nobody waits for a write it needs at once:
//   iteration ks:  MFMAs on vc (= slab ks, in registers), and in the gaps
//     gap 0   vn <- LDS slab ks + 1                     (complete since the barrier that ended iteration ks - 1)
//     gap 1   own blocks of slab ks + 2 = sum_t theta_t raw_t  -> LDS buffer (ks + 2) % 3   (last read in iteration ks - 2)
//     gap 2,3 raw[ks & 1] <- table rows of k-step ks + 4        (two k-steps to land: a k-step here is only ~9 MFMAs long)
//     gap 4   scalar loads: thetas of k-step ks + 4, record of k-step ks + 5
//   s_waitcnt lgkmcnt(0); s_barrier      (NOT __syncthreads(): that would also wait for the table loads just issued)
// ---------------------------------------------------------------------------------------
// Which wave builds which 16-column block of the slab.  The block triangle's NT tiles do not always divide by NW (r = 200: 91
// tiles over 8 waves = 3 x 12 + 5 x 11), and every wave waits for the slowest at the k-step's barrier: when the waves with one
// tile less can take ALL the slab building (<= 3 blocks each; a block costs about as much pipe time as a third of an MFMA), they
// do, and the waves with the extra tile build nothing.  Otherwise blocks go round robin (block b -> wave b % NW) as before.
template <int NB, int NW> struct SlabOwner {
  static constexpr int NT = NB * (NB + 1) / 2;
  static constexpr int tiles(int w) { return (NT - w + NW - 1) / NW; }
  static constexpr int nlight() { int n = 0; for (int w = 0; w < NW; ++w) n += tiles(w) == tiles(NW - 1); return n; }
  static constexpr bool light_only() { return NW == 8 && nlight() < NW && (NB + nlight() - 1) / nlight() <= 3; }      // (measured: r = 200 -2.5 %, r = 170 -1.4 %; with four waves, r = 136, +1.6 %: round robin there)
  static constexpr int owner(int b) { return light_only() ? NW - nlight() + b % nlight() : b % NW; }      // (light waves = the last nlight())
  static constexpr int count(int w) { int n = 0; for (int b = 0; b < NB; ++b) n += owner(b) == w; return n; }
  static constexpr int block(int w, int j) { for (int b = 0; b < NB; ++b) if (owner(b) == w && j-- == 0) return b; return 0; }
};

// ---------------------------------------------------------------------------------------
// Bases whose last 16-column block is at most half full (r mod 16 in 1..8: r = 120, 200): the block triangle pays for the empty
// half in every tile of its last column and, like every basis, for the redundant half of every diagonal tile.  In units of 8
// columns the last block is ONE unit u_last; the MFMA does not care which 16 columns sit on its two sides, so (HF = true):
//   * tile (i, NB-1), i < NB-1, becomes  block i  x  [u_last | unit 2i]: its left half is what it was (block i x u_last), its
//     right half is block i x unit 2i = the left half of the DIAGONAL tile (i, i), including that tile's lower-left block;
//   * what is then missing of the diagonal tiles are the 8 x 8 blocks (2i+1, 2i+1) and (u_last, u_last): two of them per
//     tile  [a | b] x [a | b]  -- so only every second diagonal slot issues an MFMA at all (NB / 2 of them stay empty, chosen
//     from the waves that own the most tiles: r = 120: 36 -> 32 MFMAs per k-step, 9 -> 8 per wave; r = 200: 91 -> 85, the three
//     waves that had 12 go to 11).
// Slots keep their owners (tile idx in wave idx % NW, as the epilogues expect); behind the main loop mw_half_fixup moves the
// pieces through LDS into the aligned tiles (the diagonal tile's upper-right block is its lower-left one read transposed).
// The slab in LDS is [block][row q][16 columns]: an operand of two arbitrary units is one LDS read with a per-lane offset.
template <int NB, int NW> struct HalfCover {
  static constexpr int NT = NB * (NB + 1) / 2;
  static constexpr int diag_idx(int i) { return i * NB - (i * (i - 1)) / 2; }
  static constexpr int loop_unit(int i) { return i == NB - 1 ? 2 * NB - 2 : 2 * i + 1; }      // the unit whose square tile (i, i) still needs
  struct Plan { int empty[NB]; int partner[NB]; };       // partner: of an empty slot its host, of a host the slot whose block it carries (-1: none)
  static constexpr Plan make() {
    Plan pl{};
    int load[NW] = {};
    for (int w = 0; w < NW; ++w) load[w] = (NT - w + NW - 1) / NW;
    for (int i = 0; i < NB; ++i) { pl.empty[i] = 0; pl.partner[i] = -1; }
    for (int e = 0; e < NB / 2; ++e) {
      int best = -1;
      for (int i = 0; i < NB; ++i)
        if (!pl.empty[i] && (best < 0 || load[diag_idx(i) % NW] > load[diag_idx(best) % NW])) best = i;
      pl.empty[best] = 1; --load[diag_idx(best) % NW];
    }
    int h = 0;
    for (int i = 0; i < NB; ++i)
      if (pl.empty[i]) { while (pl.empty[h] || pl.partner[h] >= 0) ++h; pl.partner[i] = h; pl.partner[h] = i; }
    return pl;
  }
  static constexpr Plan plan = make();
  static constexpr bool is_t1(int idx) { return tile_ti<NB>(idx) < NB - 1 && tile_tj<NB>(idx) == NB - 1; }
  static constexpr bool is_diag(int idx) { return tile_ti<NB>(idx) == tile_tj<NB>(idx); }
  static constexpr bool is_host(int idx) { return is_diag(idx) && !plan.empty[tile_ti<NB>(idx)]; }
  static constexpr bool is_empty(int idx) { return is_diag(idx) && plan.empty[tile_ti<NB>(idx)]; }
  static constexpr bool irregular(int idx) { return is_t1(idx) || is_host(idx); }
  static constexpr int irr_count(int w) { int n = 0; for (int idx = w; idx < NT; idx += NW) n += irregular(idx); return n; }
  static constexpr int irr_of_slot(int w, int i) { int n = 0; for (int idx = w; idx < w + i * NW; idx += NW) n += irregular(idx); return n; }
  static constexpr int slot_of_irr(int w, int k) { int i = 0; for (int idx = w; idx < NT; idx += NW, ++i) if (irregular(idx) && k-- == 0) return i; return 0; }
  // the two units of irregular operand k of wave w: lanes c < 8 / c >= 8
  static constexpr int irr_x(int w, int k) { const int idx = w + slot_of_irr(w, k) * NW; return is_t1(idx) ? 2 * NB - 2 : loop_unit(tile_ti<NB>(idx)); }
  static constexpr int irr_y(int w, int k) {
    const int idx = w + slot_of_irr(w, k) * NW, ti = tile_ti<NB>(idx);
    if (is_t1(idx)) return 2 * ti;
    return plan.partner[ti] >= 0 ? loop_unit(plan.partner[ti]) : loop_unit(ti);
  }
  static constexpr int mfmas(int w) { int n = 0; for (int idx = w; idx < NT; idx += NW) n += !is_empty(idx); return n; }
  static constexpr int lds_doubles() { return (NB - 1) * 128 + NB * 64; }
};

// the pieces of the diagonal tiles from where the HF main loop left them into the aligned tiles (all waves of the sample call it;
// lds: the slab buffers, free behind the loop's last barrier)
template <int NB, int NW, int W>
__device__ __forceinline__ void mw_half_fixup(d4 (&acc)[(NB * (NB + 1) / 2 + NW - 1) / NW], double* lds, int lane) {
  using HC = HalfCover<NB, NW>;
  constexpr int NT = NB * (NB + 1) / 2, MINE = (NT - W + NW - 1) / NW;
  const int q = lane >> 4, c = lane & 15;
  double* t1h = lds;                                     // [NB - 1][16 rows][8]: block i x unit 2i
  double* loops = lds + (NB - 1) * 128;                  // [NB][8][8]: the square of loop_unit(i)
  sfor<0, MINE>([&](auto ic) {
    constexpr int i = decltype(ic)::value, idx = W + i * NW, ti = tile_ti<NB>(idx);
    if constexpr (HC::is_t1(idx)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (c >= 8) { t1h[ti * 128 + (q + 4 * g) * 8 + (c - 8)] = acc[i][g]; acc[i][g] = 0.0; }      // what stays is the aligned tile (ti, NB - 1)
      }
    } else if constexpr (HC::is_host(idx)) {
      constexpr int pt = HC::plan.partner[ti];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        if (c < 8) loops[ti * 64 + (q + 4 * g) * 8 + c] = acc[i][g];
        if constexpr (pt >= 0) { if (c >= 8) loops[pt * 64 + (q + 4 * g) * 8 + (c - 8)] = acc[i][g + 2]; }
      }
    }
  });
  __syncthreads();
  sfor<0, MINE>([&](auto ic) {
    constexpr int i = decltype(ic)::value, idx = W + i * NW, ti = tile_ti<NB>(idx);
    if constexpr (HC::is_diag(idx)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = q + 4 * g;
        double v = 0.0;
        if constexpr (ti < NB - 1) {
          if (c < 8) v = t1h[ti * 128 + row * 8 + c];
          else if (row >= 8) v = loops[ti * 64 + (row - 8) * 8 + (c - 8)];
          else v = t1h[ti * 128 + c * 8 + row];           // (row, 8 + cc) = (8 + cc, row): c = 8 + cc
        } else {
          if (c < 8 && row < 8) v = loops[ti * 64 + row * 8 + c];
        }
        acc[i][g] = v;
      }
    }
  });
  __syncthreads();
}

template <int NB, int NW, int W, bool HF = false>
__device__ __forceinline__ void proj_main_uniform_mw(const RomDev& p, const int* __restrict__ kmeta_g,
                                                     const double* __restrict__ theta_g, int q, int c, int lane, double* slab,
                                                     d4 (&acc)[(NB * (NB + 1) / 2 + NW - 1) / NW]) {
  constexpr int NT = NB * (NB + 1) / 2;
  constexpr int MINE = (NT - W + NW - 1) / NW;           // tiles of this wave
  using SO = SlabOwner<NB, NW>;
  constexpr int NOWN0 = SO::count(W);                   // slab blocks of this wave (may be none)
  constexpr int NOWN = NOWN0 > 0 ? NOWN0 : 1;            // (array extents)
  static_assert(MINE >= 5, "five gaps are used");
  using HC = HalfCover<NB, NW>;
  constexpr int NI0 = HF ? HC::irr_count(W) : 0, NI = NI0 > 0 ? NI0 : 1;      // operands of two arbitrary units (HF)
  int ioff[NI];
  sfor<0, NI0>([&](auto kc) {
    constexpr int k = decltype(kc)::value, x = HC::irr_x(W, k), y = HC::irr_y(W, k);
    ioff[k] = c < 8 ? (x >> 1) * 64 + q * 16 + (x & 1) * 8 + c : (y >> 1) * 64 + q * 16 + (y & 1) * 8 + (c - 8);
  });
  typedef const i4 __attribute__((address_space(4)))* c_i4_p;
  const c_i4_p kmeta = (c_i4_p)(unsigned long long)kmeta_g;
  const c_f64_p theta_s = (c_f64_p)(unsigned long long)theta_g;
  const __amdgpu_buffer_rsrc_t tres = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.tvu), 0, p.tvu_bytes, 0x00020000);
  const int voff = (q * p.rp + c) * 8;
  const int rowb = 4 * p.rp * 8;
  auto theta_of = [&](int pp) -> double { return pp == 0 ? 1.0 : theta_s[pp - 1]; };
  auto thetas = [&](double (&th)[ROM_MAX_NT], const i4& k) { sfor<0, ROM_MAX_NT>([&](auto tc) { th[decltype(tc)::value] = theta_of(k[decltype(tc)::value]); }); };
  auto load_raw = [&](double (&raw)[ROM_MAX_NT][NOWN], auto tc, int slot, int nt) {
    constexpr int t = decltype(tc)::value;
    if (t < nt) {
      asm volatile("" ::: "memory");
      sfor<0, NOWN0>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        raw[t][j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(tres, voff + 128 * SO::block(W, j), (slot + t) * rowb, 0));
      });
    }
  };
  auto load_all = [&](double (&raw)[ROM_MAX_NT][NOWN], const i4& m) { sfor<0, ROM_MAX_NT>([&](auto tc) { load_raw(raw, tc, m[0], m[1]); }); };
  // own blocks of one slab -> LDS buffer at (double) offset o
  auto build_store = [&](const double (&raw)[ROM_MAX_NT][NOWN], const double (&th)[ROM_MAX_NT], int nt, int o) {
    double own[NOWN];
    sfor<0, NOWN0>([&](auto jc) { constexpr int j = decltype(jc)::value; own[j] = th[0] * raw[0][j]; });
    sfor<1, ROM_MAX_NT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if (t < nt) {
        asm volatile("" ::: "memory");
        sfor<0, NOWN0>([&](auto jc) { constexpr int j = decltype(jc)::value; own[j] = fma(th[t], raw[t][j], own[j]); });
      }
    });
    sfor<0, NOWN0>([&](auto jc) { constexpr int j = decltype(jc)::value; slab[o + SO::block(W, j) * 64 + lane] = own[j]; });
  };
  auto read_slab = [&](double (&v)[NB], double (&vi)[NI], int o) {
    sfor<0, NB>([&](auto bc) { constexpr int b = decltype(bc)::value; v[b] = slab[o + b * 64 + lane]; });
    sfor<0, NI0>([&](auto kc) { constexpr int k = decltype(kc)::value; vi[k] = slab[o + ioff[k]]; });
  };
  auto exchange = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  double rawa[ROM_MAX_NT][NOWN], rawb[ROM_MAX_NT][NOWN], thB[ROM_MAX_NT], thN[ROM_MAX_NT], va[NB], vb[NB], ia[NI], ib[NI];
  sfor<0, ROM_MAX_NT>([&](auto tc) { sfor<0, NOWN0>([&](auto jc) { rawa[decltype(tc)::value][decltype(jc)::value] = 0.0; rawb[decltype(tc)::value][decltype(jc)::value] = 0.0; }); });
  // prologue: slabs 0 and 1 into LDS, va <- slab 0, raw buffers <- k-steps 2 and 3, scalars up to k-step 4
  constexpr int SL = NB * 64;
  int o0 = 0, o1 = SL, o2 = 2 * SL;
  {
    const i4 m0 = kmeta[0], k0 = kmeta[1], m1 = kmeta[2], k1 = kmeta[3];
    load_all(rawa, m0); load_all(rawb, m1);
    thetas(thB, k0); thetas(thN, k1);
    build_store(rawa, thB, m0[1], o0);
    build_store(rawb, thN, m1[1], o1);
  }
  const i4 m2 = kmeta[4], k2 = kmeta[5], m3 = kmeta[6], k3 = kmeta[7];
  load_all(rawa, m2); load_all(rawb, m3);
  thetas(thB, k2); thetas(thN, k3);
  i4 m4 = kmeta[8], k4 = kmeta[9];
  exchange();
  read_slab(va, ia, o0);

  // One iteration, specialised by the term count NTS of the slab it BUILDS (k-step ks + 2).  Term counts fall along the
  // table, so the k-step it loads (ks + 4) has at most NTS terms: NTS rows are fetched (a fixed number of loads in flight:
  // the compiler can then wait for exactly the older buffer instead of vmcnt(0)), the build two iterations later uses its own.
  auto step = [&](auto ntc, double (&vc)[NB], double (&vn)[NB], double (&ic_)[NI], double (&in_)[NI], double (&raw)[ROM_MAX_NT][NOWN], int ks) {
    constexpr int NTS = decltype(ntc)::value;
    double thNN[ROM_MAX_NT];
    i4 m5, k5;
    sfor<0, MINE>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int idx = W + i * NW;
      constexpr int ti = tile_ti<NB>(idx), tj = tile_tj<NB>(idx);
      if constexpr (!HF) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(vc[ti]), "v"(vc[tj]));
      else if constexpr (HC::is_empty(idx)) {}            // (its blocks come out of other tiles: mw_half_fixup)
      else if constexpr (HC::is_host(idx)) { constexpr int k = HC::irr_of_slot(W, i); asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(ic_[k])); }
      else if constexpr (HC::is_t1(idx)) { constexpr int k = HC::irr_of_slot(W, i); asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(vc[ti]), "v"(ic_[k])); }
      else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(vc[ti]), "v"(vc[tj]));
      if constexpr (i == 0) read_slab(vn, in_, o1);
      if constexpr (i == 1) {
        double own[NOWN];
        sfor<0, NOWN0>([&](auto jc) { constexpr int j = decltype(jc)::value; own[j] = thB[0] * raw[0][j]; });
        sfor<1, NTS>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          sfor<0, NOWN0>([&](auto jc) { constexpr int j = decltype(jc)::value; own[j] = fma(thB[t], raw[t][j], own[j]); });
        });
        sfor<0, NOWN0>([&](auto jc) { constexpr int j = decltype(jc)::value; slab[o2 + SO::block(W, j) * 64 + lane] = own[j]; });
      }
      if constexpr (i == 2 || i == 3) {
        sfor<(i - 2) * 2, (i - 2) * 2 + 2>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          if constexpr (t < NTS)
            sfor<0, NOWN0>([&](auto jc) {
              constexpr int j = decltype(jc)::value;
              raw[t][j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(tres, voff + 128 * SO::block(W, j), (m4[0] + t) * rowb, 0));
            });
        });
      }
      if constexpr (i == 4) { thetas(thNN, k4); m5 = kmeta[2 * (ks + 5)]; k5 = kmeta[2 * (ks + 5) + 1]; }
      __builtin_amdgcn_sched_barrier(0);
    });
    sfor<0, ROM_MAX_NT>([&](auto tc) { constexpr int t = decltype(tc)::value; thB[t] = thN[t]; thN[t] = thNN[t]; });
    m4 = m5; k4 = k5;
    const int o = o0; o0 = o1; o1 = o2; o2 = o;
    exchange();
  };
  // iterations [0, uend[3] - 2) build four-term slabs, then three-, two-, one-term ones up to nku (ends are even)
  int ks = 0;
  sfor<0, ROM_MAX_NT>([&](auto pc) {
    constexpr int NTS = ROM_MAX_NT - decltype(pc)::value;
    const int end = NTS == 1 ? p.nku : p.uend[NTS - 1] - 2;
#pragma unroll 1
    for (; ks < end; ks += 2) {
      step(std::integral_constant<int, NTS>{}, va, vb, ia, ib, rawa, ks);
      step(std::integral_constant<int, NTS>{}, vb, va, ib, ia, rawb, ks + 1);
    }
  });
}

// ---------------------------------------------------------------------------------------
// Factorisation of a 16 x 16 SPD diagonal tile D = U^T U and M = U^-T, 4 x 4-blocked, on the matrix cores (round 3; it replaces
// 16 shuffle steps of ~75 instructions in the one-sample solve kernel and in fused_solve_mw).  D: C/D layout, destroyed.
// Returns M in C/D layout (lane (q, c), register g: M[4 g + q][c]); bad |= 1 where a pivot is not positive.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ d4 diag_tile_inverse(d4& D, int q, int c, int lane, int& bad) {
  auto mma_ = [](double a, double b, d4 cacc) -> d4 { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, cacc, 0, 0, 0); };
  // 4 x 4-blocked: in C/D layout block row b of the tile IS register b.  Per block: the 4 x 4 diagonal block S is read out
  // (ten v_readlanes), factored S = R^T R and inverted (Ib = R^-1) in wave-uniform arithmetic -- four 1/sqrt chains, no
  // cross-lane traffic -- and Ib is placed as an MFMA A operand (AI: lane (q, c < 4) = Ib[q][c]).  Then everything else
  // is the matrix cores': the block row of U (Ib^T D[b]: one MFMA), the rank-4 trailing update of the whole tile (ONE
  // MFMA, A = B = that row), and block row b of M = U^-T = L^-1 by block forward substitution (b + 1 MFMAs; the operand
  // U[j, b]^T is register j shifted along its rows: DPP row_shl, no LDS).  ~135 instructions per block against 4 x ~75.
  d4 Mc = (d4){0.0, 0.0, 0.0, 0.0};               // M, C/D layout: Mc[g] at lane (q, c) = M[4 g + q][c]
  double Ug[4];                                     // U, C/D layout
  const d4 zero4 = (d4){0.0, 0.0, 0.0, 0.0};
  sfor<0, 4>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    const double s00 = read_lane_f64(D[b], 0 * 16 + 4 * b + 0), s01 = read_lane_f64(D[b], 0 * 16 + 4 * b + 1),
                 s02 = read_lane_f64(D[b], 0 * 16 + 4 * b + 2), s03 = read_lane_f64(D[b], 0 * 16 + 4 * b + 3),
                 s11 = read_lane_f64(D[b], 1 * 16 + 4 * b + 1), s12 = read_lane_f64(D[b], 1 * 16 + 4 * b + 2),
                 s13 = read_lane_f64(D[b], 1 * 16 + 4 * b + 3), s22 = read_lane_f64(D[b], 2 * 16 + 4 * b + 2),
                 s23 = read_lane_f64(D[b], 2 * 16 + 4 * b + 3), s33 = read_lane_f64(D[b], 3 * 16 + 4 * b + 3);
    // 1/sqrt: the hardware estimate (relative error 5e-8 measured) and ONE third-order step, e = 1 - p r^2,
    // r <- r + r e (1/2 + 3/8 e): four dependent operations instead of the six of two Newton steps, and 1.4e-16 instead of
    // 2.4e-16 maximum relative error over 1e6 arguments in [1e-13, 1e13] (these chains are the tile's critical path)
    auto rsqrt2 = [&](double p_) {
      bad |= !(p_ > 0.0);
      const double r0 = __builtin_amdgcn_rsq(p_);
      const double e = fma(-p_ * r0, r0, 1.0);
      return fma(r0 * e, fma(0.375, e, 0.5), r0);
    };
    const double i00 = rsqrt2(s00);
    const double r01 = s01 * i00, r02 = s02 * i00, r03 = s03 * i00;
    const double i11 = rsqrt2(fma(-r01, r01, s11));
    const double r12 = fma(-r01, r02, s12) * i11, r13 = fma(-r01, r03, s13) * i11;
    const double i22 = rsqrt2(fma(-r12, r12, fma(-r02, r02, s22)));
    const double r23 = fma(-r12, r13, fma(-r02, r03, s23)) * i22;
    const double i33 = rsqrt2(fma(-r23, r23, fma(-r13, r13, fma(-r03, r03, s33))));
    const double i01 = -(r01 * i11) * i00, i12 = -(r12 * i22) * i11, i23 = -(r23 * i33) * i22;
    const double i02 = -fma(r01, i12, r02 * i22) * i00, i13 = -fma(r12, i23, r13 * i33) * i11;
    const double i03 = -fma(r01, i13, fma(r02, i23, r03 * i33)) * i00;
    double AI = 0.0;                                 // lane (q, c): Ib[q][c] for c < 4, q <= c
    AI = lane == 0 * 16 + 0 ? i00 : AI; AI = lane == 0 * 16 + 1 ? i01 : AI; AI = lane == 0 * 16 + 2 ? i02 : AI;
    AI = lane == 0 * 16 + 3 ? i03 : AI; AI = lane == 1 * 16 + 1 ? i11 : AI; AI = lane == 1 * 16 + 2 ? i12 : AI;
    AI = lane == 1 * 16 + 3 ? i13 : AI; AI = lane == 2 * 16 + 2 ? i22 : AI; AI = lane == 2 * 16 + 3 ? i23 : AI;
    AI = lane == 3 * 16 + 3 ? i33 : AI;
    // block row b of U = Ib^T D[b, :] (columns left of the block are not part of U)
    const d4 ur = mma_(AI, D[b], zero4);
    const double Ub = c >= 4 * b ? ur[0] : 0.0;
    Ug[b] = Ub;
    if constexpr (b < 3) D = mma_(c >= 4 * (b + 1) ? -Ub : 0.0, Ub, D);      // rows and columns behind the block: -= U[b, i]^T U[b, j]
    // block row b of M: R^-T (I[b, :] - sum_{j < b} U[j, b]^T M[j, :])
    d4 tm = (d4){c == 4 * b + q ? 1.0 : 0.0, 0.0, 0.0, 0.0};
    sfor<0, b>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const unsigned long long ub = __builtin_bit_cast(unsigned long long, Ug[j]);
      const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)ub, 0x100 + 4 * b, 0xf, 0xf, true);
      const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(ub >> 32), 0x100 + 4 * b, 0xf, 0xf, true);
      const double sh = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);      // lane (q, c) <- U[4 j + q][4 b + c]
      tm = mma_(c < 4 ? -sh : 0.0, Mc[j], tm);
    });
    Mc[b] = mma_(AI, tm[0], zero4)[0];
  });
  return Mc;
}

// ---------------------------------------------------------------------------------------
// Factorisation + reduced QoI for the multi-wave kernels WITHOUT leaving the registers (factor == 3: only qoi_r is wanted).
// The block triangle of A_r stays where the main loop accumulated it -- tile idx in wave idx % NW -- and LDS only carries what
// one block row hands to the others.  Right-looking by block rows kb:
//   (a) the owner of the diagonal tile factors it with the 16 row operations of chol_tiles, applied to [A_kk | I]:
//       that leaves U_kk and M = U_kk^-T (M A_kk = U_kk); M goes to LDS (transposed: the layout of an MFMA A operand);
//   (b) every tile right of it becomes U(kb, tj) = M A(kb, tj): 4 MFMAs instead of 16 shuffle steps; result -> LDS panel;
//   (c) every wave updates its own tiles T(ti, tj) -= U(kb, ti)^T U(kb, tj), operands straight from the LDS panel (an
//       accumulator tile in the C/D layout IS a valid A / B operand of the next product, register g = k-step g).
// Two barriers per block row.  The substitutions ride along as ROM_NXM extra tile columns G = [B_r | (B_obs Phi)^T]
// (1 + n_obs <= 16 ROM_NXM columns): the same operations turn them into Z = U^-T G, and
//   qoi_r = (B_obs Phi) U^-1 U^-T B_r = Z[:, 1:]^T Z[:, 0]      (rom :304, :323-333)
// needs no backward substitution, so U is never needed again: nothing but qoi_r leaves the kernel.
// ---------------------------------------------------------------------------------------
constexpr int ROM_NXM = 3;
template <int NB> constexpr int tidx(int ti, int tj) { return ti * NB - (ti * (ti - 1)) / 2 + (tj - ti); }
template <int NB> constexpr int fused_mw_lds_doubles() { return 256 + (NB + ROM_NXM) * 256 + 2 * 16 * NB + NB * 16 * ROM_NXM + 2; }

template <int NB, int NW, int W>
__device__ __forceinline__ void fused_solve_mw(const RomDev& p, d4 (&acc)[(NB * (NB + 1) / 2 + NW - 1) / NW], const double (&bacc)[NB],
                                               int64_t s, int lane, double* lds, double* __restrict__ qoi_r, int* __restrict__ info) {
  constexpr int NT = NB * (NB + 1) / 2, MINE = (NT - W + NW - 1) / NW;
  constexpr int NE = NB * ROM_NXM, EMINE = (NE - W + NW - 1) / NW;
  const int q = lane >> 4, c = lane & 15;
  double* minvT = lds;
  double* panel = lds + 256;
  double* bvec = panel + (NB + ROM_NXM) * 256;
  double* ybuf = bvec + 16 * NB;
  double* qpart = ybuf + 16 * NB;
  int* flag = (int*)(qpart + NB * 16 * ROM_NXM);
  const int nx = (p.n_obs + 1 + 15) / 16;
  if constexpr (W == 0) {
    if (q == 0) sfor<0, NB>([&](auto bc) { constexpr int b = decltype(bc)::value; bvec[16 * b + c] = bacc[b]; });
    if (lane == 0) *flag = 0;
  }
  __syncthreads();
  d4 ext[EMINE];
  sfor<0, EMINE>([&](auto ec) {
    constexpr int e = decltype(ec)::value, eidx = W + e * NW, kb = eidx / ROM_NXM, x = eidx % ROM_NXM;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = 16 * kb + q + 4 * g, col = 16 * x + c;
      double v = 0.0;
      if (x < nx) {
        if (col == 0) v = bvec[row];
        else if (col <= p.n_obs && row < p.r) v = p.obs_phi[(col - 1) * p.r + row];
      }
      ext[e][g] = v;
    }
  });
  // padding rows / columns (>= r) of psi^T psi are zero: unit diagonal
  sfor<0, MINE>([&](auto ic) {
    constexpr int i = decltype(ic)::value, idx = W + i * NW, ti = tile_ti<NB>(idx), tj = tile_tj<NB>(idx);
    if constexpr (ti == tj) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (q + 4 * g == c && 16 * ti + c >= p.r) acc[i][g] = 1.0;
    }
  });
  sfor<0, NB>([&](auto kc) {
    constexpr int kb = decltype(kc)::value;
    constexpr int dI = tidx<NB>(kb, kb);
    if constexpr (dI % NW == W) {                          // (a)
      int bad = 0;
      const d4 E = diag_tile_inverse(acc[dI / NW], q, c, lane, bad);      // (4 x 4-blocked, on the matrix cores; 16 shuffle steps before)
#pragma unroll
      for (int g = 0; g < 4; ++g) minvT[c * 16 + q + 4 * g] = E[g];        // E = M[row q + 4g][col c]
      if (bad && lane == 0) { atomicOr(flag, 1); if (info != nullptr) atomicOr(&info[s], 2); }
    }
    __syncthreads();
    {                                                     // (b)
      double Am[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) Am[g] = minvT[(q + 4 * g) * 16 + c];     // A operand of k-step g: M[m = c][k = q + 4g]
      auto solve_tile = [&](d4& T, int slot) {
        d4 n = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < 4; ++g) n = __builtin_amdgcn_mfma_f64_16x16x4f64(Am[g], T[g], n, 0, 0, 0);
        T = n;
#pragma unroll
        for (int g = 0; g < 4; ++g) panel[slot * 256 + g * 64 + lane] = n[g];
      };
      sfor<kb + 1, NB>([&](auto jc) {
        constexpr int tj = decltype(jc)::value, I = tidx<NB>(kb, tj);
        if constexpr (I % NW == W) solve_tile(acc[I / NW], tj);
      });
      sfor<0, ROM_NXM>([&](auto xc) {
        constexpr int x = decltype(xc)::value, e = kb * ROM_NXM + x;
        if constexpr (e % NW == W) { if (x < nx) solve_tile(ext[e / NW], NB + x); }
      });
    }
    __syncthreads();
    if constexpr (kb + 1 < NB) {                           // (c)
      auto update = [&](d4& T, int sa, int sb) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          T = __builtin_amdgcn_mfma_f64_16x16x4f64(-panel[sa * 256 + g * 64 + lane], panel[sb * 256 + g * 64 + lane], T, 0, 0, 0);
      };
      sfor<0, MINE>([&](auto ic) {
        constexpr int i = decltype(ic)::value, idx = W + i * NW, ti = tile_ti<NB>(idx), tj = tile_tj<NB>(idx);
        if constexpr (ti > kb) update(acc[i], ti, tj);
      });
      sfor<0, EMINE>([&](auto ec) {
        constexpr int e = decltype(ec)::value, eidx = W + e * NW, kbe = eidx / ROM_NXM, x = eidx % ROM_NXM;
        if constexpr (kbe > kb) { if (x < nx) update(ext[e], kbe, NB + x); }
      });
    }
  });
  // y = Z[:, 0] -> LDS; per block row the partial sums of Z[:, col]^T y; summed over the block rows in a fixed order
  sfor<0, EMINE>([&](auto ec) {
    constexpr int e = decltype(ec)::value, eidx = W + e * NW, kb = eidx / ROM_NXM, x = eidx % ROM_NXM;
    if constexpr (x == 0) {
      if (c == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) ybuf[16 * kb + q + 4 * g] = ext[e][g];
      }
    }
  });
  __syncthreads();
  sfor<0, EMINE>([&](auto ec) {
    constexpr int e = decltype(ec)::value, eidx = W + e * NW, kb = eidx / ROM_NXM, x = eidx % ROM_NXM;
    double part = 0.0;
    if (x < nx) {
#pragma unroll
      for (int g = 0; g < 4; ++g) part = fma(ext[e][g], ybuf[16 * kb + q + 4 * g], part);
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (q == 0) qpart[kb * (16 * ROM_NXM) + 16 * x + c] = part;
  });
  __syncthreads();
  if constexpr (W == 0) {
    if (qoi_r != nullptr && lane >= 1 && lane <= p.n_obs && lane < 16 * ROM_NXM) {
      double d = 0.0;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) d += qpart[kb * (16 * ROM_NXM) + lane];
      qoi_r[s * p.n_obs + lane - 1] = *flag ? __builtin_nan("") : d;
    }
  }
}

// ---------------------------------------------------------------------------------------
// The same idea for ONE wave (r <= 80, factor == 2, only qoi_r wanted -- what finrom_solve_pairs asks for): the epilogue of the
// single-wave kernel used to be chol_tiles (16 shuffle steps per block row over ALL its tiles) + solve_tiles (2 r dependent
// pivot steps): ~8.5 k vector instructions per sample, and every fp64 vector instruction also costs the partner wave's MFMAs
// pipe time.  Here, per block row kb:
//   (a) only the DIAGONAL tile goes through the 16 shuffle steps, applied to [A_kk | I]: leaves U_kk and M = U_kk^-T;
//   (b) U(kb, tj) = M A(kb, tj) for the tiles right of it and Z_kb = M G_kb for the extra column: 4 MFMAs per tile (M is
//       built transposed, which is its A-operand layout);
//   (c) trailing update T(ti, tj) -= U(kb, ti)^T U(kb, tj) and G_ti -= U(kb, ti)^T Z_kb on the matrix cores (operands = the
//       registers of U themselves, as in chol_tiles).
// G = [B_r | (B_obs Phi)^T] (1 + n_obs <= 16 columns: ONE extra tile column), Z = U^-T G, and
//   qoi_r = (B_obs Phi) U^-1 U^-T B_r = Z[:, 1:]^T Z[:, 0]      (rom :304, :323-333)
// needs no backward substitution.  ~2.4 k vector instructions + 180 MFMAs per sample at r = 80.
// ---------------------------------------------------------------------------------------
constexpr int ROM_SW_LDS = 16 * 5;      // doubles of LDS per wave (B_r)
#ifndef FINROM_SW_MFMA_DIAG_TILES
#define FINROM_SW_MFMA_DIAG_TILES 6     // block rows with at most this many live tiles factor their diagonal tile on the matrix cores
#endif
template <int NB>
__device__ __forceinline__ int fused_solve_sw(const RomDev& p, d4 (&acc)[NB * (NB + 1) / 2], const double (&bacc)[NB], int q, int c,
                                              double* __restrict__ mt, double& qout) {
  // The extra column is formed LEFT-looking -- G_kb -= sum_{j < kb} U(j, kb)^T Z_j when block row kb is reached -- so that at most
  // kb + 1 of its tiles are live beside the shrinking block triangle: 16 tiles at the peak instead of 20 (the kernel has to stay
  // at <= 200 VGPRs to share a SIMD with a wave of the FOM band sweep).
  d4 z[NB];
  const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.obs_phi), 0, p.n_obs * p.r * 8, 0x00020000);
  const int ovoff = ((c - 1) * p.r + q) * 8;             // row q of observation c - 1 (c == 0: far out of range)
  // B_r waits in LDS until its block row is reached: 10 registers the trailing updates need
  double* __restrict__ bl = mt;
  if (q == 0) sfor<0, NB>([&](auto bc) { constexpr int b = decltype(bc)::value; bl[16 * b + c] = bacc[b]; });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // padding rows / columns (>= r) of psi^T psi are zero: unit diagonal
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (q + 4 * g == c && 16 * t + c >= p.r) acc[tidx<NB>(t, t)][g] = 1.0;
  int bad = 0;
  sfor<0, NB>([&](auto kc) {
    constexpr int kb = decltype(kc)::value;
    // G_kb = [B_r | (B_obs Phi)^T | 0] rows of this block: column 0 = B_r, columns 1 .. n_obs = rows of B_obs Phi
    d4 e[1];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = 16 * kb + q + 4 * g;
      // (buffer load: one per-lane offset for the whole epilogue, the block row in the scalar offset; lane column 0 and the
      // columns beyond n_obs address outside the table and read 0)
      double v = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ores, ovoff, (16 * kb + 4 * g) * 8, 0));
      if (row >= p.r) v = 0.0;
      if (c == 0) v = bl[row];
      e[0][g] = v;
    }
    // ... minus what the block rows above contribute: a chain of 4 kb MFMAs on one accumulator (each waits for its
    // predecessor; the partner wave has the pipe meanwhile), its last wait behind the diagonal tile's factorisation
    sfor<0, kb>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (j > 0 || g > 0) mfma_drain(e);
        const double ua = acc[tidx<NB>(j, kb)][g], zb = z[j][g];
        asm volatile("s_nop 3\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0 neg:[1,0,0]" : "+v"(e[0]) : "v"(ua), "v"(zb));
      }
    });
    // (a) the 16 elimination steps on the diagonal tile; M^T is carried instead of M (COLUMN operations on F = E^T: lane
    // column c plays row c of E), because the C/D layout of M^T IS the A-operand layout of M -- F[g] at lane (q, c) =
    // M[c][4 g + q] -- so M needs no transposition through LDS (whose allocation would cost the band sweep, which shares the CU
    // and wants 37 KB per wave, a quarter of its waves)
    double Am[4];
    // (the first block rows of a wide basis keep the shuffle steps: with 15 or 10 tiles live, diag_tile_inverse's working set does
    //  not fit beside them under the kernel's 200-register cap -- hipcc spills, and spill code beside inline-asm MFMAs is not
    //  hazard-safe; with <= 6 live tiles it does)
#ifndef FINROM_SW_SHUFFLE_DIAG
    constexpr bool kMfmaDiag = (NB - kb) * (NB - kb + 1) / 2 <= FINROM_SW_MFMA_DIAG_TILES;
#else
    constexpr bool kMfmaDiag = false;
#endif
    if constexpr (kMfmaDiag) {
      // Round 4: the diagonal tile 4 x 4-blocked on the matrix cores (diag_tile_inverse, as the other two epilogues since round 3)
      // instead of the 16 shuffle steps below.  It returns M in the C/D layout; the A operand of M T is M^T in that layout --
      // and a tile in the C/D layout IS a valid A operand holding its own transpose (A[i = c][k = q] <- X[q + 4 g][c]), so
      // M^T = sum_g X_g^T E_g with the identity's rows as B operands: FOUR MFMAs transpose the tile, no LDS (2 KB per wave here
      // would cost the co-resident band sweep a quarter of its waves), no shuffles (16 of them spilled 140 registers).
      d4 D = acc[tidx<NB>(kb, kb)];
      const d4 Mc = diag_tile_inverse(D, q, c, q * 16 + c, bad);
      d4 MT = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int g = 0; g < 4; ++g) MT = __builtin_amdgcn_mfma_f64_16x16x4f64(Mc[g], c == q + 4 * g ? 1.0 : 0.0, MT, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) Am[g] = MT[g];
    } else {
      d4& D = acc[tidx<NB>(kb, kb)];
#pragma unroll
      for (int g = 0; g < 4; ++g) Am[g] = (q + 4 * g == c) ? 1.0 : 0.0;
#pragma unroll
      for (int gs = 0; gs < 4; ++gs)
#pragma unroll 1
        for (int qs = 0; qs < 4; ++qs) {
          const int st = 4 * gs + qs;
          const double piv = read_lane_f64(D[gs], qs * 16 + st);
          bad |= !(piv > 0.0);
          double rinv = __builtin_amdgcn_rsq(piv);
#pragma unroll
          for (int it = 0; it < 2; ++it) rinv = rinv * fma(-0.5 * piv * rinv, rinv, 1.5);
          D[gs] *= (q == qs) ? rinv : 1.0;                 // row st of U is final
          const double rvD = __shfl(D[gs], qs * 16 + c);   // U[st][c]
          const double mcol = (c > st) ? rvD : 0.0, scol = (c == st) ? rinv : 1.0;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (g <= gs) {                                 // column st of M^T is zero below row st
              Am[g] *= scol;
              Am[g] = fma(-mcol, __shfl(Am[g], q * 16 + st), Am[g]);
            }
            if (g >= gs) {                                 // rows below the pivot only (gs is unrolled)
              const double v = __shfl(D[gs], qs * 16 + ((q + 4 * g) & 15));
              const double m = (q + 4 * g > st) ? v : 0.0;
              D[g] = fma(-m, rvD, D[g]);
            }
          }
        }
    }
    if constexpr (kb > 0) mfma_drain(e);
    // (b) one tile at a time: four dependent MFMAs, the wait between them covers the same-accumulator hazard
    auto solve_tile = [&](d4& T) {
      double b[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) b[g] = T[g];             // plain 64-bit registers: legal MFMA sources (see chol_tiles)
      d4 n[1] = {(d4){0.0, 0.0, 0.0, 0.0}};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        asm volatile("s_nop 3\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(n[0]) : "v"(Am[g]), "v"(b[g]));
        mfma_drain(n);
      }
      T = n[0];
    };
    sfor<kb + 1, NB>([&](auto jc) { solve_tile(acc[tidx<NB>(kb, decltype(jc)::value)]); });
    solve_tile(e[0]);
    z[kb] = e[0];
    if constexpr (kb + 1 < NB) {                           // (c)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // (operands: element g of the tiles of block row kb, as 64-bit "v" inputs -- the compiler picks legal, even-aligned
        // pairs, and coalesces them with the tuples' sub-registers when it can; the kernel is register-bound)
#pragma unroll
        for (int ti = kb + 1; ti < NB; ++ti)
#pragma unroll
          for (int tj = ti; tj < NB; ++tj)
            asm volatile("s_nop 3\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0 neg:[1,0,0]"
                         : "+v"(acc[tidx<NB>(ti, tj)]) : "v"((double)acc[tidx<NB>(kb, ti)][g]), "v"((double)acc[tidx<NB>(kb, tj)][g]));
        // (wait for the in-flight updates; only the tiles being updated are named -- naming all of acc would keep the finished
        //  block rows' registers alive to the end of the epilogue)
        bool first = true;
#pragma unroll
        for (int ti = kb + 1; ti < NB; ++ti)
#pragma unroll
          for (int tj = ti; tj < NB; ++tj) {
            if (first) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc[tidx<NB>(ti, tj)]));
            else asm volatile("" : "+v"(acc[tidx<NB>(ti, tj)]));
            first = false;
          }
      }
    }
  });
  // qoi_r[o] = sum_rows Z[row][o + 1] Z[row][0]: column 0 of a row lives in lane (q, 0) of the row's group
  double part = 0.0;
  sfor<0, NB>([&](auto kc) {
    constexpr int kb = decltype(kc)::value;
#pragma unroll
    for (int g = 0; g < 4; ++g) part = fma(z[kb][g], __shfl(z[kb][g], q * 16), part);
  });
  part += __shfl_xor(part, 16);
  part += __shfl_xor(part, 32);
  qout = part;                                             // lane c = o + 1 (any row group): qoi_r[o]
  return bad;
}

template <int NB, int NW, int W, bool SK = false, bool GR = false, bool HF = false>
__device__ __forceinline__ void rom_proj_body(const RomDev& p, const double* thw, int64_t s, int lane,
                                              double* __restrict__ Ar, double* __restrict__ Br, int factor,
                                              int* __restrict__ info, double* __restrict__ w_r = nullptr,
                                              double* __restrict__ qoi_r = nullptr, double* slab = nullptr,
                                              const double* __restrict__ theta_s = nullptr, const int* __restrict__ kpat = nullptr,
                                              int kpart = 0, int kparts = 1, const RomGradArgs* ga = nullptr, double* mt_lds = nullptr,
                                              const double* __restrict__ ext_s = nullptr) {
  // GR with ext_s: the grouped main loop -- kpat is then RomDev::kmg and ext_s the sample's row of RomDev::ext
  // kparts > 1 (NW == 1, small batches): the sample's k-steps are split over the kparts waves of the workgroup, the partial
  // block triangles are summed through LDS (`slab`) in a fixed order and wave 0 alone runs the epilogue
  constexpr int NTL = (NB * (NB + 1) / 2 + NW - 1) / NW;
  const int q = lane >> 4, c = lane & 15;
  d4 acc[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};

  const int rp_ld = p.clock_probe == 2 ? 0 : p.rp;      // FINROM_CLOCK_PROBE=2 (timing experiment, results are garbage): every
                                                       // table fetch hits the same four rows -> no L2 traffic, same instructions
  if constexpr (NW > 1) {
    // (round 1's loop -- per-lane theta indices, two slab buffers, __syncthreads() per k-step -- reached 0.50 of the MFMA peak
    // at r = 120 where this one reaches 0.7; it is gone: its run-time indexing of the phase tables inside RomDev made hipcc keep a
    // copy of the kernel arguments in scratch memory once the fused epilogue was added)
    proj_main_uniform_mw<NB, NW, W, HF>(p, kpat, theta_s, q, c, lane, slab, acc);
  } else if ((p.clock_probe & 15) != 3) {      // (bits 4 / 5: timing experiments without table loads / without scalar loads)
    if constexpr (NW == 1 && GR) {
      if (ext_s != nullptr) proj_main_grouped<NB>(p, kpat, ext_s, q, c, acc);
      else proj_main_uniform<NB>(p, p.ext != nullptr ? p.kmeta : kpat, theta_s, q, c, acc, p.nku);      // (kpat is RomDev::kmg when the launch is a grouped one)
    } else if constexpr (NW == 1) {
      const int per = ((p.nku + kparts - 1) / kparts + 1) / 2 * 2, k0 = kpart * per;
      const int cnt = kparts == 1 ? p.nku : (k0 >= p.nku ? 0 : (p.nku - k0 < per ? p.nku - k0 : per));
      if (cnt > 0) proj_main_uniform<NB>(p, kpat + 8 * k0, theta_s, q, c, acc, cnt);      // kpat = RomDev::kmeta as a kernel parameter  // (FINROM_CLOCK_PROBE=3: the per-lane-theta loop below, for A/B timing)
    }
  } else
  // ONE copy of the MFMA group for all phases (runtime term count): several unrolled copies make hipcc
  // spill the inline-asm accumulators around every copy
  {
    double raw[ROM_MAX_NT][NB];
    int pi[ROM_MAX_NT];
    int ph = 0;
    while (ph < p.n_phases && p.phase_ks0[ph] >= p.phase_ks1[ph]) ++ph;
    if (ph < p.n_phases) load_kstep<NB>(p.tv, p.pidx, p.phase_slot0[ph], p.phase_nt[ph], rp_ld, q, c, raw, pi);
#pragma unroll 1
    for (; ph < p.n_phases; ++ph) {
      const int ks0 = p.phase_ks0[ph], ks1 = p.phase_ks1[ph], slot0 = p.phase_slot0[ph], nt = p.phase_nt[ph];
      // what follows the last k-step of this phase: the first k-step of the next non-empty phase (or padding)
      int nph = ph + 1;
      while (nph < p.n_phases && p.phase_ks0[nph] >= p.phase_ks1[nph]) ++nph;
      const int next_slot = nph < p.n_phases ? p.phase_slot0[nph] : slot0 + (ks1 - ks0) * nt;
      const int next_nt = nph < p.n_phases ? p.phase_nt[nph] : 1;
#pragma unroll 1
      for (int ks = ks0; ks < ks1; ++ks) {
        double v[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) v[b] = 0.0;
        double thp[ROM_MAX_NT];
#pragma unroll
        for (int t = 0; t < ROM_MAX_NT; ++t) thp[t] = thw[pi[t]];      // four LDS reads in flight together
#pragma unroll
        for (int t = 0; t < ROM_MAX_NT; ++t) {
          if (t < nt) {
#pragma unroll
            for (int b = 0; b < NB; ++b) v[b] = fma(thp[t], raw[t][b], v[b]);
          }
        }
        // raw is dead now: fetch the next k-step into it; the loads fly while this k-step's MFMAs issue
        // (the table is padded by one k-step of zeros, so the last prefetch stays inside it)
        const bool last = ks + 1 == ks1;
        load_kstep<NB>(p.tv, p.pidx, last ? next_slot : slot0 + (ks + 1 - ks0) * nt, last ? next_nt : nt, rp_ld, q, c, raw, pi);
        mfma_tiles<NB, NW, W>(v, acc);
      }
    }
  }

  // let the last MFMAs retire before their results are read (no hazard padding around inline asm)
  mfma_drain(acc);
  if constexpr (NW > 1 && HF) mw_half_fixup<NB, NW, W>(acc, slab, lane);      // from here on: the aligned block triangle
  if constexpr (NW == 1) {
    if (kparts > 1) {
      if (kpart > 0) {
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) slab[((kpart - 1) * NTL + t) * 256 + g * 64 + lane] = acc[t][g];
      }
      __syncthreads();
      if (kpart > 0) return;
      for (int w = 1; w < kparts; ++w)
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[t][g] += slab[((w - 1) * NTL + t) * 256 + g * 64 + lane];
    }
  }
  int bad = 0;
  // factor == 2 with nothing but qoi_r wanted (the sample-pair path): factorisation, both substitutions and the reduced QoI as
  // MFMA-form panel operations (fused_solve_sw), after B_r below
  const bool qoi_only = NW == 1 && NB <= 5 && !SK && factor == 2 && w_r == nullptr && qoi_r != nullptr && mt_lds != nullptr &&
                        p.n_obs <= 15;
  if constexpr (NW == 1 && NB <= 6) {
    if (factor && !qoi_only) {      // A_r = U^T U in registers; what is written below is then U^T (= L, packed by columns)
      bad = chol_tiles<NB>(acc, q, c, p.r);
      if (bad && info != nullptr && lane == 0) atomicOr(&info[s], 2);
    }
  }
  // factor == 2: substitutions + QoI below, nothing stored but w_r, qoi_r (r <= 80)
  // factor == 3 (multi-wave kernels): factorisation + QoI in registers, nothing stored but qoi_r (fused_solve_mw)
  // factor == 4 (split-K kernel, SK): also the adjoint solve; v_r, w_r go to RomGradArgs::vw for rom_grad_contract_small_kernel
  const bool fused_solve = (NW == 1 && (NB <= 5 || SK) && (factor == 2 || (SK && factor == 4))) || (NW > 1 && factor == 3);
  // C/D layout of v_mfma_f64_16x16x4_f64: lane holds D[row = (lane>>4) + 4*g][col = lane&15].
  // A_r is symmetric: tile (ti <= tj) element (row, col) is written as the LOWER element
  // (i = col, k = row) of the packed column-major lower triangle the solve kernel reads.
  const int R = p.rp;
  if (!fused_solve) {
  double* A = Ar + s * (int64_t)(R * (R + 1) / 2);
  int idx = 0, mine = 0;
#pragma unroll
  for (int ti = 0; ti < NB; ++ti)
#pragma unroll
    for (int tj = ti; tj < NB; ++tj) {
      if (idx % NW == W) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * ti + q + 4 * g, col = 16 * tj + c;
          if (ti != tj || col >= row) A[row * R - (row * (row - 1)) / 2 + col - row] = acc[mine][g];
        }
        ++mine;
      }
      ++idx;
    }
  }

  // B_r = psi^T F (rom :297): F is non-zero on the root nodes only; their rows of psi are rebuilt
  // here (a handful of k-steps, VALU only) and reduced over the 4 row-groups by lane shuffles.
  double bacc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) bacc[b] = 0.0;
  if (W == 0) {
#pragma unroll 1
    for (int ks = 0; ks < p.rhs_nk; ++ks) {        // (not unrolled: the accumulator tiles are still live here)
      double v[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) v[b] = 0.0;
#pragma unroll 1
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
      bacc[b] = x;                               // B_r[16 b + c], the same in all four row groups
      if (!fused_solve && q == 0) Br[s * p.rp + 16 * b + c] = x;
    }
    if constexpr (NW == 1 && NB <= 5 && !SK) {
      if (qoi_only) {
        double qv;
        bad = fused_solve_sw<NB>(p, acc, bacc, q, c, mt_lds, qv);
        if (bad && info != nullptr && lane == 0) atomicOr(&info[s], 2);
        if (lane >= 1 && lane <= p.n_obs) qoi_r[s * p.n_obs + lane - 1] = bad ? __builtin_nan("") : qv;
        return;
      }
    }
    if constexpr (NW == 1 && (NB <= 5 || SK)) {
      if (fused_solve) {
        double xr[NB];
        solve_tiles<NB>(acc, bacc, q, c, xr);
        const double nanv = __builtin_nan("");
        if constexpr (SK) {
          if (factor == 4) {
            // adjoint gradient (rom/averaged_affine_ROM.py:335-356), the solves in registers: residual of the observations,
            // v_r = A_r^-1 (B_obs Phi)^T (data - qoi_r); w_r, v_r -> LDS for the contraction with the blocks G_pi
            const int R = p.rp;
            double* rs = slab;
            const double* dat = ga->data + (ga->data_stride ? s * ga->data_stride : 0);
            double jl = 0.0;
            for (int o0 = 0; o0 < p.n_obs; o0 += 4) {
              const int o = o0 + q;
              double d = 0.0;
#pragma unroll
              for (int t = 0; t < NB; ++t)
                if (o < p.n_obs && 16 * t + c < p.r) d = fma(p.obs_phi[o * p.r + 16 * t + c], xr[t], d);
              d += __shfl_xor(d, 8); d += __shfl_xor(d, 4); d += __shfl_xor(d, 2); d += __shfl_xor(d, 1);
              if (o < p.n_obs && c == 0) {
                if (qoi_r != nullptr) qoi_r[s * p.n_obs + o] = bad ? nanv : d;
                const double res = dat[o] - d;
                rs[o] = res;
                jl = fma(res, res, jl);
              }
            }
            jl += __shfl_xor(jl, 16); jl += __shfl_xor(jl, 32);      // the four row groups' lanes c == 0
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            double b2[NB];
#pragma unroll
            for (int t = 0; t < NB; ++t) b2[t] = 0.0;
            for (int o = q; o < p.n_obs; o += 4) {             // row group q takes observations q, q + 4, ...
              const double ro = rs[o];
#pragma unroll
              for (int t = 0; t < NB; ++t)
                if (16 * t + c < p.r) b2[t] = fma(p.obs_phi[o * p.r + 16 * t + c], ro, b2[t]);
            }
#pragma unroll
            for (int t = 0; t < NB; ++t) { double x = b2[t]; x += __shfl_xor(x, 16); x += __shfl_xor(x, 32); b2[t] = x; }
            double vr[NB];
            solve_tiles<NB>(acc, b2, q, c, vr);
            if (q == 0) {                                    // v_r | w_r -> scratch for the contraction kernel (NaN marks a failed sample)
              double* dst = ga->vw + s * (int64_t)(2 * R);
#pragma unroll
              for (int t = 0; t < NB; ++t) {
                const bool in = 16 * t + c < p.r;
                dst[16 * t + c] = in ? (bad ? nanv : vr[t]) : 0.0; dst[R + 16 * t + c] = in ? (bad ? nanv : xr[t]) : 0.0;
                if (w_r != nullptr && in) w_r[s * p.r + 16 * t + c] = bad ? nanv : xr[t];
              }
            }
            if (lane == 0) ga->J[s] = bad ? nanv : 0.5 * jl;
            return;
          }
        }
        if (w_r != nullptr && q == 0) {
#pragma unroll
          for (int t = 0; t < NB; ++t)
            if (16 * t + c < p.r) w_r[s * p.r + 16 * t + c] = bad ? nanv : xr[t];
        }
        // qoi_r = (B_obs Phi) w_r (rom :323-333): row group q takes observations q, q + 4, ...
        if (qoi_r != nullptr)
          for (int o0 = 0; o0 < p.n_obs; o0 += 4) {
            const int o = o0 + q;
            double d = 0.0;
#pragma unroll
            for (int t = 0; t < NB; ++t)
              if (o < p.n_obs && 16 * t + c < p.r) d = fma(p.obs_phi[o * p.r + 16 * t + c], xr[t], d);
            d += __shfl_xor(d, 8); d += __shfl_xor(d, 4); d += __shfl_xor(d, 2); d += __shfl_xor(d, 1);
            if (o < p.n_obs && c == 0) qoi_r[s * p.n_obs + o] = bad ? nanv : d;
          }
      }
    }
  }
  if constexpr (NW > 1) {
    if (factor == 3) fused_solve_mw<NB, NW, W>(p, acc, bacc, s, lane, slab, qoi_r, info);
  }
}

// Small batches (one-sample call patterns: MAP / HMC): ONE sample per workgroup, its k-steps split over the KS waves -- a lone
// wave walking all 400 k-steps is 0.23 ms of pure MFMA latency at r = 81; four waves take a quarter each and hand their partial
// block triangles to wave 0 through LDS ((KS - 1) x NT tiles of 2 KB: dynamic shared memory)
template <int NB, int KS>
__device__ __forceinline__ void rom_proj_entry_splitk(RomDev p, const double* __restrict__ theta, int64_t S,
                                                      double* __restrict__ Ar, double* __restrict__ Br, int factor,
                                                      int* __restrict__ info, double* __restrict__ w_r, double* __restrict__ qoi_r,
                                                      const int* __restrict__ kpat, const RomGradArgs& ga) {
  extern __shared__ __attribute__((aligned(16))) double red_lds[];
  __shared__ double th[KS][32];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t s = blockIdx.x;
  if (s >= S) return;
  if (lane == 0) { th[wave][0] = 1.0; th[wave][p.P + 1] = 0.0; }
  if (lane < p.P) th[wave][lane + 1] = theta[s * p.P + lane];
  __builtin_amdgcn_wave_barrier();
  const unsigned long long ta = (unsigned long long)(theta + (int64_t)blockIdx.x * p.P);
  const double* theta_s = (const double*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ta >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ta));
  rom_proj_body<NB, 1, 0, true>(p, th[wave], s, lane, Ar, Br, factor, info, w_r, qoi_r, red_lds, theta_s, kpat, wave, KS, &ga);
}

// NW waves share one sample (each owns every NW-th tile of the upper block triangle); a
// workgroup is max(4, NW) waves = max(4, NW)/NW samples.
template <int NB, int NW, bool GR = false, bool HF = false>
__device__ __forceinline__ void rom_proj_entry(RomDev p, const double* __restrict__ theta, int64_t S,
                                                       double* __restrict__ Ar, double* __restrict__ Br, int factor,
                                                       int* __restrict__ info, double* __restrict__ w_r, double* __restrict__ qoi_r,
                                                       const int* __restrict__ kpat = nullptr) {
  constexpr int WPB = NW > 4 ? NW : 4;
  static_assert(NW == 1 || NW == 4 || NW == 8, "a workgroup is one sample when its waves share the slab (uniform early exit, barriers)");
  __shared__ double th[WPB][32];
  __shared__ double mt_sw[NW == 1 && NB <= 5 ? WPB * ROM_SW_LDS : 1];      // fused_solve_sw: B_r, per wave
  __shared__ double slab_lds[NW > 1 ? (3 * NB * 64 > fused_mw_lds_doubles<NB>() ? 3 * NB * 64 : fused_mw_lds_doubles<NB>()) : 1];
  double* slab = slab_lds;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: say so (scalar branches below)
  trace_begin(p.trace, blockIdx.x);
  const int64_t s = (int64_t)blockIdx.x * (WPB / NW) + wave / NW;
  if (s >= S) return;                       // no block-wide barrier below
  if (lane == 0) { th[wave][0] = 1.0; th[wave][p.P + 1] = 0.0; }    // theta index P + 1 = the zero of padded table slots
  if (lane < p.P) th[wave][lane + 1] = theta[s * p.P + lane];
  __builtin_amdgcn_wave_barrier();
  const double* thw = th[wave];
  // the sample index is wave-uniform: say so, so that the sample's parameters can be fetched with scalar loads
  const unsigned long long ta = (unsigned long long)(theta + ((int64_t)blockIdx.x * (WPB / NW) + wave / NW) * p.P);
  const double* theta_s = (const double*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ta >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ta));     // provably in SGPRs
  const double* theta_u = theta_s;
  if constexpr (NW == 1) {
    const double* ext_s = nullptr;
    if constexpr (GR) {
      // (the grouped form divides by the conductivities: a sample with a zero, tiny (< 1e-60), huge (> 1e60) or non-finite one
      // takes the ungrouped loop -- same sums as a handle without the grouped tables)
      const double tl = lane < p.P ? theta[s * p.P + lane] : 1.0;
      const bool plain = __ballot(!(__builtin_fabs(tl) > 1e-60 && __builtin_fabs(tl) < 1e60)) == 0;      // (ratios are squared: <= 1e240)
      if (p.ext != nullptr && plain) {
        const unsigned long long ea = (unsigned long long)(p.ext + ((int64_t)blockIdx.x * WPB + wave) * p.n_ext);
        ext_s = (const double*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ea >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ea));
      }
    }
    rom_proj_body<NB, 1, 0, false, GR>(p, thw, s, lane, Ar, Br, factor, info, w_r, qoi_r, nullptr, theta_s, kpat, 0, 1, nullptr,
                                       NB <= 5 ? mt_sw + wave * ROM_SW_LDS : nullptr, ext_s);
  } else if constexpr (NW == 4) {
    switch (wave % 4) {
      case 0: rom_proj_body<NB, 4, 0, false, false, HF>(p, thw, s, lane, Ar, Br, factor, info, nullptr, qoi_r, slab, theta_u, kpat); break;
      case 1: rom_proj_body<NB, 4, 1, false, false, HF>(p, thw, s, lane, Ar, Br, factor, info, nullptr, qoi_r, slab, theta_u, kpat); break;
      case 2: rom_proj_body<NB, 4, 2, false, false, HF>(p, thw, s, lane, Ar, Br, factor, info, nullptr, qoi_r, slab, theta_u, kpat); break;
      default: rom_proj_body<NB, 4, 3, false, false, HF>(p, thw, s, lane, Ar, Br, factor, info, nullptr, qoi_r, slab, theta_u, kpat); break;
    }
  } else {
    switch (wave % 8) {
      case 0: rom_proj_body<NB, 8, 0, false, false, HF>(p, thw, s, lane, Ar, Br, factor, info, nullptr, qoi_r, slab, theta_u, kpat); break;
      case 1: rom_proj_body<NB, 8, 1, false, false, HF>(p, thw, s, lane, Ar, Br, factor, info, nullptr, qoi_r, slab, theta_u, kpat); break;
      case 2: rom_proj_body<NB, 8, 2, false, false, HF>(p, thw, s, lane, Ar, Br, factor, info, nullptr, qoi_r, slab, theta_u, kpat); break;
      case 3: rom_proj_body<NB, 8, 3, false, false, HF>(p, thw, s, lane, Ar, Br, factor, info, nullptr, qoi_r, slab, theta_u, kpat); break;
      case 4: rom_proj_body<NB, 8, 4, false, false, HF>(p, thw, s, lane, Ar, Br, factor, info, nullptr, qoi_r, slab, theta_u, kpat); break;
      case 5: rom_proj_body<NB, 8, 5, false, false, HF>(p, thw, s, lane, Ar, Br, factor, info, nullptr, qoi_r, slab, theta_u, kpat); break;
      case 6: rom_proj_body<NB, 8, 6, false, false, HF>(p, thw, s, lane, Ar, Br, factor, info, nullptr, qoi_r, slab, theta_u, kpat); break;
      default: rom_proj_body<NB, 8, 7, false, false, HF>(p, thw, s, lane, Ar, Br, factor, info, nullptr, qoi_r, slab, theta_u, kpat); break;
    }
  }
  trace_end(p.trace, blockIdx.x);
}


template <int NB, int NW, bool HF = false>
__global__ __launch_bounds__((NW > 4 ? NW : 4) * 64, (NW == 4 && NB <= 8) ? 3 : 2) void rom_proj_kernel(RomDev p, const double* __restrict__ theta, int64_t S,
                                                       double* __restrict__ Ar, double* __restrict__ Br, int factor,
                                                       int* __restrict__ info, double* __restrict__ w_r, double* __restrict__ qoi_r,
                                                       const int* __restrict__ kpat) {
  rom_proj_entry<NB, NW, false, HF>(p, theta, S, Ar, Br, factor, info, w_r, qoi_r, kpat);
}

}  // namespace finrom
