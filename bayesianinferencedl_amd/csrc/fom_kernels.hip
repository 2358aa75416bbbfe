// FOM side of the hot path: batched sparse SPD solve with a fixed symbolic factorisation.
//
// Replaces, for a whole batch, the per-sample DOLFIN calls of
//   Fin.forward            fom/forward_solve.py:270-291  (assemble A(k), sparse direct solve)
//   Fin.qoi_operator       fom/forward_solve.py:408-412  (B_obs @ w)
//   AffineROMFin.forward   rom/averaged_affine_ROM.py:237-258 (same solve, affine operator)
//
// Mapping to the hardware: lane = sample.  All samples share the elimination schedule, so
// control flow and every index are wave-uniform (SGPRs, scalar loads), and each vector
// memory access is one fully coalesced 512-byte line: element e of the wave's 64 samples
// lives at  base + e*64 + lane  ("sample-blocked" layout [S/64][elements][64]).
// The schedule is a flat op stream (bayesianinferencedl_amd/symbolic.py::build_op_streams)
// interpreted by fom_vm_kernel; the "a" operand of a multiply-add comes from an LDS cache of the
// row being eliminated, the "b" operand from global memory, fetched one 8-op chunk ahead.
// Bound: HBM/L2 bandwidth (one 8-B load per multiply-add and sample, no reuse in registers).
#include "finrom_core.h"
#include <type_traits>

namespace finrom {

// ---------------------------------------------------------------------------------------
// pack: row-major x[S][d] -> blocked xT[nblk][d][64]; tail samples replicate sample S-1
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kernel(const double* __restrict__ x, int64_t S, int d,
                                                   double* __restrict__ xT) {
  __shared__ double tile[64][65];
  const int64_t blk = blockIdx.x;
  const int d0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int sl = ty; sl < 64; sl += 4) {           // read: dims fastest (coalesced along a row of x)
    int64_t s = blk * 64 + sl;
    if (s >= S) s = S - 1;
    int j = d0 + tx;
    tile[sl][tx] = (j < d) ? x[s * d + j] : 0.0;
  }
  __syncthreads();
  for (int jl = ty; jl < 64; jl += 4) {           // write: samples fastest
    int j = d0 + jl;
    if (j < d) xT[(blk * d + j) * 64 + tx] = tile[tx][jl];
  }
}

int launch_pack(const double* x, int64_t S, int d, double* xT, hipStream_t st) {
  int64_t nblk = (S + 63) / 64;
  if (nblk == 0) return 0;
  ScopedKernelTimer t(K_PACK, st);
  dim3 grid((unsigned)nblk, (unsigned)((d + 63) / 64));
  hipLaunchKernelGGL(pack_kernel, grid, dim3(256), 0, st, x, S, d, xT);
  FR_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// assembly pre-pass: G[e] = A_e(x) = c0_e + sum_t w_t x[idx_t] for every entry of L that carries a value of A
// (fom :160-161 / rom :154-163).  lane = sample, coalesced 512-B stores.  The per-entry table is a fixed-size
// record (rec_i: entry, 4 parameter indices, range of the remaining terms; rec_d: c0, 4 weights) fetched with
// two independent scalar loads, so the records of the next entries are in flight while this one is applied.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fom_assemble_kernel(FomDev p, const int* __restrict__ rec_i, const double* __restrict__ rec_d,
                                                           const int* __restrict__ asm_idx, const double* __restrict__ asm_w,
                                                           const double* __restrict__ xT, double* __restrict__ Gw) {
  const int lane = threadIdx.x & 63;
  // the wave index is uniform, but only readfirstlane tells the compiler so (scalar record loads)
  const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + 4 * blockIdx.y, nparts = 4 * gridDim.y;
  const int64_t blk = blockIdx.x;
  const double* xb = xT + blk * (int64_t)p.xdim * 64 + lane;
  double* G = Gw + blk * (int64_t)p.gsize * 64 + lane;
  constexpr int U = 4;                                    // entries in flight per wave
  for (int t0 = part * U; t0 < p.n_alist; t0 += nparts * U) {
    int ri[U][8];
    double rd[U][5];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = (t0 + u < p.n_alist) ? t0 + u : p.n_alist - 1;     // tail: recompute the last entry
#pragma unroll
      for (int k = 0; k < 8; ++k) ri[u][k] = rec_i[t * 8 + k];
#pragma unroll
      for (int k = 0; k < 5; ++k) rd[u][k] = rec_d[t * 5 + k];
    }
    double xv[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) xv[u][k] = xb[(int64_t)ri[u][1 + k] * 64];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double acc = rd[u][0];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc = fma(rd[u][1 + k], xv[u][k], acc);
      for (int q = ri[u][5]; q < ri[u][6]; ++q) acc = fma(asm_w[q], xb[(int64_t)asm_idx[q] * 64], acc);   // entries with > 4 terms
      G[(int64_t)ri[u][0] * 64] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------
// schedule interpreter: one wave per block of 64 samples executes the two op streams.
// Global operands of chunk c+1 are in flight while chunk c executes (double-buffered in
// registers); the op descriptors themselves are sequential scalar loads.
// ---------------------------------------------------------------------------------------
enum { F_FMA = 0, F_LDX = 3, F_FMAX = 4, F_FINOFF = 5, F_FINDIAG = 6, F_YSET = 7, F_FINY = 8, F_XFMA = 9, F_CADD = 10, F_FMALL = 11 };
enum { B_NOP = 0, B_WFMA = 1, B_WSET = 3, B_WFIN = 5 };

// The op arrays are separate __restrict__ kernel parameters on purpose: only then can the compiler
// prove that the stores to G never clobber them and fetch the descriptors with SCALAR loads
// (s_load_dwordx8); through the by-value struct they become vector loads + v_readfirstlane whose
// s_waitcnt vmcnt(0) drains the operand prefetch on every op.
template <int FCH>      // FCH = ops per prefetch chunk of the forward stream
__global__ __launch_bounds__(64, 5) void fom_vm_kernel(FomDev p, const int* __restrict__ fA, const int* __restrict__ fKB,
                                                    const int* __restrict__ fD, const int* __restrict__ fM,
                                                    const double* __restrict__ rhs, const double* __restrict__ fImm,
                                                    const double* __restrict__ xT, double* __restrict__ Gw, int64_t S,
                                                    int* __restrict__ info) {
  // LDS: [cache_slots] row cache | NEG1 | ZERO | INV | XREG | BAD | x[xdim] (fused assembly only), each 64 lanes x 8 B.  The interpreter state that
  // only the rare ops touch (1/L_ii of the current row, the LDX operand, the failure flag) lives in LDS, not in
  // registers: loop-carried registers that the common multiply-add does not modify cost a register move per
  // unrolled op, and every VALU instruction of this kernel competes with the projection kernel's MFMAs.
  extern __shared__ __attribute__((aligned(16))) double rowc[];
  const unsigned lane = threadIdx.x;
  const int64_t blk = blockIdx.x;
  double* __restrict__ Gs = Gw + blk * (int64_t)p.gsize * 64;     // wave-uniform base: scalar address + lane offset
  double* G = Gs + lane;
  double* rc = rowc + lane;
  trace_begin(p.trace, blk);
  const int S_NEG1 = p.cache_slots, S_ZERO = p.cache_slots + 1, S_INV = p.cache_slots + 2, S_XREG = p.cache_slots + 3,
            S_BAD = p.cache_slots + 4;
  rc[S_NEG1 * 64] = -1.0;          // "acc = A_e" is an FMA against it
  rc[S_ZERO * 64] = 0.0;           // padding ops
  rc[S_INV * 64] = 0.0; rc[S_XREG * 64] = 0.0; rc[S_BAD * 64] = 0.0;
  const int S_X0 = p.cache_slots + 5;
  if (p.fused)
    for (int j = 0; j < p.xdim; ++j) rc[(S_X0 + j) * 64] = xT[(blk * p.xdim + j) * 64 + lane];
  double acc = 0.0;
  // operand fetches are buffer loads: descriptor + element offset in SGPRs, lane offset in one VGPR -- no
  // per-load vector address arithmetic
  const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(Gs, 0, p.gsize * 512, 0x00020000);
  const int lane8 = (int)lane * 8;
  auto ldg = [&](int e) -> double {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(gres, lane8, e * 512, 0));
  };

  auto ldgb = [&](int byte_off) -> double {      // forward stream: offsets are pre-scaled on the host
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(gres, lane8, byte_off, 0));
  };
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  auto stg = [&](double v, int e) {             // G[e] = v for this lane
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), gres, lane8, e * 512, 0);
  };
  const char* rcb = reinterpret_cast<const char*>(rowc) + lane8;
  // FINROM_FOM_PHASES bits 4 / 5 (timing experiments only, results are garbage): FINOFF does not store to global memory /
  // every op takes the multiply-add path
  const int xmask = (p.debug_phases & 32) ? 0 : -1;
  const int nogroup = (p.debug_phases & 128) ? 0xFFFF : 0;      // bit 7: no group-of-four path (A/B)
  const bool nostore = p.debug_phases & 16;

#define VM_LOAD1(buf, c)                                                      \
  _Pragma("unroll") for (int u = 0; u < FCH; ++u) buf[u] = ldgb(A[(c) * FCH + u]);

  // ---- forward: numeric factorisation A = L L^T fused with L y = F -------------------
  // Instruction count per op matters: beside the projection kernel every instruction of this wave waits for a
  // gap in the MFMA stream.  Plain multiply-add: bit test + branch, LDS address, ds_read, fma.
  if (p.debug_phases & 1) {
    const int* __restrict__ A = fA; const int* __restrict__ KB = fKB; const int* __restrict__ D = fD;
    double bufA[FCH], bufB[FCH];
#define VM_STEP(buf, nxt, c)   /* execute chunk c from buf while the operands of chunk c+1 fly into nxt */ \
  /* ONE batch of scalar loads per step (descriptors of chunk c, load offsets of chunk c+1): a single scalar-memory */ \
  /* round trip instead of two -- the only latency a lone wave (one-sample calls) cannot hide */ \
  int kbv[FCH], dv[FCH], an[FCH];                                             \
  _Pragma("unroll") for (int u = 0; u < FCH; ++u) {                           \
    kbv[u] = KB[(c) * FCH + u]; dv[u] = D[(c) * FCH + u]; an[u] = A[((c) + 1) * FCH + u]; \
  }                                                                           \
  const int msk = fM[c];                                                      \
  _Pragma("unroll") for (int u = 0; u < FCH; ++u) nxt[u] = ldgb(an[u]);       \
  _Pragma("unroll") for (int h4 = 0; h4 < FCH; h4 += 4)                       \
  if ((((msk & xmask) | nogroup) & (0xF << h4)) == 0) {                       \
    /* four multiply-adds in a row (42 % of the groups): one test, the LDS operands requested together; when none of them */ \
    /* has its second operand in LDS (bits 16.. of the mask word; 32 % of the groups) that half of the work is skipped */ \
    if (((msk >> 16) & (0xF << h4)) == 0) {                                   \
      double l1[4];                                                           \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) l1[u] = *reinterpret_cast<const double*>(rcb + kbv[h4 + u]); \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) acc = fma(-l1[u], buf[h4 + u], acc); \
    } else {                                                                  \
    double l1[4], l2[4];                                                      \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                           \
      l1[u] = *reinterpret_cast<const double*>(rcb + kbv[h4 + u]);            \
      l2[u] = *reinterpret_cast<const double*>(rcb + dv[h4 + u]);             \
    }                                                                         \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) acc = fma(-l1[u], buf[h4 + u] + l2[u], acc); \
    }                                                                         \
  } else                                                                      \
  _Pragma("unroll") for (int u = h4; u < h4 + 4; ++u) {                       \
    const double ld = buf[u];                                                 \
    if (__builtin_expect(!(msk & xmask & (1 << u)), 1)) {                     \
      /* acc -= rc[b] * (G[a] + rc[d]): one of the two is exactly zero (FMA: d = the ZERO slot; FMALL: a out of range); */ \
      /* also "acc = A_e" (b = NEG1) and padding (b = ZERO) */                 \
      acc = fma(-*reinterpret_cast<const double*>(rcb + kbv[u]), ld + *reinterpret_cast<const double*>(rcb + dv[u]), acc); \
    } else {                                                                  \
      const int kb = kbv[u];                                                  \
      const int kind = kb & 255, b = (kb >> 8) - 1;                           \
      const int d = dv[u];                                                    \
      /* tested in order of frequency; stores are buffer stores (descriptor + SGPR element offset, like the loads) */ \
      if (kind == F_FINOFF) {                                                 \
        const double l = acc * ld;                                            \
        if (!nostore) stg(l, d);                                              \
        if (b >= 0) rc[b * 64] = l;                                           \
        acc = 0.0;                                                            \
      } else if (kind == F_XFMA) {                                            \
        acc = fma(fImm[d], rc[(S_X0 + b) * 64], acc);     /* fused assembly: A_e += w * x_b */ \
      } else if (kind == F_FMAX) {                                            \
        acc = fma(-rc[S_XREG * 64], ld, acc);                                 \
      } else if (kind == F_LDX) {                                             \
        rc[S_XREG * 64] = ld;                              /* row entry beyond the LDS cache */ \
      } else if (kind == F_CADD) {                                            \
        acc += fImm[d];                                                       \
      } else if (kind == F_FINDIAG) {                                         \
        if (!(acc > 0.0)) rc[S_BAD * 64] = 1.0;                               \
        /* 1/sqrt by the hardware estimate + two Newton steps (as in the ROM's in-register Cholesky): a tenth of the */ \
        /* code of sqrt() and 1.0 / t, and this block exists once per op slot of the unrolled loop */ \
        double inv = __builtin_amdgcn_rsq(acc);                               \
        inv = inv * fma(-0.5 * acc * inv, inv, 1.5);                          \
        inv = inv * fma(-0.5 * acc * inv, inv, 1.5);                          \
        const double t = acc * inv;                                           \
        stg(t, d);                                                            \
        stg(inv, p.nnzL + b);                                                 \
        rc[S_INV * 64] = inv;                                                 \
        acc = 0.0;                                                            \
      } else if (kind == F_FINY) {                                            \
        stg(acc * rc[S_INV * 64], d); acc = 0.0;                              \
      } else if (kind == F_YSET) {                                            \
        acc = rhs[d];                                                         \
      }                                                                       \
    }                                                                         \
  }
    VM_LOAD1(bufA, 0)
    for (int c = 0; c < p.nchunks_fwd; c += 2) {
      { VM_STEP(bufA, bufB, c) }
      { VM_STEP(bufB, bufA, c + 1) }
    }
#undef VM_STEP
  }
  // A sample whose matrix turned out not to be positive definite: NaN right-hand side, so that the substitution kernel
  // produces NaN w and NaN QoI for it (rows outside the failed pivot's subtree would otherwise look fine), and the flag.
  const int bad = rc[S_BAD * 64] != 0.0;
  const int64_t s = blk * 64 + lane;
  if (bad) {
    const double nanv = __builtin_nan("");
    for (int i = 0; i < p.n; ++i) G[(int64_t)(p.nnzL + p.n + i) * 64] = nanv;
  }
  if (info != nullptr && s < S && bad) atomicOr(&info[s], 1);   // the ROM half may set bit 1 concurrently
  trace_end(p.trace, blk);
}

// ---- backward: L^T w = y (w overwrites y), then QoI = B_obs w (fom :408-412) -------------------------------------------
// Its own kernel: no LDS, hence not tied to the forward interpreter's LDS-limited residency or its register budget, and the
// operands of chunk c+1 are in flight while chunk c executes (a value stored in chunk c is not fetched before chunk c+2,
// as in the forward stream).  Ops: WSET acc = G[a] | WFMA acc -= G[a] * G[b] | WFIN G[d] = acc * G[a].
__global__ __launch_bounds__(64) void fom_bwd_kernel(FomDev p, const int* __restrict__ A, const int* __restrict__ B,
                                                     const int* __restrict__ KD, const int* __restrict__ obs_ptr,
                                                     const int* __restrict__ obs_idx, const double* __restrict__ obs_w,
                                                     double* __restrict__ Gw, int64_t S, double* __restrict__ qoi) {
  const unsigned lane = threadIdx.x;
  const int64_t blk = blockIdx.x;
  double* __restrict__ Gs = Gw + blk * (int64_t)p.gsize * 64;
  double* G = Gs + lane;
  const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(Gs, 0, p.gsize * 512, 0x00020000);
  const int lane8 = (int)lane * 8;
  auto ldgb = [&](int byte_off) -> double {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(gres, lane8, byte_off, 0));
  };
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  auto stg = [&](double v, int e) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), gres, lane8, e * 512, 0);
  };
  double acc = 0.0;
  if (p.debug_phases & 2) {
    // The operands of chunk c+1 are in flight while chunk c executes; the descriptors of a step -- kinds of chunk c, operand
    // offsets of chunk c+1 -- are requested a whole step before they are used.  (Two chunks ahead measured slower, 3.29 vs
    // 3.04 ms per 100k samples: this kernel moves 196 KB per sample at 6.4 TB/s, it is bandwidth-bound.)
    double a0[VM_CHUNK], b0[VM_CHUNK], a1[VM_CHUNK], b1[VM_CHUNK];
#define BW_DESC(kd, an, bn, c)                                                \
    _Pragma("unroll") for (int u = 0; u < VM_CHUNK; ++u) {                    \
      kd[u] = KD[(c) * VM_CHUNK + u]; an[u] = A[((c) + 1) * VM_CHUNK + u]; bn[u] = B[((c) + 1) * VM_CHUNK + u]; \
    }
#define BW_STEP(ca, cb, na, nb, kd, an, bn, kdn, ann, bnn, c)                  \
  {                                                                           \
    _Pragma("unroll") for (int u = 0; u < VM_CHUNK; ++u) { na[u] = ldgb(an[u]); nb[u] = ldgb(bn[u]); } \
    BW_DESC(kdn, ann, bnn, (c) + 1)                                           \
    _Pragma("unroll") for (int u = 0; u < VM_CHUNK; ++u) {                    \
      const int kind = kd[u] & 255;                                           \
      if (kind == B_WFMA) acc = fma(-ca[u], cb[u], acc);                      \
      else if (kind == B_WSET) acc = ca[u];                                   \
      else if (kind == B_WFIN) stg(acc * ca[u], kd[u] >> 8);                  \
    }                                                                         \
  }
#pragma unroll
    for (int u = 0; u < VM_CHUNK; ++u) { a0[u] = ldgb(A[u]); b0[u] = ldgb(B[u]); }
    int kdA[VM_CHUNK], anA[VM_CHUNK], bnA[VM_CHUNK], kdB[VM_CHUNK], anB[VM_CHUNK], bnB[VM_CHUNK];
    BW_DESC(kdA, anA, bnA, 0)
    for (int c = 0; c < p.nchunks_bwd; c += 2) {     // (two spare chunks of padding cover the look-ahead and an odd count)
      BW_STEP(a0, b0, a1, b1, kdA, anA, bnA, kdB, anB, bnB, c)
      BW_STEP(a1, b1, a0, b0, kdB, anB, bnB, kdA, anA, bnA, c + 1)
    }
#undef BW_DESC
#undef BW_STEP
  }
  const int64_t s = blk * 64 + lane;
  const double* wv = G + (int64_t)(p.nnzL + p.n) * 64;
  if (p.debug_phases & 4)
  for (int o = 0; o < p.n_obs; ++o) {      // loads batched by 8
    double q0 = 0.0, q1 = 0.0;
    const int t0 = obs_ptr[o], t1 = obs_ptr[o + 1];
    for (int t = t0; t < t1; t += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = wv[(int64_t)obs_idx[(t + u < t1) ? t + u : t1 - 1] * 64];
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        q0 = fma((t + u < t1) ? obs_w[t + u] : 0.0, v[u], q0);
        q1 = fma((t + u + 1 < t1) ? obs_w[t + u + 1] : 0.0, v[u + 1], q1);
      }
    }
    if (s < S) qoi[s * p.n_obs + o] = q0 + q1;       // NaN for a flagged sample: its w is NaN
  }
}

int launch_fom_assemble(const FomDev& p, const double* xT, int64_t nblk, double* Gw, hipStream_t st) {
  if (nblk == 0) return 0;
  ScopedKernelTimer t(K_FOM_ASM, st);
  hipLaunchKernelGGL(fom_assemble_kernel, dim3((unsigned)nblk, 4), dim3(256), 0, st, p, p.asm_rec_i, p.asm_rec_d, p.asm_idx,
                     p.asm_w, xT, Gw);
  FR_HIP(hipGetLastError());
  return 0;
}

int launch_fom(const FomDev& p, const double* xT, int64_t nblk, int64_t S, double* Gw, double* qoi, int* info, hipStream_t st) {
  if (nblk == 0) return 0;
  ScopedKernelTimer t(K_FOM_PATH_INTERP, st);
  const size_t lds = (size_t)(p.cache_slots + 5 + (p.fused ? p.xdim : 0)) * 64 * sizeof(double);
  if (p.fwd_chunk == 16)
    hipLaunchKernelGGL(fom_vm_kernel<16>, dim3((unsigned)nblk), dim3(64), lds, st, p, p.f_a, p.f_kb, p.f_d, p.f_mask, p.rhs, p.f_imm, xT,
                       Gw, S, info);
  else
    hipLaunchKernelGGL(fom_vm_kernel<8>, dim3((unsigned)nblk), dim3(64), lds, st, p, p.f_a, p.f_kb, p.f_d, p.f_mask, p.rhs, p.f_imm, xT,
                       Gw, S, info);
  FR_HIP(hipGetLastError());
  hipLaunchKernelGGL(fom_bwd_kernel, dim3((unsigned)nblk), dim3(64), 0, st, p, p.b_a, p.b_b, p.b_kd, p.obs_ptr, p.obs_idx, p.obs_w, Gw, S, qoi);
  FR_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// adjoint gradient (Fin.gradient, fom/forward_solve.py:293-322), after fom_vm_kernel left L, 1/L_ii and w in G:
//   r = B_obs w - data,  J = |r|^2 / 2,  b = -B_obs^T r,  v = A^{-1} b (stored factor, "solve again" stream),
//   grad_j = sum_(a,b) dA_ab/dx_j * v_a * w_b.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void fom_adjoint_kernel(FomDev p, const int* __restrict__ rA, const int* __restrict__ rKB,
                                                         const int* __restrict__ rD, const int* __restrict__ bt_ptr,
                                                         const int* __restrict__ bt_obs, const double* __restrict__ bt_w,
                                                         const int* __restrict__ g_ptr, const int* __restrict__ g_a,
                                                         const int* __restrict__ g_b, const double* __restrict__ g_w,
                                                         double* __restrict__ Gw, int64_t S,
                                                         const double* __restrict__ qoi, const double* __restrict__ data,
                                                         int64_t data_stride, double* __restrict__ gradT,
                                                         double* __restrict__ Jout) {
  extern __shared__ __attribute__((aligned(16))) double rsd[];    // [n_obs][64] residuals
  const int lane = threadIdx.x;
  const int64_t blk = blockIdx.x;
  double* G = Gw + blk * (int64_t)p.gsize * 64 + lane;
  const int ZV = p.nnzL + p.n, VV = p.nnzL + 2 * p.n;
  int64_t s = blk * 64 + lane;
  const bool live = s < S;
  if (!live) s = S - 1;                                  // tail lanes replicate the last sample (as pack_kernel does)
  double* rs = rsd + lane;
  double jl = 0.0;
  for (int o = 0; o < p.n_obs; ++o) {
    const double r = qoi[s * p.n_obs + o] - data[(data_stride ? s * data_stride : 0) + o];
    rs[o * 64] = r;
    jl = fma(r, r, jl);
  }
  for (int i = 0; i < p.n; ++i) {                        // b = -B_obs^T r
    double acc = 0.0;
    for (int t = bt_ptr[i], t1 = bt_ptr[i + 1]; t < t1; ++t) acc = fma(-bt_w[t], rs[bt_obs[t] * 64], acc);
    G[(int64_t)(VV + i) * 64] = acc;
  }
  // v = (L L^T)^{-1} b : same interpreter as the backward stream (both operands from global, chunk by chunk)
  {
    double acc = 0.0, a1[VM_CHUNK], b1[VM_CHUNK];
    for (int c = 0; c < p.nchunks_res; ++c) {
      int kbv[VM_CHUNK], dv[VM_CHUNK];
#pragma unroll
      for (int u = 0; u < VM_CHUNK; ++u) { kbv[u] = rKB[c * VM_CHUNK + u]; dv[u] = rD[c * VM_CHUNK + u]; }
#pragma unroll
      for (int u = 0; u < VM_CHUNK; ++u) {
        const int a_ = rA[c * VM_CHUNK + u];
        const int b_ = (kbv[u] >> 8) - 1;
        a1[u] = G[(int64_t)(a_ < 0 ? 0 : a_) * 64];
        b1[u] = G[(int64_t)(b_ < 0 ? 0 : b_) * 64];
      }
#pragma unroll
      for (int u = 0; u < VM_CHUNK; ++u) {
        const int kind = kbv[u] & 255;
        if (kind == B_WFMA) acc = fma(-a1[u], b1[u], acc);
        else if (kind == B_WSET) acc = a1[u];
        else if (kind == B_WFIN) G[(int64_t)dv[u] * 64] = acc * a1[u];
      }
    }
  }
  // grad_j = sum dA_ab/dx_j v_a w_b, loads batched by 4 pairs
  for (int j = 0; j < p.xdim; ++j) {
    double g0 = 0.0, g1 = 0.0;
    const int t0 = g_ptr[j], t1 = g_ptr[j + 1];
    for (int t = t0; t < t1; t += 4) {
      double va[4], wb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int tt = (t + u < t1) ? t + u : t1 - 1;
        va[u] = G[(int64_t)(VV + g_a[tt]) * 64];
        wb[u] = G[(int64_t)(ZV + g_b[tt]) * 64];
      }
#pragma unroll
      for (int u = 0; u < 4; u += 2) {
        g0 = fma((t + u < t1) ? g_w[t + u] : 0.0, va[u] * wb[u], g0);
        g1 = fma((t + u + 1 < t1) ? g_w[t + u + 1] : 0.0, va[u + 1] * wb[u + 1], g1);
      }
    }
    gradT[(blk * p.xdim + j) * 64 + lane] = g0 + g1;      // NaN propagates from a failed factorisation
  }
  if (live) Jout[s] = 0.5 * jl;
}

int launch_fom_adjoint(const FomDev& p, int64_t nblk, int64_t S, double* Gw, const double* qoi, const double* data,
                       int64_t data_stride, double* gradT, double* J, hipStream_t st) {
  if (nblk == 0) return 0;
  ScopedKernelTimer t(K_FOM_PATH_INTERP, st);
  const size_t lds = (size_t)(p.n_obs > 0 ? p.n_obs : 1) * 64 * sizeof(double);
  hipLaunchKernelGGL(fom_adjoint_kernel, dim3((unsigned)nblk), dim3(64), lds, st, p, p.r_a, p.r_kb, p.r_d, p.bt_ptr, p.bt_obs,
                     p.bt_w, p.g_ptr, p.g_a, p.g_b, p.g_w, Gw, S, qoi, data, data_stride, gradT, J);
  FR_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// Small batches (the scalar call surface): one workgroup per sample, lane = ROW of L.  Rows are grouped into
// dependency levels (row i needs the rows of its structure); within a level every thread runs the plain up-looking
// row recurrence for its row, all operands in the sample's value vector V = [L | 1/L_ii | y -> w], which lives in LDS
// when it fits.  One barrier per level.  ~0.05 of the latency of a lone interpreter wave.
// ---------------------------------------------------------------------------------------
template <bool IN_LDS>
__global__ __launch_bounds__(256) void fom_small_kernel(FomDev p, FomSmallDev q, const double* __restrict__ x, int64_t S,
                                                        double* __restrict__ Gscratch, double* __restrict__ qoi,
                                                        double* __restrict__ w, int* __restrict__ info, FomSmallGrad ga) {
  extern __shared__ __attribute__((aligned(16))) double vlds[];
  __shared__ int bad_flag;
  __shared__ double red[4];
  __shared__ double resid[64];
  const int64_t s = blockIdx.x;
  const int tid = threadIdx.x;
  double* V = IN_LDS ? vlds : Gscratch + s * (int64_t)p.gsize;
  // the parameter vector sits in LDS behind the value vector (a per-entry global round trip otherwise)
  double* xs = vlds + (IN_LDS ? p.gsize : 0);
  for (int j = tid; j < p.xdim; j += 256) xs[j] = x[s * (int64_t)p.xdim + j];
  const int IV = p.nnzL, YV = p.nnzL + p.n;
  if (tid == 0) bad_flag = 0;
  __syncthreads();
  // ---- factorisation fused with L y = F, level by level.  Latency matters here, not bandwidth: levels are narrow
  // (median 3 rows at the fin), so SIXTEEN lanes share a row -- the index pairs of an entry are spread over them, the
  // partial sums meet in an xor butterfly inside the 16-lane group -- and the record + first index pairs of the next
  // entry are fetched while the current entry is being reduced.  The lanes of a group sit in one wave: the LDS write of
  // an entry is ordered before the reads of the following entries without a barrier.
  const int sg = tid >> 4, sl = tid & 15;
  // all-reduce over the 16 lanes of a DPP row by rotations (v_mov_b32 row_ror: register-to-register, no LDS crossbar)
  auto ror = [](double v, auto n) {
    constexpr int ctrl = 0x120 + decltype(n)::value;
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, ctrl, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), ctrl, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  };
  auto group_sum = [&](double v) {
    v += ror(v, std::integral_constant<int, 8>{}); v += ror(v, std::integral_constant<int, 4>{});
    v += ror(v, std::integral_constant<int, 2>{}); v += ror(v, std::integral_constant<int, 1>{});
    return v;
  };
  for (int l = 0; l < q.nlev_f; ++l) {
    for (int t = q.lev_ptr_f[l] + sg; t < q.lev_ptr_f[l + 1]; t += 16) {
      const int i = q.lev_rows_f[t];
      const int r0 = q.rec_ptr[i], r1 = q.rec_ptr[i + 1];
      double yacc = p.rhs[i], acc = 0.0;
      // the records of the next three steps are always in flight (the stream is padded by 4 records)
      struct Hd { int e, npair, col, flags; double c0; int4 ai0, ai1; double aw[8]; int2 pr; };
      auto fetch = [&](int r) {
        const FomSmallRec* R = q.rec + r;
        const int4* ai = reinterpret_cast<const int4*>(R->aidx);
        Hd h{R->e, R->npair, R->col, R->flags, R->c0, ai[0], ai[1], {}, R->first[sl]};
#pragma unroll
        for (int u = 0; u < 8; ++u) h.aw[u] = R->aw[u];
        return h;
      };
      auto process = [&](const Hd& h) {
        const double part = (sl < h.npair) ? -V[h.pr.x] * V[h.pr.y] : 0.0;
        if (h.flags & 4) {                               // A_e = c0 + sum_t w_t x[idx_t]; unused terms carry a zero weight
          double a0 = fma(h.aw[0], xs[h.ai0.x], h.c0), a1 = h.aw[1] * xs[h.ai0.y];
          a0 = fma(h.aw[2], xs[h.ai0.z], a0); a1 = fma(h.aw[3], xs[h.ai0.w], a1);
          a0 = fma(h.aw[4], xs[h.ai1.x], a0); a1 = fma(h.aw[5], xs[h.ai1.y], a1);
          a0 = fma(h.aw[6], xs[h.ai1.z], a0); a1 = fma(h.aw[7], xs[h.ai1.w], a1);
          acc += a0 + a1;
        }
        acc += group_sum(part);                          // the same value in all 16 lanes
        if (h.flags & 1) {                               // last record of the entry: finish it
          if (h.flags & 2) {
            if (!(acc > 0.0)) bad_flag = 1;
            const double d = sqrt(acc), inv = 1.0 / d;
            if (sl == 0) { V[h.e] = d; V[IV + i] = inv; V[YV + i] = yacc * inv; }
          } else {
            const double lij = acc * V[IV + h.col];
            if (sl == 0) V[h.e] = lij;
            yacc = fma(-lij, V[YV + h.col], yacc);
          }
          acc = 0.0;
        }
      };
      // three record buffers with fixed roles (rotating them through register copies would make every step wait for the
      // youngest load): consume one, refill it with the record three steps ahead
      Hd ha = fetch(r0), hb = fetch(r0 + 1), hc = fetch(r0 + 2);
      for (int r = r0; r < r1; r += 3) {
        { const Hd h = ha; ha = fetch(r + 3); process(h); }
        if (r + 1 < r1) { const Hd h = hb; hb = fetch(r + 4); process(h); }
        if (r + 2 < r1) { const Hd h = hc; hc = fetch(r + 5); process(h); }
      }
    }
    __syncthreads();
  }
  // ---- L^T w = y, w overwrites y
  for (int l = 0; l < q.nlev_b; ++l) {
    for (int t = q.lev_ptr_b[l] + sg; t < q.lev_ptr_b[l + 1]; t += 16) {
      const int i = q.lev_rows_b[t];
      const int c0 = q.col_ptr[i], c1 = q.col_ptr[i + 1];
      double part = 0.0;
      for (int c = c0 + sl; c < c1; c += 16) {
        const int2 it = q.colv[c];
        part = fma(-V[it.x], V[YV + it.y], part);
      }
      const double wi = (V[YV + i] + group_sum(part)) * V[IV + i];
      if (sl == 0) V[YV + i] = wi;
    }
    __syncthreads();
  }
  const bool bad = bad_flag != 0;
  const double nanv = __builtin_nan("");
  // ---- QoI = B_obs w, one block reduction per observation
  for (int o = 0; o < p.n_obs; ++o) {
    double part = 0.0;
    for (int t = p.obs_ptr[o] + tid; t < p.obs_ptr[o + 1]; t += 256) part = fma(p.obs_w[t], V[YV + p.obs_idx[t]], part);
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
      const double qv = red[0] + red[1] + red[2] + red[3];
      qoi[s * p.n_obs + o] = bad ? nanv : qv;
      if (ga.grad != nullptr) resid[o] = qv - ga.data[(ga.data_stride ? s * ga.data_stride : 0) + o];
    }
    __syncthreads();
  }
  // ---- adjoint gradient (Fin.gradient, fom :293-322) with the factor still in place:  b = -B_obs^T r,  L L^T v = b,
  //      grad_j = sum dA_ab/dx_j v_a w_b,  J = |r|^2 / 2
  if (ga.grad != nullptr) {
    const int VV = p.nnzL + 2 * p.n;
    for (int i = tid; i < p.n; i += 256) {
      double acc = 0.0;
      for (int t = p.bt_ptr[i]; t < p.bt_ptr[i + 1]; ++t) acc = fma(-p.bt_w[t], resid[p.bt_obs[t]], acc);
      V[VV + i] = acc;
    }
    __syncthreads();
    for (int l = 0; l < q.nlev_f; ++l) {                 // L z = b, rows level by level, 16 lanes per row
      for (int t = q.lev_ptr_f[l] + sg; t < q.lev_ptr_f[l + 1]; t += 16) {
        const int i = q.lev_rows_f[t];
        const int e0 = q.row_ptr[i], e1 = q.row_ptr[i + 1] - 1;
        double part = 0.0;
        for (int e = e0 + sl; e < e1; e += 16) part = fma(-V[e], V[VV + q.ent_col[e]], part);
        const double zi = (V[VV + i] + group_sum(part)) * V[IV + i];
        if (sl == 0) V[VV + i] = zi;
      }
      __syncthreads();
    }
    for (int l = 0; l < q.nlev_b; ++l) {                 // L^T v = z
      for (int t = q.lev_ptr_b[l] + sg; t < q.lev_ptr_b[l + 1]; t += 16) {
        const int i = q.lev_rows_b[t];
        const int c0 = q.col_ptr[i], c1 = q.col_ptr[i + 1];
        double part = 0.0;
        for (int c = c0 + sl; c < c1; c += 16) {
          const int2 it = q.colv[c];
          part = fma(-V[it.x], V[VV + it.y], part);
        }
        const double vi = (V[VV + i] + group_sum(part)) * V[IV + i];
        if (sl == 0) V[VV + i] = vi;
      }
      __syncthreads();
    }
    for (int j0 = 0; j0 < p.xdim; j0 += 16) {            // one 16-lane group per parameter
      const int j = j0 + sg;
      double part = 0.0;
      if (j < p.xdim)
        for (int t = p.g_ptr[j] + sl; t < p.g_ptr[j + 1]; t += 16) part = fma(p.g_w[t], V[VV + p.g_a[t]] * V[YV + p.g_b[t]], part);
      part = group_sum(part);
      if (j < p.xdim && sl == 0) ga.grad[s * (int64_t)p.xdim + j] = bad ? nanv : part;
    }
    if (tid == 0) {
      double jl = 0.0;
      for (int o = 0; o < p.n_obs; ++o) jl = fma(resid[o], resid[o], jl);
      ga.J[s] = bad ? nanv : 0.5 * jl;
    }
  }
  if (w != nullptr)
    for (int i = tid; i < p.n; i += 256) w[s * (int64_t)p.n + p.perm[i]] = bad ? nanv : V[YV + i];
  if (info != nullptr && tid == 0 && bad) atomicOr(&info[s], 1);
}

int launch_fom_small(const FomDev& p, const FomSmallDev& q, const double* x, int64_t S, double* Gscratch, double* qoi, double* w,
                     int* info, hipStream_t st, const FomSmallGrad& g) {
  if (S == 0) return 0;
  ScopedKernelTimer t(K_FOM_PATH_SMALL, st);
  if (q.in_lds) {
    const size_t lds = (size_t)(p.gsize + p.xdim) * sizeof(double);
    FR_HIP(hipFuncSetAttribute((const void*)fom_small_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fom_small_kernel<true>, dim3((unsigned)S), dim3(256), lds, st, p, q, x, S, Gscratch, qoi, w, info, g);
  } else {
    hipLaunchKernelGGL(fom_small_kernel<false>, dim3((unsigned)S), dim3(256), (size_t)p.xdim * sizeof(double), st, p, q, x, S, Gscratch, qoi, w, info, g);
  }
  FR_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// unpack: blocked (optionally permuted) values -> row-major [S][d] in the caller's order
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unpack_w_kernel(const double* __restrict__ Gw, const int* __restrict__ perm,
                                                       int n, int64_t gsize, int woff, int64_t S, double* __restrict__ w) {
  __shared__ double tile[64][65];
  const int64_t blk = blockIdx.x;
  const int i0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int il = ty; il < 64; il += 4) {
    int i = i0 + il;
    tile[il][tx] = (i < n) ? Gw[(blk * gsize + woff + i) * 64 + tx] : 0.0;
  }
  __syncthreads();
  const int i = i0 + tx;
  const int col = (i < n) ? (perm != nullptr ? perm[i] : i) : 0;
  for (int sl = ty; sl < 64; sl += 4) {
    int64_t s = blk * 64 + sl;
    if (s < S && i < n) w[s * n + col] = tile[tx][sl];
  }
}

int launch_unpack(const double* srcT, int64_t S, int d, int64_t blk_stride, int off, const int* perm, double* dst, hipStream_t st) {
  int64_t nblk = (S + 63) / 64;
  if (nblk == 0) return 0;
  ScopedKernelTimer t(K_UNPACK_W, st);
  dim3 grid((unsigned)nblk, (unsigned)((d + 63) / 64));
  hipLaunchKernelGGL(unpack_w_kernel, grid, dim3(256), 0, st, srcT, perm, d, blk_stride, off, S, dst);
  FR_HIP(hipGetLastError());
  return 0;
}

int launch_unpack_w(const FomDev& p, const double* Gw, int64_t S, double* w, hipStream_t st) {
  return launch_unpack(Gw, S, p.n, p.gsize, p.nnzL + p.n, p.perm, w, st);
}

}  // namespace finrom
