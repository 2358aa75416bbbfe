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
// Bound: HBM/L2 bandwidth (2 loads of 8 B per multiply-add, no reuse in registers).
#include "finrom_internal.h"

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
// fused FOM kernel, one wave per block of 64 samples
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void fom_kernel(FomDev p, const double* __restrict__ xT, int64_t S,
                                                 double* __restrict__ Lw, double* __restrict__ invd,
                                                 double* __restrict__ yw, double* __restrict__ qoi,
                                                 int* __restrict__ info) {
  // LDS: the entries of the row being eliminated, [maxrow][64] (lane-private column of 8-B slots:
  // conflict-free ds_read_b64 / ds_write_b64).  The "a" operand of every multiply-add is an entry
  // of the current row, so only the "b" operand (an earlier row of L) comes from global memory.
  extern __shared__ __attribute__((aligned(16))) double rowc[];
  const int lane = threadIdx.x;
  const int64_t blk = blockIdx.x;
  const double* xb = xT + blk * (int64_t)p.xdim * 64 + lane;
  double* Lb = Lw + blk * (int64_t)p.nnzL * 64 + lane;
  double* ib = invd + blk * (int64_t)p.n * 64 + lane;
  double* yb = yw + blk * (int64_t)p.n * 64 + lane;
  double* rc = rowc + lane;
  int bad = 0;

  // Every dependent global round trip costs ~1 us here (the wave has nothing else to do), so each
  // entry issues ALL the loads it needs in one batch: up to 8 b-operands (predicated past the end of
  // the list), the 1/L_jj it will be scaled by, and the first x values of its assembly.
  // ---- numeric factorisation A = L L^T, row by row, fused with  L y = F ---------------
  if (p.debug_phases & 1)
  for (int i = 0; i < p.n; ++i) {
    const int e0 = p.row_ptr[i], e1 = p.row_ptr[i + 1];
    double inv_i = 0.0;
    for (int e = e0; e < e1; ++e) {
      int q = p.pair_ptr[e];
      const int q1 = p.pair_mid[e], q2 = p.pair_ptr[e + 1];
      const int t0 = p.asm_ptr[e], t1 = p.asm_ptr[e + 1];
      const bool diag = (e == e1 - 1);
      // one batch of loads
      double lb[8], xv[4];
#pragma unroll
      for (int u = 0; u < 8; ++u) lb[u] = Lb[(int64_t)p.pair_b[(q + u < q1) ? q + u : (q1 > q ? q1 - 1 : 0)] * 64];
#pragma unroll
      for (int u = 0; u < 4; ++u) xv[u] = xb[(int64_t)p.asm_idx[(t0 + u < t1) ? t0 + u : 0] * 64];
      const double invj = diag ? 0.0 : ib[(int64_t)p.ent_col[e] * 64];
      // affine assembly of A_e(x)            (fom :160-161 / rom :154-163)
      double acc = p.asm_c0[e];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = fma((t0 + u < t1) ? p.asm_w[t0 + u] : 0.0, xv[u], acc);
      for (int t = t0 + 4; t < t1; ++t) acc = fma(p.asm_w[t], xb[(int64_t)p.asm_idx[t] * 64], acc);
      // acc -= sum_k L_ik L_jk : L_ik from the LDS row cache, L_jk from global
      double acc2 = 0.0;
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        const double la0 = (q + u < q1) ? rc[(p.pair_a[q + u] - e0) * 64] : 0.0;
        const double la1 = (q + u + 1 < q1) ? rc[(p.pair_a[q + u + 1] - e0) * 64] : 0.0;
        acc = fma(-la0, lb[u], acc); acc2 = fma(-la1, lb[u + 1], acc2);
      }
      q += 8;
      for (; q < q1; q += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) lb[u] = Lb[(int64_t)p.pair_b[(q + u < q1) ? q + u : q1 - 1] * 64];
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          const double la0 = (q + u < q1) ? rc[(p.pair_a[q + u] - e0) * 64] : 0.0;
          const double la1 = (q + u + 1 < q1) ? rc[(p.pair_a[q + u + 1] - e0) * 64] : 0.0;
          acc = fma(-la0, lb[u], acc); acc2 = fma(-la1, lb[u + 1], acc2);
        }
      }
      for (q = q1; q < q2; ++q)      // tail of very long rows: both operands from global
        acc2 = fma(-Lb[(int64_t)p.pair_a[q] * 64], Lb[(int64_t)p.pair_b[q] * 64], acc2);
      acc += acc2;
      if (diag) {
        bad |= !(acc > 0.0);
        const double d = sqrt(acc);
        inv_i = 1.0 / d;
        Lb[(int64_t)e * 64] = d;
        ib[(int64_t)i * 64] = inv_i;
      } else {
        const double l = acc * invj;
        Lb[(int64_t)e * 64] = l;
        if (e - e0 < FOM_ROW_CACHE) rc[(e - e0) * 64] = l;
      }
    }
    // forward substitution row: y_i = (F_i - sum_k L_ik y_k) / L_ii, loads batched by 8
    double yi = p.rhs[i], yi2 = 0.0;
    for (int e = e0; e < e1 - 1; e += 8) {
      double yv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) yv[u] = yb[(int64_t)p.ent_col[(e + u < e1 - 1) ? e + u : e1 - 2] * 64];
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        const int ea = e + u, eb = e + u + 1;
        const double la = (ea < e1 - 1) ? ((ea - e0 < FOM_ROW_CACHE) ? rc[(ea - e0) * 64] : Lb[(int64_t)ea * 64]) : 0.0;
        const double lc = (eb < e1 - 1) ? ((eb - e0 < FOM_ROW_CACHE) ? rc[(eb - e0) * 64] : Lb[(int64_t)eb * 64]) : 0.0;
        yi = fma(-la, yv[u], yi); yi2 = fma(-lc, yv[u + 1], yi2);
      }
    }
    yb[(int64_t)i * 64] = (yi + yi2) * inv_i;
  }

  // ---- L^T w = y, in place (w overwrites y); loads batched by 8 -------------------------
  if (p.debug_phases & 2)
  for (int i = p.n - 1; i >= 0; --i) {
    const int c0 = p.col_ptr[i], c1 = p.col_ptr[i + 1];
    double wi = yb[(int64_t)i * 64], wi2 = 0.0;
    const double invi = ib[(int64_t)i * 64];
    for (int c = c0; c < c1; c += 8) {
      double lv[8], wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int cc = (c + u < c1) ? c + u : c1 - 1;
        lv[u] = Lb[(int64_t)p.col_ent[cc] * 64];
        wv[u] = yb[(int64_t)p.col_row[cc] * 64];
      }
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        wi = fma((c + u < c1) ? -lv[u] : 0.0, wv[u], wi);
        wi2 = fma((c + u + 1 < c1) ? -lv[u + 1] : 0.0, wv[u + 1], wi2);
      }
    }
    yb[(int64_t)i * 64] = (wi + wi2) * invi;
  }

  // ---- QoI = B_obs w  (fom :408-412) ---------------------------------------------------
  const int64_t s = blk * 64 + lane;
  const double nanv = __builtin_nan("");
  if (p.debug_phases & 4)
  for (int o = 0; o < p.n_obs; ++o) {
    double qv = 0.0;
    for (int t = p.obs_ptr[o], t1 = p.obs_ptr[o + 1]; t < t1; ++t)
      qv = fma(p.obs_w[t], yb[(int64_t)p.obs_idx[t] * 64], qv);
    if (s < S) qoi[s * p.n_obs + o] = bad ? nanv : qv;
  }
  if (bad)
    for (int i = 0; i < p.n; ++i) yb[(int64_t)i * 64] = nanv;
  if (info != nullptr && s < S && bad) atomicOr(&info[s], 1);   // the ROM half may set bit 1 concurrently
}

int launch_fom(const FomDev& p, const double* xT, int64_t nblk, int64_t S, double* Lw, double* invd,
               double* yw, double* qoi, int* info, hipStream_t st) {
  if (nblk == 0) return 0;
  ScopedKernelTimer t(K_FOM, st);
  // 36 slots x 512 B = 18 KiB per wave: 7 waves per CU (1792 resident waves >= 1563 for 100k samples)
  const size_t lds = (size_t)(p.maxrow < FOM_ROW_CACHE ? p.maxrow : FOM_ROW_CACHE) * 64 * sizeof(double);
  hipLaunchKernelGGL(fom_kernel, dim3((unsigned)nblk), dim3(64), lds, st, p, xT, S, Lw, invd, yw, qoi, info);
  FR_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// unpack: blocked, permuted w -> row-major w[S][n] in the caller's dof order
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unpack_w_kernel(const double* __restrict__ yw, const int* __restrict__ perm,
                                                       int n, int64_t S, double* __restrict__ w) {
  __shared__ double tile[64][65];
  const int64_t blk = blockIdx.x;
  const int i0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int il = ty; il < 64; il += 4) {
    int i = i0 + il;
    tile[il][tx] = (i < n) ? yw[(blk * n + i) * 64 + tx] : 0.0;
  }
  __syncthreads();
  const int i = i0 + tx;
  const int col = (i < n) ? perm[i] : 0;
  for (int sl = ty; sl < 64; sl += 4) {
    int64_t s = blk * 64 + sl;
    if (s < S && i < n) w[s * n + col] = tile[tx][sl];
  }
}

int launch_unpack_w(const FomDev& p, const double* yw, int64_t S, double* w, hipStream_t st) {
  int64_t nblk = (S + 63) / 64;
  if (nblk == 0) return 0;
  ScopedKernelTimer t(K_UNPACK_W, st);
  dim3 grid((unsigned)nblk, (unsigned)((p.n + 63) / 64));
  hipLaunchKernelGGL(unpack_w_kernel, grid, dim3(256), 0, st, yw, p.perm, p.n, S, w);
  FR_HIP(hipGetLastError());
  return 0;
}

}  // namespace finrom
