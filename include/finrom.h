/*
 * finrom.h -- C ABI of libfinrom_hip.so: the MI355X (gfx950) implementation of the
 * thermal-fin FOM + ROM forward-solve hot path of sheroze1123/BayesianInferenceDL.
 *
 * The reference has no FFI of its own (pure Python); its boundary is the
 * Python class surface.  Each entry point below names the reference call it replaces
 * (paths relative to the reference repository).  The Python mirror of those classes
 * (bayesianinferencedl_amd/fom/forward_solve.py, rom/averaged_affine_ROM.py, ...) binds
 * these symbols through ctypes (bayesianinferencedl_amd/_ffi.py); see INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes, no C++ / torch types; all arithmetic is IEEE fp64;
 *  - every function returns 0 on success or a negative finrom_status; it never throws
 *    or aborts; finrom_last_error() gives the thread-local message of the last failure;
 *  - descriptor arrays are HOST pointers, borrowed for the duration of the create call
 *    and copied to the device; the handle owns all device memory it allocates;
 *  - batch arrays (x, theta, qoi, w, ...) are DEVICE pointers, row-major [S x dim],
 *    caller-allocated (finrom_malloc or any other HIP allocator, e.g. a torch tensor);
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are
 *    asynchronous on that stream unless stated; a handle is used by one host thread at
 *    a time (the reference classes are not re-entrant either).
 *  - info[s] != 0 marks a failed sample (bit 0: FOM pivot <= 0 or NaN, bit 1: ROM pivot);
 *    its outputs are NaN.  info must be zero-initialised by the caller (bits are OR-ed in).
 */
#ifndef FINROM_H
#define FINROM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FINROM_ABI_VERSION 12

typedef enum {
  FINROM_OK = 0,
  FINROM_ERR_ARG = -1,      /* bad argument / inconsistent descriptor */
  FINROM_ERR_HIP = -2,      /* a HIP runtime call failed (message has the hipError name) */
  FINROM_ERR_NOMEM = -3,    /* device allocation failed */
  FINROM_ERR_UNSUPPORTED = -4
} finrom_status;

typedef struct finrom_fom_s* finrom_fom_t;
typedef struct finrom_rom_s* finrom_rom_t;
typedef struct finrom_sampler_s* finrom_sampler_t;
typedef struct finrom_mlp_s* finrom_mlp_t;

int finrom_version(void);
const char* finrom_last_error(void);

/* ---- device plumbing (so that callers need no other HIP binding) -------------------- */
int finrom_device_count(int* count);
int finrom_set_device(int ordinal);
int finrom_malloc(void** dptr, size_t bytes);
int finrom_free(void* dptr);
/* Stream capture (HIP graphs).  Library calls may be captured (hmc.run_chains_device captures a whole HMC proposal around
 * finrom_romml_grad), with two rules the library enforces itself: (1) while a capture is open on any stream the library knows
 * of, it issues no hipFree / hipMalloc / synchronous call -- finrom_free and the *_destroy functions QUEUE their work (the
 * handle is dead for the caller at once) and the next library call that finds no capture open runs the queue; finrom_malloc,
 * the *_create functions, the synchronous copies and any call that would have to grow a handle's workspace fail with
 * FINROM_ERR_UNSUPPORTED instead (run the call once with the same batch size before capturing); (2) a workspace that a captured
 * call used is kept alive until its handle is destroyed, even if a later, larger call replaces it -- a replayed graph never
 * points at freed memory (destroying the handle while its graph is still replayed remains the caller's error).
 * The library learns of a capture from the streams it is handed: every entry point queries its `stream` argument, and
 * finrom_note_stream(stream) tells it about a stream without doing anything else (returns 1 if that stream is capturing, else 0;
 * the Python layer calls it with torch's current stream before it frees or destroys anything from a finaliser, which may run
 * inside somebody's capture).  finrom_free_async(ptr, stream): as finrom_free for a buffer whose last user was a launch on
 * `stream` -- the buffer is parked with an event recorded there and its next owner waits for that event (finrom_free assumes the
 * last user has finished or ran on the default stream).  finrom_deferred_count: queued frees / destructions (tests);
 * finrom_flush_deferred runs the queue if no capture is open and returns what is left. */
int finrom_free_async(void* dptr, void* stream);
int finrom_note_stream(void* stream);
int finrom_deferred_count(void);
int finrom_flush_deferred(void);
int finrom_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int finrom_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int finrom_memset(void* dst, int value, size_t bytes, void* stream);
int finrom_stream_sync(void* stream);

/* ---- per-kernel timing with HIP events on the launch stream ------------------------- *
 * When enabled every kernel launch of the library is bracketed by hipEventRecord on its
 * stream; finrom_profile_read synchronises and returns, for kernel slot `slot`
 * (0 <= slot < finrom_profile_slots()), its name, launch count and total milliseconds. */
int finrom_profile_enable(int on);
/* finrom_solve_pairs runs its two halves on two streams (default, 1) or serialised on the caller's
 * stream (0; stand-alone kernel timings).  Same results either way. */
int finrom_set_overlap(int on);
int finrom_profile_reset(void);
int finrom_profile_slots(void);
int finrom_profile_read(int slot, const char** name, int64_t* launches, double* total_ms);

/* ---- FOM: batched sparse SPD solve  A(x) w = F  +  QoI ------------------------------ *
 * Replaces Fin.forward + Fin.qoi_operator   (fom/forward_solve.py:270-291, 408-412),
 * and AffineROMFin.forward + .qoi           (rom/averaged_affine_ROM.py:237-258, 312-320)
 * for S samples at once.  The operator is a sparse-affine map on the entries of the
 * Cholesky factor's pattern:  A_e(x) = asm_c0[e] + sum_t asm_w[t] * x[asm_idx[t]],
 * t in [asm_ptr[e], asm_ptr[e+1]), which covers all three reference operators:
 *   x = nodal field k (xdim = n)       int k_h grad w.grad v dx + Bi int w v ds   fom :160-161
 *   x = 9 / 5 fin conductivities       same, through nine_param_to_function        fom :61-91
 *   x = 9 sub-fin averages             sum_i x_i A_i + Bi M                        rom :154-163
 * The symbolic phase (ordering, pattern of L, elimination schedule) is done once on the
 * host (bayesianinferencedl_amd/symbolic.py); all arrays are in the PERMUTED dof order.
 *
 * The numeric phase is a schedule interpreter: per sample it keeps a value vector
 *   G = [ L entries (nnzL, initially the assembled A_e) | 1/L_ii (n) | y, then w (n) ]
 * and executes two op streams (factorisation + L y = F, then L^T w = y).  A wave (64 samples,
 * lane = sample) fetches the global operands of chunk c+1 (fwd_chunk = 8 or 16 ops) before it executes chunk c,
 * so the host must order/pad the forward stream such that a value stored in chunk c is not loaded
 * before chunk c+2 (both streams; the backward stream runs in its own kernel, 8 ops per chunk); checked at create.  Ops (kind, a, b, d), acc = per-sample accumulator,
 * rc = the LDS row cache (cache_slots entries; holds the row being eliminated and whatever the host left in the other slots):
 *   forward  0 FMA acc -= rc[b]*G[a]   (rc[cache_slots] == -1 and rc[cache_slots+1] == 0 are constants:
 *                  "acc = A_e" is an FMA against the first, padding an FMA against the second with a = -1)
 *            3 LDX x = G[a] | 4 FMAX acc -= x*G[a]   (a row entry that does not fit the LDS cache)
 *            5 FINOFF l = acc*G[a]; G[d] = l; if b >= 0: rc[b] = l; acc = 0
 *            6 FINDIAG t = sqrt(acc); G[d] = t; G[nnzL+b] = inv = 1/t; acc = 0 (acc <= 0 flags the sample)
 *            7 YSET acc = rhs[d] | 8 FINY G[d] = acc*inv; acc = 0
 *            9 XFMA acc += imm[d]*x[b] | 10 CADD acc += imm[d]   (fused affine assembly, xdim <= 16: the stream builds
 *                  A_e = c0_e + sum_t w_t x[idx_t] itself, x sits in LDS, and there is no assembly pre-pass: n_alist = 0)
 *           11 FMALL acc -= rc[b]*rc[d]   (both operands in LDS: the host may allocate the slots of rc as a ring, so that
 *                  the entries of the rows finished just before the current one are still there; slots must have been
 *                  written by an earlier FINOFF.  No global operand, hence no chunk distance to respect.)
 *   backward 0 NOP | 1 WFMA acc -= G[a]*G[b] | 3 WSET acc = G[a] | 5 WFIN G[d] = acc*G[a]
 */
typedef struct {
  int32_t n;               /* dofs */
  int32_t nnzL;            /* entries of L */
  int32_t xdim;            /* length of one parameter vector x */
  int32_t n_obs;           /* rows of the observation operator */
  int32_t nasm;            /* entries of asm_idx / asm_w */
  int32_t n_alist;         /* entries of L that carry a value of A and are assembled by the pre-pass (0: fused stream) */
  int32_t cache_slots;     /* LDS row-cache slots the forward stream assumes (1..126): (cache_slots + 5 [+ xdim for a fused
                              stream]) * 512 B of LDS per wave decide how many interpreter waves share a CU (45 slots -> 7) */
  int32_t fwd_chunk;       /* ops per prefetch chunk of the forward stream: 8 or 16 */
  int32_t nops_fwd;        /* multiple of 2*fwd_chunk, the last 2*fwd_chunk ops are padding */
  int32_t nops_bwd;        /* multiple of 16 (two 8-op chunks), the last 16 ops are NOPs */
  const int32_t* a_list;   /* [n_alist] entry indices, ascending */
  const double*  asm_c0;   /* [nnzL] */
  const int32_t* asm_ptr;  /* [nnzL+1] */
  const int32_t* asm_idx;  /* [nasm] */
  const double*  asm_w;    /* [nasm] */
  const double*  rhs;      /* [n]   load vector F (fom :162-163), permuted */
  const int32_t* fwd_kind; const int32_t* fwd_a; const int32_t* fwd_b; const int32_t* fwd_d;   /* [nops_fwd] each */
  const int32_t* bwd_kind; const int32_t* bwd_a; const int32_t* bwd_b; const int32_t* bwd_d;   /* [nops_bwd] each */
  const int32_t* obs_ptr;  /* [n_obs+1]  CSR of B_obs (fom :215-231) over permuted dofs */
  const int32_t* obs_idx;
  const double*  obs_w;
  const int32_t* perm;     /* [n] permuted -> original dof (to return w in caller order) */
  int32_t n_imm;           /* immediates of the XFMA / CADD ops (0 if the stream has none) */
  const double*  imm;      /* [n_imm] */
} finrom_fom_desc;

int finrom_fom_create(const finrom_fom_desc* desc, finrom_fom_t* out);
void finrom_fom_destroy(finrom_fom_t h);
/* x [S x xdim] -> qoi [S x n_obs], optional w [S x n] (NULL to skip), info [S] (NULL ok) */
int finrom_fom_solve(finrom_fom_t h, const double* x, int64_t S,
                     double* qoi, double* w, int32_t* info, void* stream);

/* ---- FOM adjoint gradient (Fin.gradient, fom/forward_solve.py:293-322) ---------------------- *
 * J = 1/2 |B_obs w - data|^2 and dJ/dx_j = v^T (dA/dx_j) w with the adjoint v = -A^{-1} B_obs^T (B_obs w - data)
 * (A is symmetric: the reference's `_adj_F` is the same operator, so the stored factor is reused through a
 * second op stream).  finrom_fom_set_gradient installs, all over PERMUTED dofs:
 *   res_*    the "solve again" op stream (backward-interpreter op format, value region [nnzL+2n, nnzL+3n)),
 *   bt_*     B_obs^T as CSR by dof  (row i: observation indices and weights),
 *   g_*      for every parameter j the list of (row a, column b, weight dA_ab/dx_j) triples.
 * finrom_fom_gradient: x [S x xdim], data [n_obs] (data_per_sample = 0) or [S x n_obs] ->
 * grad [S x xdim], J [S], optional qoi [S x n_obs]; info as for finrom_fom_solve. */
typedef struct {
  int32_t nops_res;
  const int32_t* res_kind; const int32_t* res_a; const int32_t* res_b; const int32_t* res_d;
  const int32_t* bt_ptr; const int32_t* bt_obs; const double* bt_w;     /* [n+1], [nnz(B_obs)] */
  const int32_t* g_ptr; const int32_t* g_a; const int32_t* g_b; const double* g_w;   /* [xdim+1], [g_ptr[xdim]] */
} finrom_fom_grad_desc;
int finrom_fom_set_gradient(finrom_fom_t h, const finrom_fom_grad_desc* desc);
int finrom_fom_gradient(finrom_fom_t h, const double* x, const double* data, int32_t data_per_sample, int64_t S,
                        double* grad, double* J, double* qoi, int32_t* info, void* stream);

/* ---- FOM, small batches (the scalar call surface: Fin.forward for ONE conductivity, fom :270-291) --------------- *
 * The interpreter above is a throughput design (lane = sample): a lone wave takes ~12 ms per solve at n = 1597.  For small
 * batches finrom_fom_set_small installs a second, latency-oriented schedule of the same factorisation: one workgroup per
 * sample, lane = ROW of L, rows grouped into dependency levels of the elimination (row i needs the rows of its
 * structure), one barrier per level, the value vector in LDS when it fits (nnzL + 2n doubles <= 152 KiB).  All arrays over
 * PERMUTED dofs; entries of L ordered as for the interpreter streams: row-major, diagonal last in its row.
 *   row_ptr/ent_col         structure of L;  pair_ptr/pair_a/pair_b   L_e = (A_e - sum_q L[pair_a[q]] L[pair_b[q]]) (/ L_jj)
 *   asm_c0/asm_ptr/asm_idx/asm_w   A_e = c0_e + sum_t w_t x[idx_t] per entry (empty range: fill entry)
 *   col_ptr/col_ent/col_row strictly-lower entries of column j (backward substitution)
 *   lev_ptr_f/lev_rows_f, lev_ptr_b/lev_rows_b   rows of each forward / backward level
 * finrom_fom_solve and finrom_fom_gradient then use this schedule whenever S <= small_max. */
typedef struct {
  int32_t small_max;        /* largest batch solved with this schedule */
  int32_t npairs, nasm, nlev_f, nlev_b;
  const int32_t* row_ptr; const int32_t* ent_col;                           /* [n+1], [nnzL] */
  const int32_t* pair_ptr; const int32_t* pair_a; const int32_t* pair_b;    /* [nnzL+1], [npairs] x 2 */
  const double*  asm_c0; const int32_t* asm_ptr; const int32_t* asm_idx; const double* asm_w;   /* [nnzL], [nnzL+1], [nasm] x 2 */
  const int32_t* col_ptr; const int32_t* col_ent; const int32_t* col_row;   /* [n+1], [nnzL-n] x 2 */
  const int32_t* lev_ptr_f; const int32_t* lev_rows_f;                      /* [nlev_f+1], [n] */
  const int32_t* lev_ptr_b; const int32_t* lev_rows_b;                      /* [nlev_b+1], [n] */
} finrom_fom_small_desc;
int finrom_fom_set_small(finrom_fom_t h, const finrom_fom_small_desc* desc);

/* ---- FOM, frontal band sweep (the throughput path of finrom_fom_solve / finrom_solve_pairs) ------------------------ *
 * Same solve as the interpreter (Fin.forward + qoi_operator, fom/forward_solve.py:270-291, 408-412), ordered for the shape
 * of the fin: fin by fin from the tip to the root, then up the post row by row (bayesianinferencedl_amd/bandplan.py).
 * The factorisation then only touches a window of NS = B + 1 consecutive nodes, which the kernel keeps in registers
 * (lane = sample); finished columns of L are written once and read once.  Segment node g (fins: npf own nodes followed by
 * their nif interface nodes; then the post) brings three entries, diagonal, coupling to the previous node, coupling to the
 * node B back: the values of the PHYSICAL slots abmap[3g..3g+2] (slots with the same affine record may be shared; the slots
 * the fins write to must be private); slot value AB[e] = ab_c0[e] + sum_t ab_w[t] x[ab_idx[t]], t in [ab_ptr[e], ab_ptr[e+1])
 * (a pre-pass over the nAB physical slots).  A fin leaves its Schur complement on its interface nodes: entry (t, s), t >= s,
 * is added to the physical slot schur_off[f][..]; ecp_off are physical slots too.
 * Couplings longer than B make the far node an *extra* of the post sweep: act[p] = bit mask of extra slots that pivot p
 * updates (their L values are stored at lx_ptr[p]..), ent_extra[p] = slot + 1 if node p was an extra before it entered the
 * window, (ecp_slot, ecp_off) in [ecp_ptr[p], ecp_ptr[p+1]) = couplings AB[off] of node p to extras, set when p enters.
 * Supported windows: (NSF, NSP) = (3, 6), (4, 10), (5, 14) (m = 4, 8, 12; window in registers), NX <= 4, and (6, 18), (7, 22)
 * (m = 16, 20; the post's window over four waves), NX <= 8, and (8, 26), NX <= 10, (9, 30), NX <= 12 (m = 24, 28; same kernel); otherwise FINROM_ERR_UNSUPPORTED and the handle keeps using the
 * interpreter -- as it does when the load Fg is not zero on the fins' own nodes (the sweep does not carry a fin's load to its
 * interface).  finrom_fom_gradient uses the band layout once finrom_fom_set_band_gradient has installed its tables. */
typedef struct {
  int32_t NSF, NSP, NX;      /* window slots of a fin sweep / of the post sweep, extra slots */
  int32_t nfins, npf, nif;   /* fins, pivots per fin, interface nodes per fin */
  int32_t npost;             /* pivots of the post sweep; n = nfins * npf + npost */
  int32_t nAB, nterms;       /* PHYSICAL value slots, terms of their affine map */
  int32_t nLx;               /* stored extras' L values per sample (= lx_ptr[npost]) */
  const double* ab_c0; const int32_t* ab_ptr; const int32_t* ab_idx; const double* ab_w;   /* [nAB], [nAB+1], [nterms] x 2 */
  const int32_t* abmap;      /* [3 (nfins (npf + nif) + npost)] physical slot of each of the three entries of a segment node */
  const double* Fg;          /* [nfins * (npf + nif) + npost] load per segment node (0 for interface nodes inside a fin) */
  const int32_t* act; const int32_t* lx_ptr; const int32_t* ent_extra;      /* [npost], [npost+1], [npost] */
  const int32_t* ecp_ptr; const int32_t* ecp_slot; const int32_t* ecp_off;  /* [npost+1], [ecp_ptr[npost]] x 2 */
  const int32_t* schur_off;  /* [nfins][nif (nif + 1) / 2] */
  const int32_t* iface_elim; /* [nfins][nif] elimination index of each interface node */
  const int32_t* perm;       /* [n] elimination index -> dof */
  const int32_t* obs_ptr; const int32_t* obs_idx; const double* obs_w;      /* B_obs (CSR) over elimination indices */
  /* Optional (all five or none): the QoI-only form, used when a solve asks for no w (the dataset loop keeps only the
   * observables, generate_fin_dataset.py:93-100).  A fin's own nodes are leaves of the elimination and carry no load, so an
   * observation row's weights on them can ride through the fin's forward sweep as its right-hand side and leave a functional of
   * the fin's interface values; the fin's factor, y and backward sweep are then never stored or run (csrc/fom_band.hip).
   *   qoi_FgQ [as Fg]   the weights of the fin's row on the fin's segment nodes (own, then interface), Fg on the post's;
   *   qoi_row_fin [n_obs]  the fin whose functional belongs to row o, or -1; a fin belongs to at most one row;
   *   qoi_obs_*         CSR of what remains of B_obs: post nodes only, without the row's own fin's interface nodes.
   * Checked on the host against obs_*: both descriptions must be the same operator, entry by entry. */
  const double* qoi_FgQ; const int32_t* qoi_row_fin;
  const int32_t* qoi_obs_ptr; const int32_t* qoi_obs_idx; const double* qoi_obs_w;
} finrom_fom_band_desc;
int finrom_fom_set_band(finrom_fom_t h, const finrom_fom_band_desc* desc);
/* The checks finrom_fom_set_band applies before it touches the device, for a FOM of n dofs, xdim parameters and n_obs
 * observation rows: every index the kernels dereference, no missing table, npf >= NSF and npost >= NSP, and the slots the
 * fins write their Schur complements to (schur_off) private -- not shared between fins, not repeated, read by one entry
 * only (the four-wave kernel sweeps the fins of a sample block on different waves).  Host only: usable without a GPU. */
int finrom_fom_band_validate(const finrom_fom_band_desc* desc, int32_t n, int32_t xdim, int32_t n_obs);

/* The adjoint gradient on the band sweep's layout (finrom_fom_gradient for batches beyond the small-batch schedule): after the
 * full sweep the workspace holds the factor and w; A v = -B_obs^T (B_obs w - d) is solved with the stored columns (one forward
 * substitution, one more backward sweep) and grad_j = sum dA_ab/dx_j v_a w_b is contracted there.  Tables over the band plan's
 * ELIMINATION indices (finrom_fom_band_desc::perm):  bt_* = B_obs^T as CSR by elimination index (row e: observation indices and
 * weights);  g_* = for every parameter j the (a, b, dA_ab/dx_j) triples.  Needs finrom_fom_set_band; without it (or for meshes
 * without window sizes) finrom_fom_gradient keeps the interpreter's stored factor (finrom_fom_set_gradient). */
typedef struct {
  const int32_t* bt_ptr; const int32_t* bt_obs; const double* bt_w;     /* [n+1], [nnz(B_obs)] */
  const int32_t* g_ptr; const int32_t* g_a; const int32_t* g_b; const double* g_w;   /* [xdim+1], [g_ptr[xdim]] */
} finrom_fom_band_grad_desc;
int finrom_fom_set_band_gradient(finrom_fom_t h, const finrom_fom_band_grad_desc* desc);

/* General right-hand sides for the operator of each sample: out[s][k][:] = A(x_s)^-1 rhs[s][k][:], k < nrhs, rhs / out [S x nrhs x n]
 * row-major in dof order -- the incremental state and incremental adjoint solves of Fin.hessian_action (fom/forward_solve.py:
 * 344-368: `solve(self._a == rhs)` with a new right-hand side for the same conductivity).  One band sweep factors A(x_s), the
 * columns then stream back once per right-hand side (forward substitution + backward sweep on the stored factor).  Needs
 * finrom_fom_set_band (FINROM_ERR_UNSUPPORTED otherwise); info as for finrom_fom_solve. */
int finrom_fom_solve_rhs(finrom_fom_t h, const double* x, int64_t S, const double* rhs, int32_t nrhs, double* out, int32_t* info,
                         void* stream);

/* ---- which schedule ran (Fin.forward has ONE solver, fom/forward_solve.py:286; this library has several schedules of the
 * same factorisation, picked by batch size and mesh) ------------------------------------------------------------------- *
 * finrom_fom_last_path: the schedule the most recent finrom_fom_solve / finrom_fom_gradient / finrom_solve_pairs call on
 * this handle launched (FINROM_FOM_PATH_NONE before the first call).  Tests assert on it so that a dispatch change cannot
 * silently route a case away from the kernel it was written for; each path also has its own finrom_profile_* slot.
 * finrom_fom_set_small_max moves the batch-size threshold of the small-batch schedule on an existing handle (0 = never;
 * negative = error); it does not install a schedule that finrom_fom_set_small has not installed. */
#define FINROM_FOM_PATH_NONE 0
#define FINROM_FOM_PATH_SMALL_LDS 1        /* fom_small_kernel<true>: one workgroup per sample, value vector in LDS */
#define FINROM_FOM_PATH_SMALL_GLOBAL 2     /* fom_small_kernel<false>: value vector in the workspace */
#define FINROM_FOM_PATH_INTERPRETER 3      /* fom_vm_kernel + fom_bwd_kernel (schedule interpreter) */
#define FINROM_FOM_PATH_BAND_REGISTERS 4   /* fom_band_kernel: frontal band sweep, front in registers (m <= 12) */
#define FINROM_FOM_PATH_BAND_LDS_4WAVE 5   /* fom_band_ldsw_kernel: post's window over four waves + LDS exchange (m = 16 ... 28) */
#define FINROM_FOM_PATH_BAND_LDS_1WAVE 6   /* fom_band_lds_kernel: one-wave LDS window (A/B builds only) */
#define FINROM_FOM_PATH_BAND_REGISTERS_QOI 7   /* fom_band_kernel, QoI-only form: fins ride as functionals, no w */
#define FINROM_FOM_PATH_BAND_LDS_4WAVE_QOI 8   /* fom_band_ldsw_kernel, QoI-only form */
int finrom_fom_last_path(finrom_fom_t h);
int finrom_fom_set_small_max(finrom_fom_t h, int32_t small_max);

/* ---- ROM: batched LSPG reduced solve ------------------------------------------------- *
 * Replaces AffineROMFin.forward_nine_param_reduced + .qoi_reduced
 * (rom/averaged_affine_ROM.py:278-310, 323-333):
 *   psi = (sum_p theta_p A_p + Bi M) Phi ; A_r = psi^T psi ; B_r = psi^T F ;
 *   w_r = A_r^{-1} B_r ; qoi_r = (B_obs Phi) w_r.
 * The host passes the row-sparse tables Psi_p = A_p Phi (the reference's precomputed
 * `dA_dsigmak_phi`, :215-220) as one list of r-vectors ("terms"): row j of psi is
 *   sum_{t in [row_ptr[j], row_ptr[j+1])} theta[term_p[t]] * term_val[t][:]
 * with theta[0] == 1 for the constant Robin term and theta[1..P] the parameters.
 */
typedef struct {
  int32_t n;               /* rows of psi (dofs; any order) */
  int32_t r;               /* basis size */
  int32_t P;               /* number of parameters (theta has P entries per sample) */
  int32_t n_obs;
  int32_t nterms;
  const int32_t* row_ptr;  /* [n+1] */
  const int32_t* term_p;   /* [nterms] 0 = constant, 1..P = parameter index + 1 */
  const double*  term_val; /* [nterms x r] row-major */
  const double*  rhs;      /* [n]  F in the same row order */
  const double*  obs_phi;  /* [n_obs x r] B_obs Phi (rom :212) */
} finrom_rom_desc;

int finrom_rom_create(const finrom_rom_desc* desc, finrom_rom_t* out);
void finrom_rom_destroy(finrom_rom_t h);
/* HOST ONLY (no device is touched; for tests of the host logic): the GROUPED k-step tables finrom_rom_create builds for r <= 80
 * (the reference forms psi = A(theta) Phi row by row, rom/averaged_affine_ROM.py:291-297 -- here the rows are
 * sorted by the sub-domain that scales them and accumulated divided by its conductivity).  Sizes first (kmg = tvg = ext_def =
 * NULL), then the tables:  kmg [(nkg + 8) x 8] records {first slot, terms, flags (1: first coefficient is 1, 2: multiply the
 * accumulators by ext[factor] first), factor, ext indices of the <= 4 coefficients};  tvg [n_slots x 4 x rp] table rows
 * (rp = r rounded up to 16);  ext_def [n_ext x 3]: ext[l] = (theta'[a] / theta'[b]) ^ (1 + squared), theta'[0] = 1;
 * ext_final: the factor behind the last k-step.  nkg = 0: no grouped form for this descriptor. */
int finrom_rom_grouped_tables(const finrom_rom_desc* desc, int32_t* nkg, int32_t* n_ext, int32_t* ext_final, int64_t* n_slots,
                              int32_t* kmg, double* tvg, int32_t* ext_def);
/* theta [S x P] -> w_r [S x r] (NULL to skip), qoi_r [S x n_obs], info [S] (NULL ok);
 * optional A_r [S x r x r] and B_r [S x r] (the state the reference keeps in
 * self._A_r / self._B_r for its gradients, :296-297) -- NULL to skip. */
int finrom_rom_solve(finrom_rom_t h, const double* theta, int64_t S,
                     double* w_r, double* qoi_r, double* A_r, double* B_r,
                     int32_t* info, void* stream);

/* ---- offline/online form of the reduced operator (opt-in) ------------------------------------------ *
 * psi = A(theta) Phi = sum_p theta_p Psi_p (rom/averaged_affine_ROM.py:282-290; theta_0 = 1 for the Robin term), so
 *   A_r = psi^T psi = sum_{p <= q} theta_p theta_q G_pq,  G_pq = Psi_p^T Psi_q + (p != q ? Psi_q^T Psi_p : 0),
 *   B_r = psi^T F   = sum_p theta_p Psi_p^T F
 * can be assembled from blocks computed once (the reference precomputes Psi_p = A_p Phi, :215-220, and contracts per
 * sample, :291-297).  finrom_rom_set_gram installs the symmetric r x r blocks G_pq, row-major, one per listed pair
 * 0 <= p <= q <= P (0 = constant term; pairs not listed are zero; at most 64 pairs); Psi_p^T F is derived from the
 * descriptor at create.  finrom_rom_set_projection then selects which form finrom_rom_solve / _grad / finrom_solve_pairs
 * use: FINROM_PROJECTION_DIRECT (default: the per-sample psi^T psi contraction on fp64 MFMA, what the reference
 * executes) or FINROM_PROJECTION_GRAM.  Results agree to round-off (tests run both against the same oracle). */
#define FINROM_PROJECTION_DIRECT 0
#define FINROM_PROJECTION_GRAM 1
int finrom_rom_set_gram(finrom_rom_t h, int32_t npairs, const int32_t* pair_p, const int32_t* pair_q,
                        const double* G);
int finrom_rom_set_projection(finrom_rom_t h, int32_t mode);

/* ---- ROM adjoint gradient (AffineROMFin.grad_reduced, rom/averaged_affine_ROM.py:335-356) ---------- *
 * J = 1/2 |data - (B_obs Phi) w_r|^2 and its gradient with respect to the P affine parameters,
 *   g_i = (psi v_r)^T (A_i Phi w_r),   v_r = A_r^{-T} (B_obs Phi)^T (data - (B_obs Phi) w_r),
 * with psi treated as theta-independent exactly as the reference does.  (The reference then maps g to the
 * nodal field through dsigma_dk: dJ_dk = g^T S, a [P x n] product left to the caller.)
 * finrom_rom_set_gradient installs the host-precomputed blocks G_pi = (A_p Phi)^T (A_i Phi), one r x r matrix
 * per listed pair (p in 0..P with 0 = constant term, i in 0..P-1), each stored COLUMN by column.
 * finrom_rom_grad: theta [S x P], data [n_obs] (data_per_sample = 0) or [S x n_obs] (1) ->
 * J [S], g [S x P]; optional w_r [S x r], qoi_r [S x n_obs]; info as for finrom_rom_solve.  Any supported r (<= 208). */
int finrom_rom_set_gradient(finrom_rom_t h, int32_t npairs, const int32_t* pair_p, const int32_t* pair_i,
                            const double* G);
int finrom_rom_grad(finrom_rom_t h, const double* theta, const double* data, int32_t data_per_sample,
                    int64_t S, double* J, double* g, double* w_r, double* qoi_r, int32_t* info, void* stream);

/* ---- learned error model + ROM: value and gradient of the ROM+ML misfit -------------------------------------------- *
 * AffineROMFin.grad_romml (rom/averaged_affine_ROM.py:358-396) for S conductivity fields in one call:
 *   e_NN = model(k) (fp32, the reference's Keras res_bn_fc_model, deep_learning/dl_model.py:149-176, batch normalisation in
 *   inference form),  theta = Sop k,  ROM adjoint against data - e_NN  ->  loss = 1/2 |data - (qoi_r + e_NN)|^2  and
 *   grad = (dJ/dtheta)^T Sop  -  (d e_NN / d k)^T (data - (qoi_r + e_NN))     (:376-395)
 * finrom_mlp_create copies the weights (host pointers, fp32, row-major as listed); n_w <= 64, n_out <= 64.
 * finrom_romml_grad: k [S x n] device, data [n_obs] (data_per_sample = 0) or [S x n_obs]; outputs grad [S x n], loss [S],
 * optional qoi_r / e_nn [S x n_obs] (NULL to skip); Sop [P x n] device (the sub-fin averaging operator = dsigma_dk);
 * finrom_rom_set_gradient must have been called on `rom`.  info [S] (optional) is OVERWRITTEN by finrom_romml_grad -- 0, or the
 * flags of finrom_rom_solve -- so the caller need not clear it.  finrom_mlp_predict: e [S x n_out] only (:360). */
typedef struct {
  int32_t n_in, n_w, n_layers, n_out;
  const float* W0; const float* b0;            /* [n_in x n_w], [n_w]                                  y0 = x W0 + b0 */
  const float* scale; const float* shift;      /* [(n_layers + 1) x n_w]  gamma / sqrt(var + eps), beta - mean * scale; last row: head */
  const float* W; const float* b;              /* [n_layers x n_w x n_w], [n_layers x n_w]             y += elu(bn(y)) W_i + b_i */
  const float* Wh; const float* bh;            /* [n_w x n_out], [n_out]                               out = elu(bn(y)) Wh + bh */
} finrom_mlp_desc;
int finrom_mlp_create(const finrom_mlp_desc* desc, finrom_mlp_t* out);
void finrom_mlp_destroy(finrom_mlp_t h);
int finrom_mlp_predict(finrom_mlp_t h, const double* k, int64_t S, double* e, void* stream);
int finrom_romml_grad(finrom_rom_t rom, finrom_mlp_t mlp, const double* Sop, const double* k, const double* data,
                      int32_t data_per_sample, int64_t S, double* grad, double* loss, double* qoi_r, double* e_nn,
                      int32_t* info, void* stream);

/* ---- HMC trajectories on the device (BASELINE configs[4]) ----------------------------------------------------------------------- *
 * PyMC3's sampler calls the reference's value-and-gradient op once per leapfrog step (bayesian_inference/pymc_func_bayes_inverse.py:
 * 92-104 `err_grad_ROMML`, :148-167 `SqErrorOpROMML.perform`; model :186-203: potential = misfit / sigma^2 on a latent Gaussian
 * field) and does the trajectory's arithmetic itself, on the host.  Here a whole proposal is library launches on device-resident
 * chain state, so that it can be captured once in a HIP graph and replayed (bayesianinferencedl_amd/bayesian_inference/hmc.py):
 *   finrom_hmc_begin     momentum of proposal *jt of the uploaded block, H0 = U + |p|^2 / 2, Kq[0] = K, first half step of p;
 *   finrom_hmc_leapfrog  ONE leapfrog step in the four launches of finrom_romml_grad: the position update k <- k + eps p rides in
 *                        front (every kernel that reads the field forms it; Kq[step & 1] -> Kq[(step + 1) & 1]), the value and
 *                        gradient of the ROM + learned-error misfit at the new point are finrom_romml_grad's, and the momentum
 *                        update p <- p - eps c_pri dU, dU = (k - mean) + (c_lik / c_pri) grad (0 for a flagged sample), rides
 *                        behind the gradient; loss [C] and info [C] of the state are overwritten; grad_out [C x n] optional;
 *   finrom_hmc_end       after n_steps steps: last half step back, U(k) = c_lik loss + c_pri |k - mean|^2 / 2 (inf if flagged),
 *                        Metropolis test log u < H0 - H1, state update, accept counters, optional trace row, *jt += 1, *pt += 1.
 * Potential: i.i.d. Gaussian prior N(mean, 1 / c_pri) per node, likelihood scale c_lik = 1 / sigma^2.  All arrays are DEVICE
 * pointers owned by the caller; C <= 64 chains advance in lockstep (one sample of the batch each); needs finrom_rom_set_gradient,
 * the direct projection, P <= 16 and a basis the one-sample pipeline serves (r <= 96) -- FINROM_ERR_UNSUPPORTED otherwise. */
typedef struct {
  int64_t C; int32_t n;                 /* chains, nodes of the field */
  double eps, c_lik, c_pri;             /* leapfrog step, 1 / sigma^2, 1 / tau^2 */
  const double* mean;                   /* [C x n] */
  double* K; double* U; double* dU;     /* chain state: position [C x n], potential [C], grad U / c_pri [C x n] */
  double* Kq[2]; double* P; double* dUq; double* H0;   /* trajectory: positions (ping-pong) [C x n] x 2, momentum, grad U / c_pri, H0 [C] */
  const double* P_block; const double* lu_block;       /* draws of a block of B proposals: momenta [B x C x n], log-uniforms [B x C] */
  int64_t* jt; int64_t* pt;             /* device counters: proposal inside the block, proposal of the chain (trace row - 1) */
  int64_t* accept;                      /* [C] accepted proposals */
  double* trace;                        /* [(proposals + 1) x C x n] or NULL */
  double* loss; int32_t* info;          /* [C] value and flags of the last evaluation */
} finrom_hmc_state;
int finrom_hmc_begin(const finrom_hmc_state* st, void* stream);
int finrom_hmc_leapfrog(finrom_rom_t rom, finrom_mlp_t mlp, const double* Sop, const finrom_hmc_state* st, int32_t step,
                        const double* data, int32_t data_per_sample, double* grad_out, double* qoi_r, double* e_nn, void* stream);
int finrom_hmc_end(const finrom_hmc_state* st, int32_t n_steps, void* stream);

/* ---- sub-fin averages  theta = S k  (fom :466-480, rom :404-418) -------------------- *
 * Sop is the dense [P x n] averaging operator on the device (finrom_malloc + h2d). */
int finrom_subfin_avg(const double* Sop, int32_t P, int32_t n,
                      const double* k, int64_t S, double* theta, void* stream);

/* ---- Gaussian-field sampler  k = exp(0.5 * U^T xi) ----------------------------------- *
 * (deep_learning/generate_fin_dataset.py:87-88 with U = make_cov_chol(...),
 *  bayesian_inference/gaussian_field.py:9-31; U is the UPPER factor, row-major [n x n],
 *  host pointer copied at create; anything below the diagonal is an error).  xi [S x n] -> k [S x n]. */
int finrom_sampler_create(const double* U, int32_t n, finrom_sampler_t* out);
void finrom_sampler_destroy(finrom_sampler_t h);
int finrom_sampler_draw(finrom_sampler_t h, const double* xi, int64_t S, double* k, void* stream);
/* The same with xi drawn ON THE DEVICE (the reference draws np.random.randn per sample, generate_fin_dataset.py:87):
 * Philox4x32-10, counter = (global sample index, pair index), key = seed, Box-Muller; sample first_global_sample + s of
 * the stream `seed` is the same numbers whatever shard or GPU draws it.  k [S x n]; xi_out [S x n] or NULL. */
int finrom_sampler_draw_seeded(finrom_sampler_t h, uint64_t seed, int64_t first_global_sample, int64_t S, double* k,
                               double* xi_out, void* stream);

/* ---- the dataset-loop body for S samples in one call ---------------------------------- *
 * (deep_learning/generate_fin_dataset.py:93-100):  FOM solve + QoI on the caller's stream;
 * concurrently, on a stream owned by the library, theta = Sop x (sub-fin averages of the
 * field, or of the interpolated per-fin conductivities) and the LSPG reduced solve + QoI;
 * then err = qoi - qoi_r.  (Large batches of the r <= 80 / m <= 12 pairing: the FOM's band
 * sweep runs on a library stream restricted to three CUs of every shader engine -- the
 * HBM-bound sweep then shares fewer SIMDs with the projection, DESIGN.md 5 -- its pack +
 * assembly pre-pass on an unmasked stream, and when the caller's stream is a non-blocking
 * one the ROM half stays on it: no cross-stream wait on the call's critical path.
 * FINROM_FOM_CUS=0 keeps the FOM half on the caller's stream.)
 * The two halves are independent and are joined with HIP events
 * on the caller's stream (every output is ready in stream order behind the call),
 * so the HBM-bound sparse solve overlaps the MFMA-bound projection.
 * Sop: device [P x xdim] (P = the ROM's parameter count, xdim = the FOM's).  Optional
 * outputs (NULL to skip): w [S x n], w_r [S x r], theta [S x P], err [S x n_obs].
 * info [S] must be zero-initialised by the caller (bits are OR-ed in). */
int finrom_solve_pairs(finrom_fom_t fom, finrom_rom_t rom, const double* Sop,
                       const double* x, int64_t S,
                       double* qoi, double* qoi_r, double* err,
                       double* w, double* w_r, double* theta, int32_t* info, void* stream);

/* ---- the gather at the end for callers without torch.distributed ------------------------------------------------ *
 * The path shards by samples with no data-path collective; the one exchange is an all-gather of per-sample rows (QoI pairs,
 * 144 B per sample) when every rank has finished its shard -- the reference's loop (deep_learning/generate_fin_dataset.py:83-100)
 * run per rank, results collected once.  RCCL over xGMI, loaded with dlopen at the first call (no link-time dependency; a process
 * that already holds PyTorch's RCCL shares it).  Protocol: rank 0 calls finrom_comm_unique_id and hands the FINROM_COMM_ID_BYTES
 * bytes to the other ranks by whatever channel the launcher offers (file, socket, MPI, environment); every rank, with ITS device
 * current (finrom_set_device), calls finrom_comm_init; finrom_gather(send [count], recv [nranks x count]) is an all-gather of
 * `count` doubles per rank on DEVICE buffers, asynchronous on `stream`, rank r's block at recv + r * count.
 * One process per GPU (RCCL rejects two ranks on one device).  bench.py and bayesianinferencedl_amd/distributed.py keep using
 * torch.distributed when torch is the launcher; both routes end in ncclAllGather. */
#define FINROM_COMM_ID_BYTES 128
typedef struct finrom_comm_s* finrom_comm_t;
int finrom_comm_unique_id(void* id_out);
int finrom_comm_init(finrom_comm_t* out, int32_t rank, int32_t nranks, const void* id);
int finrom_gather(finrom_comm_t comm, const double* send, int64_t count, double* recv, void* stream);
int finrom_comm_destroy(finrom_comm_t comm);

/* ---- elementwise helper: err = qoi - qoi_r (generate_fin_dataset.py:99) -------------- */
int finrom_sub(const double* a, const double* b, int64_t count, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FINROM_H */
