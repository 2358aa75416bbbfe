// Internal declarations shared by the translation units of libfinrom_hip.so (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>
#include "finrom.h"

namespace finrom {

constexpr int WAVE = 64;
constexpr int FOM_ROW_CACHE_UNUSED = 36;   // LDS slots (512 B each) caching the row being eliminated: 18 KiB per wave, 7 waves per CU leave LDS for the ROM kernels

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);

#define FR_HIP(call)                                        \
  do {                                                      \
    hipError_t _e = (call);                                 \
    if (_e != hipSuccess) return finrom::hip_fail(_e, #call); \
  } while (0)

// ---- per-kernel HIP-event timing (finrom_profile_*) ------------------------------------
enum KernelSlot {
  K_PACK = 0,      // row-major [S x d] -> sample-blocked [S/64][d][64]
  K_FOM_ASM,       // affine assembly of the entries of A (pre-pass of the interpreter)
  K_FOM,           // sparse Cholesky + two triangular solves + QoI (schedule interpreter)
  K_UNPACK_W,      // blocked w -> row-major, original dof order
  K_ROM_PROJ,      // psi build + psi^T psi (fp64 MFMA) + psi^T F
  K_ROM_SOLVE,     // dense Cholesky solve of the reduced system + QoI
  K_AVG,           // theta = S k
  K_SAMPLER,       // k = exp(0.5 U^T xi)
  K_MISC,
  // one slot per FOM schedule (finrom_fom_last_path); a launch timed here is ALSO added to K_FOM, the aggregate bench.py reads
  K_FOM_PATH_SMALL,
  K_FOM_PATH_INTERP,
  K_FOM_PATH_BAND_REG,
  K_FOM_PATH_BAND_LDSW,
  K_NUM
};
struct ScopedKernelTimer {
  int slot; hipStream_t s; hipEvent_t e0 = nullptr, e1 = nullptr;
  ScopedKernelTimer(int slot, hipStream_t s);
  ~ScopedKernelTimer();
};

// ---- stream capture and device memory (finrom_api.hip) ---------------------------------------------------------------------
// While a stream is being captured into a HIP graph (torch.cuda.graph: global capture mode) hipFree / hipMalloc / a synchronous
// hipMemset from ANY thread are "unsafe" calls that invalidate the capture or wait on it.  The library therefore never issues
// one while it knows of an open capture: frees and handle destructions are queued (dev_free, defer_or_run) and flushed by the
// next call that finds no capture open; allocations and workspace growth fail with FINROM_ERR_UNSUPPORTED instead.
// What it knows: every stream an entry point was handed (CallGuard) or was told about (finrom_note_stream) is queried with
// hipStreamIsCapturing and remembered while it captures.
bool note_stream(hipStream_t st);          // query st, update the registry; true = st is under capture
bool any_capture();                        // re-queries the remembered streams
bool call_captures();                      // the entry point on this thread's stack was handed a capturing stream
void dev_free(void* p);                    // hipFree now, or queued while a capture is open
void defer_or_run(void (*fn)(void*), void* arg);   // the same for any other capture-unsafe cleanup (stream / event destruction)
void flush_deferred();                     // runs the queue if no capture is open
int deferred_count();
struct CallGuard {                         // first statement of every entry point that takes a stream
  bool prev;
  explicit CallGuard(hipStream_t st);
  ~CallGuard();
};

template <class T>
int upload(T** dptr, const T* host, size_t count) {
  *dptr = nullptr;
  if (count == 0) return 0;
  if (any_capture()) { set_error("device tables cannot be created while a stream capture is open"); return FINROM_ERR_UNSUPPORTED; }
  hipError_t e = hipMalloc((void**)dptr, count * sizeof(T));
  if (e != hipSuccess) { set_error("hipMalloc failed (" + std::to_string(count * sizeof(T)) + " bytes)"); return FINROM_ERR_NOMEM; }
  FR_HIP(hipMemcpy(*dptr, host, count * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device's function object: once per device, and
// safe when two host threads launch on different handles (a lost race only repeats the idempotent call)
struct PerDeviceOnce {
  std::atomic<unsigned long long> done{0};
  template <class F> int run(F&& f) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return f();
    const unsigned long long bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return 0;
    const int rc = f();
    if (rc == 0) done.fetch_or(bit, std::memory_order_release);
    return rc;
  }
};

// grow-only device scratch buffer owned by a handle.  A buffer that was used by a call under capture is referenced by that graph
// for as long as the graph lives: when such a buffer has to grow later, the old allocation is retired (freed with the handle)
// instead of freed; growth DURING a capture is refused (the same call, run once before the capture, sizes the workspace).
struct Scratch {
  void* p = nullptr; size_t cap = 0;
  bool captured = false;
  std::vector<void*> retired;
  int reserve(size_t bytes);
  void release();
};

// FINROM_TRACE diagnostic: workgroup residency trace (start/end on the 100 MHz real-time counter + where it ran)
__device__ __forceinline__ void trace_begin(long long* tr, int64_t wg) {
  if (tr != nullptr && threadIdx.x == 0) {
    tr[wg * 6 + 0] = __builtin_amdgcn_s_memrealtime();
    tr[wg * 6 + 2] = __builtin_amdgcn_s_getreg(4 | (31 << 11));      // HW_REG_HW_ID
    tr[wg * 6 + 3] = __builtin_amdgcn_s_getreg(20 | (31 << 11));     // HW_REG_XCC_ID
    tr[wg * 6 + 4] = __builtin_amdgcn_s_memtime();                   // shader clock ticks
  }
}
__device__ __forceinline__ void trace_end(long long* tr, int64_t wg) {
  if (tr != nullptr && threadIdx.x == 0) { tr[wg * 6 + 1] = __builtin_amdgcn_s_memrealtime(); tr[wg * 6 + 5] = __builtin_amdgcn_s_memtime(); }
}

// ---- FOM ------------------------------------------------------------------------------
constexpr int FOM_MAX_FUSED_X = 16;   // longest parameter vector the interpreter keeps in LDS (fused assembly)
constexpr int VM_CHUNK = 8;        // ops whose global operands are fetched together, one chunk ahead
struct FomDev {
  long long* trace;   // FINROM_TRACE: per-workgroup {start, end (10 ns ticks), HW_ID, XCC_ID, start, end (shader clock)}; nullptr = off
  int debug_phases;   // bit 0 factor+forward, 1 backward, 2 QoI (FINROM_FOM_PHASES, timing experiments only; default 7)
  int n, nnzL, xdim, n_obs, n_alist, cache_slots, fwd_chunk;
  int gsize;                         // values per sample: nnzL + 2n  (L | 1/L_ii | y,w)
  int nchunks_fwd, nchunks_bwd;      // executed chunks (the streams carry 2 more chunks of NOP padding)
  const int* asm_rec_i; const double* asm_rec_d;       // [n_alist][8] / [n_alist][5] fixed-size assembly records
  const int* asm_idx; const double* asm_w;            // terms beyond the first four of an entry
  const double* rhs;
  const int* f_a; const int* f_kb; const int* f_d;     // forward stream: load byte offset, LDS byte offset of rc[b] (FMA, FMALL) or kind | (b+1) << 8, d (FMA / FMALL: LDS byte offset of the second LDS operand)
  const double* f_imm;                                 // immediates of the fused-assembly ops
  int fused;                                           // the forward stream assembles A itself (x in LDS, no pre-pass)
  const int* f_mask;                                   // per chunk: bit u set = slot u is not a plain multiply-add
  const int* b_a; const int* b_b; const int* b_kd;     // backward stream: byte offsets of the two operands, kind | d << 8
  const int* obs_ptr; const int* obs_idx; const double* obs_w; const int* perm;
  // adjoint gradient (finrom_fom_set_gradient); the value vector then has a 4th region v at nnzL + 2n
  int has_grad, nchunks_res;
  const int* r_a; const int* r_kb; const int* r_d;
  const int* bt_ptr; const int* bt_obs; const double* bt_w;
  const int* g_ptr; const int* g_a; const int* g_b; const double* g_w;
};
// The small-batch kernel reads a stream of 256-B records, one per entry of L (entries with more than 16 index pairs or more
// than 8 assembly terms continue in further records): header, up to 8 terms of A_e = c0 + sum_t w_t x[idx_t], 16 index
// pairs (lane l of a 16-lane group reads pair l).  The address of a record depends on its number only and the loop over
// the records of a row contains no other global load, so records several steps ahead can be in flight.
struct FomSmallRec {
  int e;            // entry of L this record belongs to
  int npair;        // index pairs in THIS record (<= 16)
  int col;          // column of the entry (== row for the diagonal)
  int flags;        // bit 0: last record of its entry, bit 1: diagonal entry, bit 2: has assembly terms
  double c0;
  int aidx[8]; double aw[8];
  int2 first[16];
  int pad[2];
};
static_assert(sizeof(FomSmallRec) == 256, "record layout");
struct FomSmallDev {                 // latency-oriented schedule for small batches (finrom_fom_set_small)
  int small_max = 0, nlev_f = 0, nlev_b = 0, in_lds = 0;
  const int* row_ptr = nullptr; const int* ent_col = nullptr;   // structure of L (adjoint substitution)
  const int* rec_ptr = nullptr;                         // [n+1] records of row i
  const FomSmallRec* rec = nullptr;                     // [nrec + 4] (padded for the lookahead)
  const int* col_ptr = nullptr; const int2* colv = nullptr;   // [nnzL - n] (entry, row)
  const int* lev_ptr_f = nullptr; const int* lev_rows_f = nullptr; const int* lev_ptr_b = nullptr; const int* lev_rows_b = nullptr;
};
struct FomSmallGrad {                // adjoint-gradient stage of the small-batch kernel (finrom_fom_gradient, S <= small_max)
  const double* data = nullptr; int64_t data_stride = 0;   // observations [n_obs] (stride 0) or [S x n_obs]
  double* grad = nullptr; double* J = nullptr;              // [S x xdim], [S]
};
int launch_fom_small(const FomDev& p, const FomSmallDev& q, const double* x, int64_t S, double* Gscratch, double* qoi, double* w,
                     int* info, hipStream_t st, const FomSmallGrad& g = FomSmallGrad());
int launch_fom_adjoint(const FomDev& p, int64_t nblk, int64_t S, double* Gw, const double* qoi, const double* data,
                       int64_t data_stride, double* gradT, double* J, hipStream_t st);
int launch_unpack(const double* srcT, int64_t S, int d, int64_t blk_stride, int off, const int* perm, double* dst, hipStream_t st);
int launch_pack(const double* x, int64_t S, int d, double* xT, hipStream_t st);
int launch_fom_assemble(const FomDev& p, const double* xT, int64_t nblk, double* Gw, hipStream_t st);
int launch_fom(const FomDev& p, const double* xT, int64_t nblk, int64_t S, double* Gw, double* qoi, int* info, hipStream_t st);
int launch_unpack_w(const FomDev& p, const double* Gw, int64_t S, double* w, hipStream_t st);

// frontal band sweep (fom_band.hip, finrom_fom_set_band): per-sample workspace [AB | L | Lx | y -> w], sample-blocked
struct BandDev {
  int on = 0;
  int n, n_obs, xdim, gsize, nAB, nL, nLx, NSF, NSP, NX, nfins, npf, nif, npost, post_g0, post_e0, post_L0;
  int offL, offLx, offY, offX;   // offX: the extras' state of the LDS-window variant (band_xsize(NSP) doubles per lane)
  int offV;                      // adjoint right-hand side -> adjoint solution (n doubles; finrom_fom_gradient)
  const int* abmap;          // [3 G] physical value slot of each entry of a segment node
  const double* Fg; const int* act; const int* lx_ptr; const int* ent_extra; const int* ecp_ptr; const int* ecp_slot; const int* ecp_off;
  const int* schur_off; const int* iface_elim; const int* obs_ptr; const int* obs_idx; const double* obs_w; const int* perm;
  // QoI-only form (fom_band.hip): qo = its tables are installed
  int qo = 0;
  const double* FgQ = nullptr;       // [G] observation weights on the fins' segment nodes, the load on the post's
  const int* qobs_ptr = nullptr; const int* qobs_idx = nullptr; const double* qobs_w = nullptr;   // post-only remainder of B_obs
  const int* row_fin = nullptr;      // [n_obs] fin whose functional belongs to the row, -1: none
};
struct BandGradDev {              // adjoint gradient on the band layout (finrom_fom_set_band_gradient)
  int on = 0;
  const int* bt_ptr = nullptr; const int* bt_obs = nullptr; const double* bt_w = nullptr;     // B_obs^T, CSR by elimination index
  const int* g_ptr = nullptr; const int* g_a = nullptr; const int* g_b = nullptr; const double* g_w = nullptr;   // per parameter: (a, b, dA_ab/dx_j)
};
int launch_fom_band_adjoint(const BandDev& p, const BandGradDev& g, double* Gw, int64_t nblk, int64_t S, const double* qoi,
                            const double* data, int64_t data_stride, double* gradT, double* J, hipStream_t st);
int launch_fom_band_resolve(const BandDev& p, double* Gw, int64_t nblk, int nrhs, const double* rhsT, double* outT, hipStream_t st);
constexpr int BAND_LDS_XSIZE = 256;                 // doubles per lane of the extras' workspace slice ...
constexpr int BAND_LDS_XSIZE_WIDE = 384;            // ... and for the windows beyond NSP = 22 (ten extras of 26 slots + their scalars)
constexpr int band_xsize(int NSP) { return NSP > 26 ? 512 : NSP > 22 ? BAND_LDS_XSIZE_WIDE : BAND_LDS_XSIZE; }   // (NSP = 30, twelve extras: 462)
bool band_supported(int NSF, int NSP, int NX);
int launch_fom_band(const BandDev& p, double* Gw, int64_t nblk, int64_t S, double* qoi, int* info, hipStream_t st, bool qoi_only);
int launch_fom_band_wide(const BandDev& p, double* Gw, int64_t nblk, int64_t S, double* qoi, int* info, hipStream_t st, bool qo);   // fom_band_wide.hip
int band_path(const BandDev& p, bool qoi_only);      // FINROM_FOM_PATH_* of the kernel launch_fom_band picks

// ---- learned error model (mlp_kernels.hip, finrom_mlp_*) ----------------------------------
struct MlpDev {
  int n_in, n_w, n_layers, n_out;
  const float* W0; const float* b0;            // [n_in x n_w], [n_w]
  const float* scale; const float* shift;      // [(n_layers + 1) x n_w] inference-form batch normalisation (last row: the head)
  const float* W; const float* b;              // [n_layers x n_w x n_w], [n_layers x n_w]
  const float* Wh; const float* bh;            // [n_w x n_out], [n_out]
};
// the error model's forward pass riding as ONE MORE workgroup per sample in rom_small_proj_kernel (finrom_romml_grad's one-sample
// form: the ROM's contraction does not depend on it, the solve kernel behind both is the first to need its output)
struct MlpFuse {
  int on = 0;
  MlpDev m{};
  const double* k = nullptr; const double* data = nullptr; int64_t data_stride = 0;
  float* tape = nullptr; double* e_out = nullptr; double* data_shift = nullptr;
  // the sub-fin averages theta = S k too: every contraction workgroup forms its sample's theta itself (the same sums in the
  // same order, into its own 16 doubles of theta_scr [S x NC x 16], which its scalar loads then read), workgroup 0 also writes
  // theta_out [S x P] for the kernels behind
  const double* Sop = nullptr; int P = 0; double* theta_scr = nullptr; double* theta_out = nullptr;
};
int launch_mlp_forward(const MlpDev& m, const double* k, int64_t S, const double* data, int64_t data_stride, float* tape,
                       double* e_out, double* data_shift, hipStream_t st, const double* Sop = nullptr, int P = 0,
                       double* theta_out = nullptr);
// (g_parts, n_parts: instead of g_theta, its n_parts partial sums [S x n_parts x 32] as rom_grad_contract_small_kernel leaves them)
int launch_mlp_backward(const MlpDev& m, int64_t S, const float* tape, const double* data, int64_t data_stride, const double* qoi_r,
                        const double* e_nn, const double* g_theta, const double* Sop, int P, double* grad, hipStream_t st,
                        const double* g_parts = nullptr, int n_parts = 0, const float* g0_in = nullptr);
// (g0_in [S x 64]: the walk back through the head and the hidden layers has been done -- MlpBackFuse -- and left its result there)
struct MlpBackFuse {             // that walk as one more workgroup per sample of rom_grad_contract_small_kernel
  int on = 0;
  MlpDev m{};
  const float* tape = nullptr; const double* data = nullptr; int64_t data_stride = 0;
  const double* qoi_r = nullptr; const double* e_nn = nullptr; float* g0_out = nullptr;
};

// ---- ROM ------------------------------------------------------------------------------
constexpr int ROM_MAX_PHASES = 8;
struct RomDev {
  int n, r, rp, NB, P, n_obs;           // rp = 16*NB padded basis size
  int solve_in_lds;                     // packed factor fits in LDS (rp <= 176)
  long long* trace;                     // FINROM_TRACE (see FomDev::trace)
  int clock_probe;                      // FINROM_CLOCK_PROBE: a few workgroups print their shader clock (diagnostic)
  // psi tables, rows grouped 4 per k-step and sorted by term count into phases of constant NT
  int n_phases;
  int phase_nt[ROM_MAX_PHASES], phase_ks0[ROM_MAX_PHASES], phase_ks1[ROM_MAX_PHASES], phase_slot0[ROM_MAX_PHASES];
  // the same tables cut into LDS-sized chunks of whole k-steps (LDS-staged kernel, NB <= 5)
  int n_chunks;
  const int* ch_nt; const int* ch_nks; const int* ch_off; const int* ch_bytes;   // [n_chunks]; ch_off in doubles
  const double* tvc;                    // chunk images: values then theta indices, each padded to 1 KiB
  // the same rows regrouped for the single-wave kernels so that the four rows of a k-step share ONE list of theta indices
  // (rows with the same term pattern together; leftovers merged under the union of their patterns, absent terms = zero
  // rows): theta then comes from SGPRs, and the k-step needs no theta-index loads, no LDS reads and no per-lane addresses
  int n_uphases;
  int uphase_nt[ROM_MAX_PHASES], uphase_ks0[ROM_MAX_PHASES], uphase_ks1[ROM_MAX_PHASES], uphase_slot0[ROM_MAX_PHASES];
  const double* tvu;                    // [(nuslots + 4) * 4 * rp]
  const int* kpat;                      // [nuslots + 16] theta index of each slot (0 = the constant 1)
  int tvu_bytes;
  int nku;                              // number of pattern-uniform k-steps (even; sorted by term count, descending; every
                                        // term-count group holds an even number of k-steps, padded with a zero k-step)
  int uend[4];                          // uend[t] = number of k-steps with more than t terms (uend[0] = nku)
  const int* kmeta;                     // [(nku + 8) * 8] per k-step: first slot, term count, 2 x padding, theta indices of its 4 slots
  const double* tv;                     // [(nslots + 4) * 4 * rp]  padded r-vectors, slot-major
  const int* pidx;                      // [(nslots + 4) * 4]       theta index of each r-vector (0 = constant 1)
  // rows with a non-zero load F (root nodes), same slot format with a runtime term count
  int rhs_nk, rhs_nt;
  const double* rhs_tv; const int* rhs_pidx; const double* rhs_f;
  const double* obs_phi;                // [n_obs x r]
};
struct RomGradArgs {                      // adjoint-gradient stage of rom_solve_kernel (finrom_rom_grad)
  const double* data = nullptr; int64_t data_stride = 0;    // observations [n_obs] (stride 0) or [S x n_obs]
  const double* theta = nullptr;                            // [S x P]
  double* J = nullptr; double* g = nullptr;                 // [S], [S x P]
  int npairs = 0; const int* pair_p = nullptr; const int* pair_i = nullptr; const double* Gt = nullptr;
  double* gpart = nullptr; int* ticket = nullptr;   // small batches (rom_grad_contract_small_kernel): [S x NG x 32] partial sums, [S] arrival counters (zero between calls)
  int defer_sum = 0;        // (small batches) leave the NG partial sums in gpart: the consumer adds them (mlp_backward_kernel in finrom_romml_grad)
  int info_store = 0;       // (one-sample solve kernel) info[s] is stored (0 or 2), not or-ed into: the caller did not have to clear it
  double* vw = nullptr;     // scratch [S x 2 rp]: when set, the substitution kernel leaves v_r and w_r there and the
                            // contraction g_i = sum_p theta_p v_r^T G_pi w_r runs in rom_grad_contract_kernel (fp64 MFMA, 16 samples per wave)
};
// offline/online form of the reduced operator (finrom_rom_set_gram, rom_gram.hip)
constexpr int ROM_GRAM_MAX_PAIRS = 64;
struct RomGramDev {
  int npairs = 0;
  const int* pair_p = nullptr; const int* pair_q = nullptr;   // [npairs], 0 = the constant term
  const double* Gt = nullptr;     // [npairs][NB(NB+1)/2][64][4]  tiles in the MFMA C/D register layout (r <= 96)
  const double* Gp = nullptr;     // [npairs][rp(rp+1)/2]         packed upper triangle, row by row (r > 96)
  const double* h = nullptr;      // [P+1][rp]                    h_p = Psi_p^T F
};
int launch_rom_gram(const RomDev& p, const RomGramDev& gm, const double* theta, int64_t S, double* Ar, double* Br, int factor,
                    int* info, hipStream_t st, double* w_r = nullptr, double* qoi_r = nullptr);
int launch_rom_grad(const RomDev& p, const double* Ar, const double* Br, int64_t S, double* w_r, double* qoi_r,
                    int* info, const RomGradArgs& ga, hipStream_t st);
int launch_rom_grad_contract(const RomDev& p, int64_t S, const RomGradArgs& ga, hipStream_t st);
int launch_rom_chol_blocked(const RomDev& p, double* Ar, int64_t S, int* info, hipStream_t st);
int launch_rom_proj(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info, hipStream_t st,
                    double* w_r = nullptr, double* qoi_r = nullptr, int* cu_ticket = nullptr);
constexpr int ROM_SPLITK_MAX_S = 64;      // batches up to this size take the split-K projection kernel (r = 49..96)
int launch_rom_proj_wide(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                         hipStream_t st, double* w_r, double* qoi_r);
bool rom_splitk_applies(const RomDev& p, int64_t S);
// one-sample call patterns (rom_onesample.hip): contraction over several workgroups per sample + MFMA-form solve kernel
bool rom_onesample_applies(const RomDev& p, int64_t S);
size_t rom_onesample_scratch_bytes(const RomDev& p, int64_t S);
int launch_rom_onesample(const RomDev& p, const double* theta, int64_t S, double* part, int grad, const RomGradArgs& ga, double* w_r,
                         double* qoi_r, int* info, hipStream_t st, const MlpFuse* fuse = nullptr);
constexpr int ROM_GRAD_SMALL_NG = 36;      // blocks per sample in rom_grad_contract_small_kernel
int launch_rom_grad_contract_small(const RomDev& p, int64_t S, const RomGradArgs& ga, hipStream_t st, const MlpBackFuse* bf = nullptr);
int launch_rom_proj_splitk(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                           hipStream_t st, double* w_r, double* qoi_r, const RomGradArgs& ga);
int launch_rom_proj_single(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                           hipStream_t st, double* w_r, double* qoi_r, int* cu_ticket);   // cu_ticket: 4096 ints of device scratch or nullptr
int launch_rom_solve(const RomDev& p, const double* Ar, const double* Br, int64_t S, double* w_r,
                     double* qoi_r, double* Ar_out, double* Br_out, int* info, int factored, hipStream_t st);

}  // namespace finrom
