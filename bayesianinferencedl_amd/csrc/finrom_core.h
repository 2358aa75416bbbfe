// Internal declarations shared by ALL translation units of libfinrom_hip.so (gfx950): errors, timers, stream capture and device
// memory, the FOM's device structures.  (finrom_internal.h adds the reduced model's and the error model's: a change there does
// not rebuild the band sweep's translation units, minutes each.)
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

}  // namespace finrom
