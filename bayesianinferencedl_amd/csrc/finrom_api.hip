// extern "C" surface of libfinrom_hip.so (see include/finrom.h for the contract).
#include "finrom_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <numeric>

namespace finrom {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int hip_fail(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")";
  return FINROM_ERR_HIP;
}

// ---- stream capture registry, deferred frees (finrom_internal.h) ----------------------------------------------------------
static std::mutex g_cap_mu;
static std::vector<hipStream_t> g_cap_streams;          // streams last seen under capture
struct DeferredCall { void (*fn)(void*); void* arg; };
static std::vector<DeferredCall> g_deferred;            // capture-unsafe cleanups waiting for the capture to end
static thread_local bool tl_call_captures = false;

static bool query_capturing(hipStream_t st) {
  if (st == nullptr) return false;                      // the legacy default stream cannot be captured (and must not be queried
                                                        //  while another stream captures: that is itself an "implicit" violation)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const hipError_t e = hipStreamIsCapturing(st, &cs);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }      // a stream that no longer exists is not capturing
  return cs != hipStreamCaptureStatusNone;              // (an invalidated capture is still open until EndCapture)
}
bool note_stream(hipStream_t st) {
  const bool cap = query_capturing(st);
  if (st == nullptr) return false;
  std::lock_guard<std::mutex> lk(g_cap_mu);
  auto it = std::find(g_cap_streams.begin(), g_cap_streams.end(), st);
  if (cap && it == g_cap_streams.end()) g_cap_streams.push_back(st);
  if (!cap && it != g_cap_streams.end()) g_cap_streams.erase(it);
  return cap;
}
static bool any_capture_locked() {
  for (size_t i = 0; i < g_cap_streams.size();) {
    if (query_capturing(g_cap_streams[i])) ++i;
    else g_cap_streams.erase(g_cap_streams.begin() + i);
  }
  return !g_cap_streams.empty();
}
bool any_capture() { std::lock_guard<std::mutex> lk(g_cap_mu); return any_capture_locked(); }
bool call_captures() { return tl_call_captures; }
static void hip_free_cb(void* p) { (void)hipFree(p); }
void defer_or_run(void (*fn)(void*), void* arg) {
  {
    std::lock_guard<std::mutex> lk(g_cap_mu);
    if (any_capture_locked()) { g_deferred.push_back({fn, arg}); return; }
  }
  fn(arg);
}
void dev_free(void* p) { if (p) defer_or_run(hip_free_cb, p); }
void flush_deferred() {
  std::vector<DeferredCall> run;
  {
    std::lock_guard<std::mutex> lk(g_cap_mu);
    if (g_deferred.empty() || any_capture_locked()) return;
    run.swap(g_deferred);
  }
  for (auto& d : run) d.fn(d.arg);
}
int deferred_count() { std::lock_guard<std::mutex> lk(g_cap_mu); return (int)g_deferred.size(); }
CallGuard::CallGuard(hipStream_t st) : prev(tl_call_captures) {
  const bool cap = note_stream(st);
  tl_call_captures = prev || cap;
  if (!tl_call_captures) flush_deferred();
}
CallGuard::~CallGuard() { tl_call_captures = prev; }

int Scratch::reserve(size_t bytes) {
  if (bytes <= cap) { captured = captured || tl_call_captures; return 0; }
  if (tl_call_captures || any_capture()) {
    set_error("a workspace of " + std::to_string(bytes) + " bytes is needed while a stream capture is open: run the same call once "
              "(same handle, same batch size) before the capture begins");
    return FINROM_ERR_UNSUPPORTED;
  }
  if (p) {
    if (captured) retired.push_back(p);        // a graph captured earlier still points at it: keep it until the handle goes
    else (void)hipFree(p);
    p = nullptr; cap = 0; captured = false;
  }
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) { p = nullptr; set_error("hipMalloc of " + std::to_string(bytes) + " scratch bytes failed"); return FINROM_ERR_NOMEM; }
  cap = bytes;
  return 0;
}
void Scratch::release() {
  dev_free(p);
  for (void* q : retired) dev_free(q);
  retired.clear();
  p = nullptr; cap = 0; captured = false;
}

// ---- profiling -------------------------------------------------------------------------
static const char* kSlotNames[K_NUM] = {"pack", "fom_assemble", "fom_chol_solve", "unpack_w", "rom_proj_mfma",
                                        "rom_reduced_solve", "subfin_avg", "sampler_gemm_exp", "misc",
                                        "fom_path_small", "fom_path_interpreter", "fom_path_band_registers", "fom_path_band_lds_4wave"};
struct Pending { int slot; hipEvent_t e0, e1; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<Pending> g_pending;
static std::vector<hipEvent_t> g_pool;
static bool g_overlap = true;
static double g_ms[K_NUM];
static int64_t g_cnt[K_NUM];

static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
ScopedKernelTimer::ScopedKernelTimer(int slot_, hipStream_t s_) : slot(slot_), s(s_) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_on) return;
  // a stream that is being captured into a HIP graph records nothing now: an event recorded there becomes a graph node and can
  // never be waited for or timed from the host (hmc.run_chains_device captures the one-sample call sequence)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
  e0 = get_event(); e1 = get_event();
  if (e0 && e1) (void)hipEventRecord(e0, s);
}
ScopedKernelTimer::~ScopedKernelTimer() {
  if (!e0 || !e1) return;
  (void)hipEventRecord(e1, s);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_pending.push_back({slot, e0, e1});
}
static void drain_pending() {
  for (auto& p : g_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      g_ms[p.slot] += ms; g_cnt[p.slot] += 1;
      if (p.slot >= K_FOM_PATH_SMALL && p.slot <= K_FOM_PATH_BAND_LDSW) { g_ms[K_FOM] += ms; g_cnt[K_FOM] += 1; }
    }
    g_pool.push_back(p.e0); g_pool.push_back(p.e1);
  }
  g_pending.clear();
}

int launch_subfin_avg(const double*, int, int, const double*, int64_t, double*, hipStream_t);
int launch_sampler(const double*, int, const double*, int64_t, double*, hipStream_t);
int launch_sub(const double*, const double*, int64_t, double*, hipStream_t);
int launch_philox_normal(unsigned long long, int64_t, int64_t, int, double*, hipStream_t);

}  // namespace finrom

using namespace finrom;

struct finrom_fom_s {
  FomDev d{};
  FomSmallDev small{};
  BandDev band{};                      // frontal band sweep (finrom_fom_set_band); band.on = 0: interpreter
  BandGradDev band_grad{};             // adjoint gradient on the band layout (finrom_fom_set_band_gradient)
  FomDev band_asm{};                   // parameters of the band sweep's assembly pre-pass (fom_assemble_kernel)
  std::vector<void*> owned;
  Scratch xT, Gw, gradT, qtmp;
  int last_path = FINROM_FOM_PATH_NONE;   // finrom_fom_last_path
};
struct finrom_rom_s {
  RomDev d{};
  std::vector<void*> owned;
  Scratch Ar, Br, theta, qtmp, vw, ticket, grad_ticket, part, ext;
  int g_npairs = 0; const int* g_pair_p = nullptr; const int* g_pair_i = nullptr; const double* g_Gt = nullptr;
  RomGramDev gram;                     // offline/online form (finrom_rom_set_gram); gram.h is filled at create
  int projection = FINROM_PROJECTION_DIRECT;
  hipStream_t side = nullptr;          // library-owned stream for the ROM half of finrom_solve_pairs / the error model of finrom_romml_grad
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // the FOM half of finrom_solve_pairs on a stream restricted to a SUBSET of the CUs (hipExtStreamCreateWithCUMask): the sweep's
  // waves then share SIMDs with the projection on those CUs only, and run faster for being fewer.  cus < 0: per_se = -cus
  // CUs in every shader engine (the mask's bit i is CU i / n_se of shader engine i % n_se -- measured: the first 32 c bits give the
  // sweep the same speed as any 32 c' <= bits < 32 (c + 1), and bit patterns that fill whole engines starve the dispatcher)
  hipStream_t fom_side = nullptr; hipEvent_t ev_join_fom = nullptr; int fom_side_cus = 0;
  int ensure_fom_side(int cus) {
    if (fom_side_cus == cus && (fom_side || cus != 0)) return 0;      // (made, or refused by the runtime once: do not ask again)
    if (call_captures() || any_capture()) { set_error("the FOM side stream cannot be created while a stream capture is open"); return FINROM_ERR_UNSUPPORTED; }
    if (fom_side) { (void)hipStreamDestroy(fom_side); fom_side = nullptr; }
    int dev = 0; hipDeviceProp_t pr;
    FR_HIP(hipGetDevice(&dev));
    FR_HIP(hipGetDeviceProperties(&pr, dev));
    const int ncu = pr.multiProcessorCount;
    const int want = cus;
    if (cus < 0) cus = -cus * (ncu / 8);                 // (eight CUs per shader engine on gfx950)
    if (cus < 1 || cus > ncu) { set_error("FINROM_FOM_CUS out of range"); return FINROM_ERR_ARG; }
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    const int mode = getenv("FINROM_FOM_CUS_MODE") != nullptr ? atoi(getenv("FINROM_FOM_CUS_MODE")) : 1;
    for (int t = 0; t < cus; ++t) {
      int b = (int)((int64_t)t * ncu / cus);                                     // mode 0: spread evenly over the bit array
      if (mode == 1) b = t;                                                       // mode 1: the first `cus` bits
      if (mode == 2) b = (t % 8) * (ncu / 8) + t / 8;                            // mode 2: round robin over eight blocks of ncu / 8 bits
      if (mode == 3) b = (t % 8) * 32 + t / 8;                                    // mode 3: round robin over 32-bit words (one per XCD?)
      mask[b >> 5] |= 1u << (b & 31);
    }
    // (a runtime that refuses the mask is not an error of the call: the FOM half then stays on the caller's stream, as before)
    if (hipExtStreamCreateWithCUMask(&fom_side, (uint32_t)mask.size(), mask.data()) != hipSuccess) { (void)hipGetLastError(); fom_side = nullptr; fom_side_cus = want; return 0; }
    if (!ev_join_fom) FR_HIP(hipEventCreateWithFlags(&ev_join_fom, hipEventDisableTiming));
    fom_side_cus = want;
    return 0;
  }
  int ensure_side() {
    if (side) return 0;
    if (call_captures() || any_capture()) { set_error("the library's side stream cannot be created while a stream capture is open: run the call once before the capture"); return FINROM_ERR_UNSUPPORTED; }
    int lo = 0, hi = 0;
    FR_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));          // numerically lower = higher priority
    FR_HIP(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi));
    FR_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    FR_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    return 0;
  }
  // finrom_romml_grad -> finrom_rom_grad (one-sample form only): leave the gradient's partial sums to the consumer; where they are
  bool defer_gsum = false; const double* last_gpart = nullptr;
  const MlpFuse* fuse = nullptr;       // (same path) the error model's forward pass as a workgroup of the contraction kernel
  const MlpBackFuse* back = nullptr;   // (same path) its walk back through the hidden layers as a workgroup of the gradient contraction
  bool info_store = false;             // (same path) the solve kernel stores info instead of or-ing into it
};
struct finrom_sampler_s { double* U = nullptr; int n = 0; Scratch xi; };
struct finrom_mlp_s {
  MlpDev d{}; std::vector<void*> owned; Scratch tape, theta, gth, shift, qtmp, etmp, g0, y0p, thp;
  // finrom_hmc_leapfrog: the partial sums of theta the last step left in thp, and the field they belong to (position buffer,
  // momentum buffer, step size, chains): the next step uses them only if it is that field's step
  const double* carry_k = nullptr; const double* carry_mom = nullptr; double carry_eps = 0.0; int64_t carry_S = 0;
};

template <class T>
static int up(std::vector<void*>& owned, const T** dst, const T* host, size_t count) {
  T* dp = nullptr;
  int rc = upload(&dp, host, count);
  if (rc) return rc;
  if (dp) owned.push_back(dp);
  *dst = dp;
  return 0;
}

extern "C" {

int finrom_version(void) { return FINROM_ABI_VERSION; }
const char* finrom_last_error(void) { return g_err.c_str(); }

int finrom_device_count(int* count) { if (!count) return FINROM_ERR_ARG; FR_HIP(hipGetDeviceCount(count)); return 0; }
int finrom_set_device(int ordinal) { FR_HIP(hipSetDevice(ordinal)); return 0; }
// finrom_malloc / finrom_free recycle small buffers (<= 16 MiB, size classes = powers of two, at most 256 MiB parked): the scalar
// call surface allocates a handful of small buffers per call, and hipMalloc / hipFree cost more than its kernels.  A parked
// buffer carries the event finrom_free_async recorded behind its last user; the next owner waits for it.
namespace {
constexpr size_t kPoolMaxEach = (size_t)16 << 20, kPoolMaxTotal = (size_t)256 << 20;
struct Parked { void* p; hipEvent_t ev; int dev; };       // (dev: a parked buffer is only handed out on the device it lives on)
std::mutex g_pool_mu;
std::vector<std::pair<void*, size_t>> g_pool_live;          // pooled allocations handed out: (pointer, size class)
std::vector<std::pair<size_t, Parked>> g_pool_free;         // parked: (size class, buffer)
size_t g_pool_free_bytes = 0;
std::vector<hipEvent_t> g_pool_events;
size_t size_class(size_t bytes) { size_t c = 256; while (c < bytes) c <<= 1; return c; }
void event_destroy_cb(void* e) { (void)hipEventDestroy((hipEvent_t)e); }
}  // namespace

int finrom_malloc(void** dptr, size_t bytes) {
  if (!dptr) return FINROM_ERR_ARG;
  *dptr = nullptr;
  if (bytes == 0) return 0;
  const bool capture = any_capture();
  if (!capture) flush_deferred();
  size_t cap = bytes;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (bytes <= kPoolMaxEach) {
    cap = size_class(bytes);
    std::unique_lock<std::mutex> lk(g_pool_mu);
    for (size_t i = g_pool_free.size(); i-- > 0;) {          // (the most recently parked first)
      if (g_pool_free[i].first != cap || g_pool_free[i].second.dev != dev) continue;
      Parked b = g_pool_free[i].second;
      bool done = b.ev == nullptr;
      if (!done) { done = hipEventQuery(b.ev) == hipSuccess; if (!done) (void)hipGetLastError(); }
      if (!done && capture) continue;                          // (no host wait while a capture is open)
      g_pool_free.erase(g_pool_free.begin() + i);
      g_pool_free_bytes -= cap;
      g_pool_live.emplace_back(b.p, cap);
      lk.unlock();
      if (b.ev != nullptr) {
        // the last user (a launch on some non-blocking stream) must have finished before the new owner writes
        const hipError_t e = done ? hipSuccess : hipEventSynchronize(b.ev);
        std::lock_guard<std::mutex> lk2(g_pool_mu);
        g_pool_events.push_back(b.ev);
        if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize(pooled buffer)");
      }
      *dptr = b.p;
      return 0;
    }
  }
  if (capture) { set_error("finrom_malloc: refused while a stream capture is open (allocate before the capture begins)"); return FINROM_ERR_UNSUPPORTED; }
  hipError_t e = hipMalloc(dptr, cap);
  if (e != hipSuccess) { set_error("hipMalloc of " + std::to_string(cap) + " bytes failed: " + hipGetErrorName(e)); return FINROM_ERR_NOMEM; }
  if (bytes <= kPoolMaxEach) { std::lock_guard<std::mutex> lk(g_pool_mu); g_pool_live.emplace_back(*dptr, cap); }
  return 0;
}
// stream: the stream of the buffer's last user, or NULL when that user is known to have finished / ran on the default stream
static int pool_free(void* dptr, hipStream_t stream, bool have_stream) {
  if (!dptr) return 0;
  size_t cap = 0;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool_live.size(); ++i)
      if (g_pool_live[i].first == dptr) { cap = g_pool_live[i].second; g_pool_live.erase(g_pool_live.begin() + i); break; }
  }
  const bool st_captures = have_stream && note_stream(stream);
  if (cap != 0 && !st_captures) {
    hipEvent_t ev = nullptr;
    if (have_stream && stream != nullptr) {
      {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool_events.empty()) { ev = g_pool_events.back(); g_pool_events.pop_back(); }
      }
      if (ev == nullptr && !any_capture() && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
      if (ev != nullptr && hipEventRecord(ev, stream) != hipSuccess) { (void)hipGetLastError(); defer_or_run(event_destroy_cb, ev); ev = nullptr; cap = 0; }
      if (ev == nullptr) cap = 0;                              // no event to be had: do not park, free (deferred under capture)
    }
    if (cap != 0) {
      int dev = 0;
      hipPointerAttribute_t at{};
      if (hipPointerGetAttributes(&at, dptr) == hipSuccess) dev = at.device; else { (void)hipGetLastError(); (void)hipGetDevice(&dev); }
      std::lock_guard<std::mutex> lk(g_pool_mu);
      if (g_pool_free_bytes + cap <= kPoolMaxTotal) {
        g_pool_free.push_back({cap, Parked{dptr, ev, dev}});
        g_pool_free_bytes += cap;
        return 0;
      }
      if (ev != nullptr) g_pool_events.push_back(ev);
    }
  }
  // not pooled, pool full, or last used inside a capture (the graph may replay it: never recycled, freed once no capture is open)
  dev_free(dptr);
  return 0;
}
int finrom_free(void* dptr) { return pool_free(dptr, nullptr, false); }
int finrom_free_async(void* dptr, void* stream) { return pool_free(dptr, (hipStream_t)stream, true); }
int finrom_note_stream(void* stream) {
  const bool cap = note_stream((hipStream_t)stream);
  if (!cap) flush_deferred();
  return cap ? 1 : 0;
}
int finrom_deferred_count(void) { return deferred_count(); }
int finrom_flush_deferred(void) { flush_deferred(); return deferred_count(); }
int finrom_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  if (bytes == 0) return 0;
  CallGuard cg((hipStream_t)stream);
  if (call_captures() || any_capture()) { set_error("finrom_memcpy_h2d synchronises: not while a stream capture is open"); return FINROM_ERR_UNSUPPORTED; }
  FR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  FR_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
int finrom_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  if (bytes == 0) return 0;
  CallGuard cg((hipStream_t)stream);
  if (call_captures() || any_capture()) { set_error("finrom_memcpy_d2h synchronises: not while a stream capture is open"); return FINROM_ERR_UNSUPPORTED; }
  FR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  FR_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
int finrom_memset(void* dst, int value, size_t bytes, void* stream) {
  if (bytes == 0) return 0;
  CallGuard cg((hipStream_t)stream);
  if (stream == nullptr && any_capture()) { set_error("finrom_memset on the default stream: not while a stream capture is open"); return FINROM_ERR_UNSUPPORTED; }
  FR_HIP(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
  return 0;
}
int finrom_stream_sync(void* stream) {
  CallGuard cg((hipStream_t)stream);
  if (call_captures()) { set_error("finrom_stream_sync on a capturing stream"); return FINROM_ERR_UNSUPPORTED; }
  FR_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int finrom_set_overlap(int on) { g_overlap = on != 0; return 0; }
int finrom_profile_enable(int on) { std::lock_guard<std::mutex> lk(g_prof_mu); g_prof_on = on != 0; return 0; }
int finrom_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  drain_pending();
  for (int i = 0; i < K_NUM; ++i) { g_ms[i] = 0; g_cnt[i] = 0; }
  return 0;
}
int finrom_profile_slots(void) { return K_NUM; }
int finrom_profile_read(int slot, const char** name, int64_t* launches, double* total_ms) {
  if (slot < 0 || slot >= K_NUM) return FINROM_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  drain_pending();
  if (name) *name = kSlotNames[slot];
  if (launches) *launches = g_cnt[slot];
  if (total_ms) *total_ms = g_ms[slot];
  return 0;
}

// ---------------------------------------------------------------------------------------
// FOM
// ---------------------------------------------------------------------------------------
int finrom_fom_create(const finrom_fom_desc* a, finrom_fom_t* out) {
  if (!a || !out) { set_error("fom_create: null argument"); return FINROM_ERR_ARG; }
  *out = nullptr;
  const int n = a->n, nnzL = a->nnzL;
  if (n <= 0 || nnzL < n || a->xdim <= 0 || a->n_obs < 0 || a->nasm < 0 || a->n_alist < 0 || a->cache_slots < 1 ||
      a->cache_slots > 126 || (a->fwd_chunk != 8 && a->fwd_chunk != 16) || a->nops_fwd < 2 * a->fwd_chunk ||
      a->nops_bwd < 2 * VM_CHUNK || a->nops_fwd % (2 * a->fwd_chunk) || a->nops_bwd % (2 * VM_CHUNK)) {
    set_error("fom_create: inconsistent sizes"); return FINROM_ERR_ARG;
  }
  // validate every index the kernels will dereference (a bad index would fault the GPU) and
  // the prefetch rule of the op streams
  auto bad = [&](const char* what) { set_error(std::string("fom_create: invalid ") + what); return FINROM_ERR_ARG; };
  const int gsize = nnzL + 2 * n;
  if ((int64_t)(nnzL + 3 * (int64_t)n) * 512 >= (int64_t)1 << 31) { set_error("fom_create: value vector too long for 32-bit buffer offsets"); return FINROM_ERR_UNSUPPORTED; }
  for (int t = 0; t < a->n_alist; ++t) if (a->a_list[t] < 0 || a->a_list[t] >= nnzL) return bad("a_list");
  if (a->asm_ptr[0] != 0 || a->asm_ptr[nnzL] != a->nasm) return bad("asm_ptr");
  for (int e = 0; e < nnzL; ++e) if (a->asm_ptr[e + 1] < a->asm_ptr[e]) return bad("asm_ptr");
  for (int t = 0; t < a->nasm; ++t) if (a->asm_idx[t] < 0 || a->asm_idx[t] >= a->xdim) return bad("asm_idx");
  int fused = 0;
  if (a->n_imm < 0 || (a->n_imm > 0 && !a->imm)) return bad("imm");
  {
    std::vector<int> stored(gsize, -10);
    std::vector<char> slot_set(std::max(a->cache_slots, 1), 0);      // row-cache slots that have been written (FINOFF)
    auto need = [&](int g, int c) { return g >= 0 && g < gsize && stored[g] + 2 <= c; };
    for (int t = 0; t < a->nops_fwd; ++t) {
      const int k = a->fwd_kind[t], A = a->fwd_a[t], B = a->fwd_b[t], D = a->fwd_d[t], c = t / a->fwd_chunk;
      if (A < -1 || A >= gsize) return bad("forward op operand index");     // fetched for every op, used or not
      switch (k) {
        case 0: if ((A >= 0 && !need(A, c)) || A < -1 || B < 0 || B >= a->cache_slots + 2) return bad("forward FMA op"); break;
        case 3: case 4: if (!need(A, c)) return bad("forward LDX/FMAX op"); break;
        case 5: if (!need(A, c) || D < 0 || D >= nnzL || B < -1 || B >= a->cache_slots) return bad("forward FINOFF op"); stored[D] = c;
                if (B >= 0) slot_set[B] = 1;
                break;
        case 11: if (B < 0 || B >= a->cache_slots || D < 0 || D >= a->cache_slots || !slot_set[B] || !slot_set[D]) return bad("forward FMALL op"); break;
        case 6: if (D < 0 || D >= nnzL || B < 0 || B >= n) return bad("forward FINDIAG op"); stored[D] = c; stored[nnzL + B] = c; break;
        case 7: if (D < 0 || D >= n) return bad("forward YSET op"); break;
        case 9: if (a->xdim > FOM_MAX_FUSED_X || B < 0 || B >= a->xdim || D < 0 || D >= a->n_imm) return bad("forward XFMA op"); fused = 1; break;
        case 10: if (a->xdim > FOM_MAX_FUSED_X || D < 0 || D >= a->n_imm) return bad("forward CADD op"); fused = 1; break;
        case 8: if (D < nnzL + n || D >= gsize) return bad("forward FINY op"); stored[D] = c; break;
        default: return bad("forward op kind");
      }
      if (t >= a->nops_fwd - 2 * a->fwd_chunk && (k != 0 || B != a->cache_slots + 1)) return bad("forward stream tail (must be padding)");
    }
    std::fill(stored.begin(), stored.end(), -10);
    auto need1 = [&](int g, int c) { return g >= 0 && g < gsize && stored[g] + 2 <= c; };   // backward: same look-ahead as forward
    for (int t = 0; t < a->nops_bwd; ++t) {
      const int k = a->bwd_kind[t], A = a->bwd_a[t], B = a->bwd_b[t], D = a->bwd_d[t], c = t / VM_CHUNK;
      if (A < -1 || A >= gsize || B < -1 || B >= gsize) return bad("backward op operand index");
      switch (k) {
        case 0: break;
        case 1: if (!need1(A, c) || !need1(B, c)) return bad("backward WFMA op"); break;
        case 3: if (!need1(A, c)) return bad("backward WSET op"); break;
        case 5: if (!need1(A, c) || D < nnzL + n || D >= gsize) return bad("backward WFIN op"); stored[D] = c; break;
        default: return bad("backward op kind");
      }
      if (t >= a->nops_bwd - 2 * VM_CHUNK && k != 0) return bad("backward stream tail (must be NOP padding)");
    }
  }
  if (a->n_obs > 0) {
    if (a->obs_ptr[0] != 0) return bad("obs_ptr");
    for (int o = 0; o < a->n_obs; ++o) if (a->obs_ptr[o + 1] < a->obs_ptr[o]) return bad("obs_ptr");
    for (int t = 0; t < a->obs_ptr[a->n_obs]; ++t) if (a->obs_idx[t] < 0 || a->obs_idx[t] >= n) return bad("obs_idx");
  }
  {
    std::vector<char> seen(n, 0);
    for (int i = 0; i < n; ++i) { int v = a->perm[i]; if (v < 0 || v >= n || seen[v]) return bad("perm"); seen[v] = 1; }
  }

  auto* h = new finrom_fom_s();
  FomDev& d = h->d;
  d.n = n; d.nnzL = nnzL; d.xdim = a->xdim; d.n_obs = a->n_obs; d.n_alist = a->n_alist; d.cache_slots = a->cache_slots;
  d.gsize = gsize;
  d.fwd_chunk = a->fwd_chunk;
  d.nchunks_fwd = a->nops_fwd / a->fwd_chunk - 2; d.nchunks_bwd = a->nops_bwd / VM_CHUNK - 2;
  d.debug_phases = 7; d.trace = nullptr;
  d.has_grad = 0; d.nchunks_res = 0;
  if (const char* ph = getenv("FINROM_FOM_PHASES")) d.debug_phases = atoi(ph);
  std::vector<int> fkb(a->nops_fwd), bkb(a->nops_bwd);
  // device encoding of the forward stream: load offsets in bytes, the common multiply-adds (FMA, FMALL) carry the LDS
  // byte offsets of their row-cache slots in kb and d, every other op kind | (b+1) << 8, and one bit mask per chunk flags
  // the slots that are NOT plain multiply-adds (bits 0..15) and the multiply-adds whose second operand is in LDS (bits 16..31)
  std::vector<int> fa2(a->nops_fwd), fd2(a->nops_fwd), fmask(a->nops_fwd / a->fwd_chunk, 0);
  const bool noload = getenv("FINROM_FOM_NOLOAD") != nullptr;      // timing experiment: no operand fetches (results are garbage)
  for (int t = 0; t < a->nops_fwd; ++t) {
    // ops without a global operand (padding, fused-assembly ops) fetch from beyond the buffer's range: the hardware
    // bounds check returns 0 without touching memory (a real element could be uninitialised: 0 * NaN)
    fa2[t] = (a->fwd_a[t] < 0 || noload) ? 0x7FFFFFF0 : a->fwd_a[t] * 512;
    fd2[t] = a->fwd_d[t];
    // the common multiply-add is acc -= rc[b] * (G[a] + rc[d']): FMA reads the ZERO slot as d', FMALL an out-of-range G[a]
    // the common multiply-add is acc -= rc[b] * (G[a] + rc[d']): FMA reads the ZERO slot as d', FMALL an out-of-range G[a]
    if (a->fwd_kind[t] == 0) { fkb[t] = a->fwd_b[t] * 512; fd2[t] = (a->cache_slots + 1) * 512; }
    else if (a->fwd_kind[t] == 11) { fkb[t] = a->fwd_b[t] * 512; fd2[t] = a->fwd_d[t] * 512;
                                       fmask[t / a->fwd_chunk] |= 1 << (16 + t % a->fwd_chunk); }     // second operand in LDS
    else { fkb[t] = a->fwd_kind[t] | ((a->fwd_b[t] + 1) << 8); fmask[t / a->fwd_chunk] |= 1 << (t % a->fwd_chunk); }
  }
  // backward stream on the device: byte offsets of both operands (out of range = no operand: the fetch returns 0 without
  // touching memory) and kind | d << 8
  std::vector<int> ba2(a->nops_bwd), bb2(a->nops_bwd);
  for (int t = 0; t < a->nops_bwd; ++t) {
    const int k = a->bwd_kind[t];
    ba2[t] = (k == 0 || a->bwd_a[t] < 0) ? 0x7FFFFFF0 : a->bwd_a[t] * 512;
    bb2[t] = (k != 1 || a->bwd_b[t] < 0) ? 0x7FFFFFF0 : a->bwd_b[t] * 512;
    bkb[t] = k | ((k == 5 ? a->bwd_d[t] : 0) << 8);
  }
  int rc = 0;
  const int nobsnz = a->n_obs > 0 ? a->obs_ptr[a->n_obs] : 0;
  // assembly records: {entry, idx0..3, first/last of the remaining terms, 0} and {c0, w0..3}; unused slots
  // multiply parameter 0 by a zero weight
  std::vector<int> reci((size_t)a->n_alist * 8, 0);
  std::vector<double> recd((size_t)a->n_alist * 5, 0.0);
  for (int t = 0; t < a->n_alist; ++t) {
    const int e = a->a_list[t], t0 = a->asm_ptr[e], t1 = a->asm_ptr[e + 1];
    reci[t * 8 + 0] = e; recd[t * 5 + 0] = a->asm_c0[e];
    for (int k = 0; k < 4 && t0 + k < t1; ++k) { reci[t * 8 + 1 + k] = a->asm_idx[t0 + k]; recd[t * 5 + 1 + k] = a->asm_w[t0 + k]; }
    reci[t * 8 + 5] = std::min(t0 + 4, t1); reci[t * 8 + 6] = t1;
  }
  if (!rc) rc = up(h->owned, &d.asm_rec_i, reci.data(), reci.size());
  if (!rc) rc = up(h->owned, &d.asm_rec_d, recd.data(), recd.size());
  if (!rc) rc = up(h->owned, &d.asm_idx, a->asm_idx, a->nasm);
  if (!rc) rc = up(h->owned, &d.asm_w, a->asm_w, a->nasm);
  if (!rc) rc = up(h->owned, &d.rhs, a->rhs, n);
  if (!rc) rc = up(h->owned, &d.f_a, fa2.data(), fa2.size());
  if (!rc) rc = up(h->owned, &d.f_mask, fmask.data(), fmask.size());
  d.fused = fused;
  if (fused && a->n_alist != 0) { finrom_fom_destroy(h); set_error("fom_create: a fused stream must come with n_alist = 0"); return FINROM_ERR_ARG; }
  if (!rc) rc = up(h->owned, &d.f_imm, a->imm, (size_t)a->n_imm);
  if (!rc) rc = up(h->owned, &d.f_kb, fkb.data(), fkb.size());
  if (!rc) rc = up(h->owned, &d.f_d, fd2.data(), fd2.size());
  if (!rc) rc = up(h->owned, &d.b_a, ba2.data(), ba2.size());
  if (!rc) rc = up(h->owned, &d.b_b, bb2.data(), bb2.size());
  if (!rc) rc = up(h->owned, &d.b_kd, bkb.data(), bkb.size());
  if (!rc) rc = up(h->owned, &d.obs_ptr, a->obs_ptr, a->n_obs + 1);
  if (!rc) rc = up(h->owned, &d.obs_idx, a->obs_idx, nobsnz);
  if (!rc) rc = up(h->owned, &d.obs_w, a->obs_w, nobsnz);
  if (!rc) rc = up(h->owned, &d.perm, a->perm, n);
  if (rc) { finrom_fom_destroy(h); return rc; }
  *out = h;
  return 0;
}

void finrom_fom_destroy(finrom_fom_t h) {
  if (!h) return;
  for (void* p : h->owned) dev_free(p);                 // (queued while a stream capture is open: finrom_internal.h)
  h->xT.release(); h->Gw.release(); h->gradT.release(); h->qtmp.release();
  delete h;
}

// stages: 1 = pack + assembly, 2 = interpreter (+ unpack of w), 3 = both.  finrom_solve_pairs runs stage 1, then
// launches the ROM half, then stage 2 (only possible when the batch fits one workspace chunk).
static int64_t fom_chunk_samples(const FomDev& d, const BandDev* band = nullptr) {
  // bound the per-call workspace (L values dominate: nnzL * 8 B per sample)
  const size_t per_sample = ((size_t)(band && band->on ? band->gsize : d.gsize) + d.xdim) * sizeof(double);
  const int64_t chunk = (int64_t)((size_t)48 << 30) / (int64_t)per_sample;
  return std::max<int64_t>(64, chunk / 64 * 64);
}
static int fom_solve_stages(finrom_fom_t h, const double* x, int64_t S, double* qoi, double* w, int32_t* info, hipStream_t st,
                            int stages) {
  const FomDev& d = h->d;
  static const bool env_no_band = getenv("FINROM_NO_BAND") != nullptr;
  // Small batches: where the mesh has a band plan the band sweep serves them too -- ONE wave walking its 1597 pivots takes 1.7 ms
  // (QoI only) / 2.3 ms (with w) at m = 12 and 4.2 / 4.4 ms at m = 20, the level-scheduled small-batch kernel 2.7 and 17.9 ms
  // (tools/fom_small_vs_band.py) -- so the latency-oriented schedule (one workgroup per sample) is the forward path only of
  // meshes without window sizes (m >= 32); finrom_fom_gradient keeps it for small batches (its adjoint stage beats a lone wave
  // of the band adjoint kernel: 3.1 against 10 ms).
  if (h->small.small_max > 0 && S <= h->small.small_max && !(h->band.on && !env_no_band)) {
    int rc;
    if (!h->small.in_lds && (rc = h->Gw.reserve((size_t)S * d.gsize * sizeof(double)))) return rc;
    if (!(stages & 2)) return 0;
    h->last_path = h->small.in_lds ? FINROM_FOM_PATH_SMALL_LDS : FINROM_FOM_PATH_SMALL_GLOBAL;
    return launch_fom_small(d, h->small, x, S, (double*)h->Gw.p, qoi, w, info, st);
  }
  if (h->band.on && !env_no_band) {
    const BandDev& b = h->band;
    h->last_path = band_path(b, w == nullptr);
    const int64_t limit = fom_chunk_samples(d, &b);
    const int64_t npieces = (S + limit - 1) / limit;
    const int64_t chunk = npieces <= 1 ? limit : ((S + npieces - 1) / npieces + 63) / 64 * 64;
    for (int64_t s0 = 0; s0 < S; s0 += chunk) {
      const int64_t Sc = std::min(chunk, S - s0), nblk = (Sc + 63) / 64;
      int rc;
      if (stages & (1 | 4)) {                            // (4: reserve the workspace only -- finrom_solve_pairs, before its fork)
        if ((rc = h->xT.reserve((size_t)nblk * d.xdim * 64 * sizeof(double)))) return rc;
        if ((rc = h->Gw.reserve((size_t)nblk * b.gsize * 64 * sizeof(double)))) return rc;
      }
      if (stages & 1) {
        if ((rc = launch_pack(x + s0 * d.xdim, Sc, d.xdim, (double*)h->xT.p, st))) return rc;
        if ((rc = launch_fom_assemble(h->band_asm, (const double*)h->xT.p, nblk, (double*)h->Gw.p, st))) return rc;
      }
      if (stages & 2) {
        if ((rc = launch_fom_band(b, (double*)h->Gw.p, nblk, Sc, qoi ? qoi + s0 * d.n_obs : nullptr, info ? info + s0 : nullptr, st, w == nullptr))) return rc;
        if (w && (rc = launch_unpack((const double*)h->Gw.p, Sc, d.n, b.gsize, b.offY, b.perm, w + s0 * d.n, st))) return rc;
      }
    }
    return 0;
  }
  h->last_path = FINROM_FOM_PATH_INTERPRETER;
  // equal pieces when the batch exceeds the workspace bound (a short last piece would leave the GPU mostly idle)
  const int64_t limit = fom_chunk_samples(d), npieces = (S + limit - 1) / limit;
  const int64_t chunk = npieces <= 1 ? limit : ((S + npieces - 1) / npieces + 63) / 64 * 64;
  for (int64_t s0 = 0; s0 < S; s0 += chunk) {
    const int64_t Sc = std::min(chunk, S - s0);
    const int64_t nblk = (Sc + 63) / 64;
    int rc;
    if (stages & (1 | 4)) {
      if ((rc = h->xT.reserve((size_t)nblk * d.xdim * 64 * sizeof(double)))) return rc;
      if ((rc = h->Gw.reserve((size_t)nblk * d.gsize * 64 * sizeof(double)))) return rc;
    }
    if (stages & 1) {
      if ((rc = launch_pack(x + s0 * d.xdim, Sc, d.xdim, (double*)h->xT.p, st))) return rc;
      if (!d.fused && (rc = launch_fom_assemble(d, (const double*)h->xT.p, nblk, (double*)h->Gw.p, st))) return rc;
    }
    if (stages & 2) {
      if ((rc = launch_fom(d, (const double*)h->xT.p, nblk, Sc, (double*)h->Gw.p, qoi ? qoi + s0 * d.n_obs : nullptr, info ? info + s0 : nullptr, st))) return rc;
      if (w && (rc = launch_unpack_w(d, (const double*)h->Gw.p, Sc, w + s0 * d.n, st))) return rc;
    }
  }
  return 0;
}

int finrom_fom_solve(finrom_fom_t h, const double* x, int64_t S, double* qoi, double* w, int32_t* info, void* stream) {
  CallGuard cg((hipStream_t)stream);
  if (!h || S < 0 || (S > 0 && (!x || (!qoi && h->d.n_obs > 0)))) { set_error("fom_solve: bad argument"); return FINROM_ERR_ARG; }
  return fom_solve_stages(h, x, S, qoi, w, info, (hipStream_t)stream, 3);
}

int finrom_fom_last_path(finrom_fom_t h) {
  if (!h) { set_error("fom_last_path: null handle"); return FINROM_ERR_ARG; }
  return h->last_path;
}

int finrom_fom_set_small_max(finrom_fom_t h, int32_t small_max) {
  if (!h || small_max < 0) { set_error("fom_set_small_max: bad argument"); return FINROM_ERR_ARG; }
  if (small_max > 0 && h->small.rec == nullptr) { set_error("fom_set_small_max: finrom_fom_set_small has not been called"); return FINROM_ERR_ARG; }
  h->small.small_max = small_max;
  return 0;
}

int finrom_fom_set_small(finrom_fom_t h, const finrom_fom_small_desc* a) {
  if (!h || !a || a->small_max < 0 || a->npairs < 0 || a->nasm < 0 || a->nlev_f <= 0 || a->nlev_b <= 0) { set_error("fom_set_small: bad argument"); return FINROM_ERR_ARG; }
  const FomDev& d = h->d;
  const int n = d.n, nnzL = d.nnzL;
  auto bad = [&](const char* what) { set_error(std::string("fom_set_small: invalid ") + what); return FINROM_ERR_ARG; };
  // every index the kernel dereferences, and the level property (a row only reads rows of lower levels)
  if (a->row_ptr[0] != 0 || a->row_ptr[n] != nnzL) return bad("row_ptr");
  std::vector<int> ent_row(nnzL);
  for (int i = 0; i < n; ++i) {
    if (a->row_ptr[i + 1] <= a->row_ptr[i]) return bad("row_ptr");
    for (int e = a->row_ptr[i]; e < a->row_ptr[i + 1]; ++e) {
      ent_row[e] = i;
      const int j = a->ent_col[e];
      if (j < 0 || j > i || (e == a->row_ptr[i + 1] - 1) != (j == i)) return bad("ent_col (diagonal must close its row)");
    }
  }
  if (a->pair_ptr[0] != 0 || a->pair_ptr[nnzL] != a->npairs) return bad("pair_ptr");
  if (a->asm_ptr[0] != 0 || a->asm_ptr[nnzL] != a->nasm) return bad("asm_ptr");
  for (int e = 0; e < nnzL; ++e) {
    if (a->pair_ptr[e + 1] < a->pair_ptr[e] || a->asm_ptr[e + 1] < a->asm_ptr[e]) return bad("pair_ptr / asm_ptr");
    for (int k = a->pair_ptr[e]; k < a->pair_ptr[e + 1]; ++k) {
      const int pa = a->pair_a[k], pb = a->pair_b[k];
      if (pa < a->row_ptr[ent_row[e]] || pa >= e || pb < 0 || pb >= nnzL || ent_row[pb] != a->ent_col[e]) return bad("pair");
    }
  }
  for (int t = 0; t < a->nasm; ++t) if (a->asm_idx[t] < 0 || a->asm_idx[t] >= d.xdim) return bad("asm_idx");
  if (a->col_ptr[0] != 0 || a->col_ptr[n] != nnzL - n) return bad("col_ptr");
  for (int j = 0; j < n; ++j) {
    if (a->col_ptr[j + 1] < a->col_ptr[j]) return bad("col_ptr");
    for (int c = a->col_ptr[j]; c < a->col_ptr[j + 1]; ++c) {
      const int e = a->col_ent[c];
      if (e < 0 || e >= nnzL || a->ent_col[e] != j || ent_row[e] != a->col_row[c] || a->col_row[c] <= j) return bad("column view");
    }
  }
  auto levels_ok = [&](const int* ptr, const int* rows, int nlev, bool forward) {
    if (ptr[0] != 0 || ptr[nlev] != n) return false;
    std::vector<int> lev(n, -1);
    for (int l = 0; l < nlev; ++l) {
      if (ptr[l + 1] < ptr[l]) return false;
      for (int t = ptr[l]; t < ptr[l + 1]; ++t) { const int i = rows[t]; if (i < 0 || i >= n || lev[i] >= 0) return false; lev[i] = l; }
    }
    for (int i = 0; i < n; ++i) {
      if (lev[i] < 0) return false;
      if (forward) { for (int e = a->row_ptr[i]; e < a->row_ptr[i + 1] - 1; ++e) if (lev[a->ent_col[e]] >= lev[i]) return false; }
      else { for (int c = a->col_ptr[i]; c < a->col_ptr[i + 1]; ++c) if (lev[a->col_row[c]] >= lev[i]) return false; }
    }
    return true;
  };
  if (!levels_ok(a->lev_ptr_f, a->lev_rows_f, a->nlev_f, true)) return bad("forward levels");
  if (!levels_ok(a->lev_ptr_b, a->lev_rows_b, a->nlev_b, false)) return bad("backward levels");
  FomSmallDev q;
  int rc = 0;
  // device layout for latency: a stream of fixed-size records (see FomSmallRec), (entry, row) column items as 8-B items
  std::vector<FomSmallRec> rec;
  std::vector<int> rec_ptr(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    for (int e = a->row_ptr[i]; e < a->row_ptr[i + 1]; ++e) {
      const int k0 = a->pair_ptr[e], np = a->pair_ptr[e + 1] - k0, t0 = a->asm_ptr[e], nt = a->asm_ptr[e + 1] - t0;
      const int nrec = std::max(1, std::max((np + 15) / 16, (nt + 7) / 8));
      for (int c = 0; c < nrec; ++c) {
        FomSmallRec r{};
        r.e = e; r.col = a->ent_col[e];
        r.npair = std::max(0, std::min(16, np - 16 * c));
        for (int u = 0; u < r.npair; ++u) r.first[u] = int2{a->pair_a[k0 + 16 * c + u], a->pair_b[k0 + 16 * c + u]};
        const int ntc = std::max(0, std::min(8, nt - 8 * c));
        for (int u = 0; u < ntc; ++u) { r.aidx[u] = a->asm_idx[t0 + 8 * c + u]; r.aw[u] = a->asm_w[t0 + 8 * c + u]; }
        r.c0 = c == 0 ? a->asm_c0[e] : 0.0;
        r.flags = (c == nrec - 1 ? 1 : 0) | (r.col == i ? 2 : 0) | ((ntc > 0 || r.c0 != 0.0) ? 4 : 0);
        rec.push_back(r);
      }
    }
    rec_ptr[i + 1] = (int)rec.size();
  }
  rec.resize(rec.size() + 6, FomSmallRec{});
  std::vector<int2> colv((size_t)(nnzL - n) + 8, int2{0, 0});
  for (int c = 0; c < nnzL - n; ++c) colv[c] = int2{a->col_ent[c], a->col_row[c]};
  if (!rc) rc = up(h->owned, &q.row_ptr, a->row_ptr, n + 1);
  if (!rc) rc = up(h->owned, &q.ent_col, a->ent_col, nnzL);
  if (!rc) rc = up(h->owned, &q.rec_ptr, rec_ptr.data(), rec_ptr.size());
  if (!rc) rc = up(h->owned, &q.rec, rec.data(), rec.size());
  if (!rc) rc = up(h->owned, &q.col_ptr, a->col_ptr, n + 1);
  if (!rc) rc = up(h->owned, &q.colv, colv.data(), colv.size());
  if (!rc) rc = up(h->owned, &q.lev_ptr_f, a->lev_ptr_f, a->nlev_f + 1);
  if (!rc) rc = up(h->owned, &q.lev_rows_f, a->lev_rows_f, n);
  if (!rc) rc = up(h->owned, &q.lev_ptr_b, a->lev_ptr_b, a->nlev_b + 1);
  if (!rc) rc = up(h->owned, &q.lev_rows_b, a->lev_rows_b, n);
  if (rc) return rc;
  q.small_max = a->small_max; q.nlev_f = a->nlev_f; q.nlev_b = a->nlev_b;
  q.in_lds = (size_t)(nnzL + 3 * n + d.xdim) * sizeof(double) <= (size_t)156 * 1024;      // value vector (+ adjoint region) + x
  h->small = q;
  return 0;
}

// Host-only check of a band descriptor against the sizes of the FOM it belongs to (every index the kernels dereference, the
// privacy of the slots the fins write to, the window / pipeline preconditions).  No device call: also exported as
// finrom_fom_band_validate so that a descriptor can be checked on a box without a GPU (tools/asan_validators.py).
static int validate_band(const finrom_fom_band_desc* a, int n, int xdim, int n_obs, int64_t* gsize_out, int64_t* nL_out) {
  if (!a || n <= 0 || xdim <= 0 || n_obs < 0) { set_error("fom_set_band: null argument"); return FINROM_ERR_ARG; }
  struct { int n, xdim, n_obs; } d{n, xdim, n_obs};
  auto bad = [&](const char* what) { set_error(std::string("fom_set_band: invalid ") + what); return FINROM_ERR_ARG; };
  if (!band_supported(a->NSF, a->NSP, a->NX)) { set_error("fom_set_band: window sizes not built into the library"); return FINROM_ERR_UNSUPPORTED; }
  if (a->nfins < 0 || a->npf < 0 || a->nif != a->NSF - 1 || a->npost <= 0 || a->nAB <= 0 || a->nterms < 0 || a->nLx < 0) return bad("sizes");
  // the sweeps' prologue fills a whole window of NS nodes and their software pipelines request the entering node NS positions
  // ahead of the pivot: a segment with fewer pivots than window slots is not a shape the kernels were written for
  if (a->nfins > 0 && a->npf < a->NSF) return bad("npf (a fin needs at least NSF pivots)");
  if (a->npost < a->NSP) return bad("npost (the post needs at least NSP pivots)");
  const int G = a->nfins * (a->npf + a->nif) + a->npost;
  if ((int64_t)a->nfins * a->npf + a->npost != n) return bad("pivot count (must equal the number of dofs)");
  // every table is dereferenced below and by the kernels: a missing pointer is a descriptor error, not a host fault
  if (!a->abmap || !a->ab_c0 || !a->ab_ptr || !a->Fg || !a->act || !a->lx_ptr || !a->ent_extra || !a->ecp_ptr || !a->perm ||
      (a->nterms > 0 && (!a->ab_idx || !a->ab_w)) || (a->nfins > 0 && (!a->schur_off || !a->iface_elim)) ||
      (d.n_obs > 0 && !a->obs_ptr)) return bad("table pointer (null)");
  for (int e = 0; e < 3 * G; ++e) if (a->abmap[e] < 0 || a->abmap[e] >= a->nAB) return bad("abmap");
  const int64_t nL = (int64_t)a->nfins * a->npf * a->NSF + (int64_t)a->npost * a->NSP;
  const int64_t gsize = (int64_t)a->nAB + nL + a->nLx + n + band_xsize(a->NSP) + n;      // ... | y -> w | extras (LDS variant) | adjoint
  if (gsize * 512 >= (int64_t)1 << 31) { set_error("fom_set_band: workspace too long for 32-bit buffer offsets"); return FINROM_ERR_UNSUPPORTED; }
  if (a->ab_ptr[0] != 0 || a->ab_ptr[a->nAB] != a->nterms) return bad("ab_ptr");
  for (int e = 0; e < a->nAB; ++e) if (a->ab_ptr[e + 1] < a->ab_ptr[e]) return bad("ab_ptr");
  for (int t = 0; t < a->nterms; ++t) if (a->ab_idx[t] < 0 || a->ab_idx[t] >= d.xdim) return bad("ab_idx");
  if (a->lx_ptr[0] != 0 || a->lx_ptr[a->npost] != a->nLx || a->ecp_ptr[0] != 0) return bad("lx_ptr / ecp_ptr");
  for (int p = 0; p < a->npost; ++p) {
    if (a->act[p] < 0 || a->act[p] >= (1 << a->NX)) return bad("act");
    if (a->lx_ptr[p + 1] - a->lx_ptr[p] != __builtin_popcount((unsigned)a->act[p])) return bad("lx_ptr (must count the bits of act)");
    if (a->ent_extra[p] < 0 || a->ent_extra[p] > a->NX) return bad("ent_extra");
    if (a->ecp_ptr[p + 1] < a->ecp_ptr[p]) return bad("ecp_ptr");
  }
  const int necp_ = a->ecp_ptr[a->npost];
  if (necp_ > 0 && (!a->ecp_slot || !a->ecp_off)) return bad("table pointer (null)");
  for (int c = 0; c < necp_; ++c)
    if (a->ecp_slot[c] < 0 || a->ecp_slot[c] >= a->NX || a->ecp_off[c] < 0 || a->ecp_off[c] >= a->nAB) return bad("ecp_slot / ecp_off");
  const int nift = a->nif * (a->nif + 1) / 2;
  for (int t = 0; t < a->nfins * nift; ++t) if (a->schur_off[t] < 0 || a->schur_off[t] >= a->nAB) return bad("schur_off");
  // (interface nodes are POST nodes: the fins' backward sweeps read their w after the post's has written it)
  for (int t = 0; t < a->nfins * a->nif; ++t) if (a->iface_elim[t] < a->nfins * a->npf || a->iface_elim[t] >= n) return bad("iface_elim");
  {
    // The fins of a workgroup are swept by DIFFERENT waves (fom_band_ldsw_kernel), each adding its Schur complement to the post's
    // value slots with a plain load-add-store: a slot written by two fins would race.  Slots are private per fin, distinct
    // within a fin, and not shared with plain (read-only, de-duplicated) value slots of segment nodes other than the post's
    // own interface entries -- i.e. a Schur / extra-coupling slot appears in abmap at most once.
    std::vector<int> owner(a->nAB, -1);                 // fin that writes the slot
    for (int f = 0; f < a->nfins; ++f)
      for (int t = 0; t < nift; ++t) {
        const int off = a->schur_off[f * nift + t];
        if (owner[off] >= 0) return bad(owner[off] == f ? "schur_off (slot repeated within a fin)" : "schur_off (slot shared by two fins)");
        owner[off] = f;
      }
    std::vector<int> uses(a->nAB, 0);
    for (int e = 0; e < 3 * G; ++e) if (a->abmap[e] >= 0 && a->abmap[e] < a->nAB) ++uses[a->abmap[e]];
    for (int c = 0; c < necp_; ++c) ++uses[a->ecp_off[c]];
    for (int e = 0; e < a->nAB; ++e) if (owner[e] >= 0 && uses[e] > 1) return bad("schur_off (a slot the fins write to is read by more than one entry)");
  }
  {
    std::vector<char> seen(n, 0);
    for (int i = 0; i < n; ++i) { int v = a->perm[i]; if (v < 0 || v >= n || seen[v]) return bad("perm"); seen[v] = 1; }
  }
  if (d.n_obs > 0) {
    if (a->obs_ptr[0] != 0) return bad("obs_ptr");
    for (int o = 0; o < d.n_obs; ++o) if (a->obs_ptr[o + 1] < a->obs_ptr[o]) return bad("obs_ptr");
    if (a->obs_ptr[d.n_obs] > 0 && (!a->obs_idx || !a->obs_w)) return bad("table pointer (null)");
    for (int t = 0; t < a->obs_ptr[d.n_obs]; ++t) if (a->obs_idx[t] < 0 || a->obs_idx[t] >= n) return bad("obs_idx");
  }
  // the fins' sweeps do not carry a load on their own nodes over to the interface nodes' right-hand sides
  for (int f = 0; f < a->nfins; ++f)
    for (int t = 0; t < a->npf + a->nif; ++t)
      if (a->Fg[f * (a->npf + a->nif) + t] != 0.0) { set_error("fom_set_band: load on a fin's segment nodes is not supported by the sweep"); return FINROM_ERR_UNSUPPORTED; }
  // QoI-only tables: the same operator as obs_*, entry by entry
  const int nq = (a->qoi_FgQ != nullptr) + (a->qoi_row_fin != nullptr) + (a->qoi_obs_ptr != nullptr);
  if (nq != 0) {
    if (nq != 3 || d.n_obs <= 0) return bad("QoI-only tables (all or none)");
    const int post_e0 = a->nfins * a->npf, post_g0 = a->nfins * (a->npf + a->nif), ntot = a->npf + a->nif;
    if (a->qoi_obs_ptr[0] != 0) return bad("qoi_obs_ptr");
    for (int o = 0; o < d.n_obs; ++o) if (a->qoi_obs_ptr[o + 1] < a->qoi_obs_ptr[o]) return bad("qoi_obs_ptr");
    const int nqz = a->qoi_obs_ptr[d.n_obs];
    if (nqz > 0 && (!a->qoi_obs_idx || !a->qoi_obs_w)) return bad("table pointer (null)");
    for (int t = 0; t < nqz; ++t) if (a->qoi_obs_idx[t] < post_e0 || a->qoi_obs_idx[t] >= n) return bad("qoi_obs_idx (post nodes only)");
    std::vector<int> fin_row(std::max(a->nfins, 1), -1);
    for (int o = 0; o < d.n_obs; ++o) {
      const int f = a->qoi_row_fin[o];
      if (f < -1 || f >= a->nfins) return bad("qoi_row_fin");
      if (f >= 0) { if (fin_row[f] >= 0) return bad("qoi_row_fin (a fin belongs to at most one row)"); fin_row[f] = o; }
    }
    for (int g = 0; g < a->npost; ++g) if (a->qoi_FgQ[post_g0 + g] != a->Fg[post_g0 + g]) return bad("qoi_FgQ (must equal Fg on the post)");
    for (int f = 0; f < a->nfins; ++f)
      if (fin_row[f] < 0)
        for (int t = 0; t < ntot; ++t) if (a->qoi_FgQ[f * ntot + t] != 0.0) return bad("qoi_FgQ (weights on a fin no row owns)");
    std::vector<double> full(n), rec(n);
    for (int o = 0; o < d.n_obs; ++o) {
      std::fill(full.begin(), full.end(), 0.0); std::fill(rec.begin(), rec.end(), 0.0);
      for (int t = a->obs_ptr[o]; t < a->obs_ptr[o + 1]; ++t) full[a->obs_idx[t]] += a->obs_w[t];
      for (int t = a->qoi_obs_ptr[o]; t < a->qoi_obs_ptr[o + 1]; ++t) rec[a->qoi_obs_idx[t]] += a->qoi_obs_w[t];
      const int f = a->qoi_row_fin[o];
      if (f >= 0) {
        for (int t = 0; t < a->npf; ++t) rec[f * a->npf + t] += a->qoi_FgQ[f * ntot + t];
        for (int t = 0; t < a->nif; ++t) rec[a->iface_elim[f * a->nif + t]] += a->qoi_FgQ[f * ntot + a->npf + t];
      }
      for (int i = 0; i < n; ++i) if (full[i] != rec[i]) return bad("QoI-only tables (not the operator obs_* describes)");
    }
  }
  if (gsize_out) *gsize_out = gsize;
  if (nL_out) *nL_out = nL;
  return 0;
}

int finrom_fom_band_validate(const finrom_fom_band_desc* a, int32_t n, int32_t xdim, int32_t n_obs) {
  return validate_band(a, n, xdim, n_obs, nullptr, nullptr);
}

int finrom_fom_set_band(finrom_fom_t h, const finrom_fom_band_desc* a) {
  if (!h || !a) { set_error("fom_set_band: null argument"); return FINROM_ERR_ARG; }
  const FomDev& d = h->d;
  int64_t gsize = 0, nL = 0;
  if (int rc = validate_band(a, d.n, d.xdim, d.n_obs, &gsize, &nL)) return rc;
  const int n = d.n, G = a->nfins * (a->npf + a->nif) + a->npost, nift = a->nif * (a->nif + 1) / 2;
  BandDev b{};
  b.n = n; b.n_obs = d.n_obs; b.xdim = d.xdim; b.gsize = (int)gsize; b.nAB = a->nAB; b.nL = (int)nL; b.nLx = a->nLx;
  b.NSF = a->NSF; b.NSP = a->NSP; b.NX = a->NX; b.nfins = a->nfins; b.npf = a->npf; b.nif = a->nif; b.npost = a->npost;
  b.post_g0 = a->nfins * (a->npf + a->nif); b.post_e0 = a->nfins * a->npf; b.post_L0 = a->nfins * a->npf * a->NSF;
  b.offL = a->nAB; b.offLx = a->nAB + (int)nL; b.offY = a->nAB + (int)nL + a->nLx; b.offX = b.offY + n;
  b.offV = b.offX + band_xsize(a->NSP);
  // records of the assembly pre-pass (fom_assemble_kernel): every value slot, so that special slots start at zero
  std::vector<int> reci((size_t)a->nAB * 8, 0);
  std::vector<double> recd((size_t)a->nAB * 5, 0.0);
  for (int e = 0; e < a->nAB; ++e) {
    const int t0 = a->ab_ptr[e], t1 = a->ab_ptr[e + 1];
    reci[e * 8 + 0] = e; recd[e * 5 + 0] = a->ab_c0[e];
    for (int k = 0; k < 4 && t0 + k < t1; ++k) { reci[e * 8 + 1 + k] = a->ab_idx[t0 + k]; recd[e * 5 + 1 + k] = a->ab_w[t0 + k]; }
    reci[e * 8 + 5] = std::min(t0 + 4, t1); reci[e * 8 + 6] = t1;
  }
  FomDev& q = h->band_asm;
  q = FomDev{};
  q.xdim = d.xdim; q.gsize = (int)gsize; q.n_alist = a->nAB;
  int rc = 0;
  const int necp = a->ecp_ptr[a->npost], nobsnz = d.n_obs > 0 ? a->obs_ptr[d.n_obs] : 0;
  if (!rc) rc = up(h->owned, &q.asm_rec_i, reci.data(), reci.size());
  if (!rc) rc = up(h->owned, &q.asm_rec_d, recd.data(), recd.size());
  if (!rc) rc = up(h->owned, &q.asm_idx, a->ab_idx, a->nterms);
  if (!rc) rc = up(h->owned, &q.asm_w, a->ab_w, a->nterms);
  if (!rc) rc = up(h->owned, &b.abmap, a->abmap, (size_t)3 * G);
  if (!rc) rc = up(h->owned, &b.Fg, a->Fg, G);
  if (!rc) rc = up(h->owned, &b.act, a->act, a->npost);
  if (!rc) rc = up(h->owned, &b.lx_ptr, a->lx_ptr, a->npost + 1);
  if (!rc) rc = up(h->owned, &b.ent_extra, a->ent_extra, a->npost);
  if (!rc) rc = up(h->owned, &b.ecp_ptr, a->ecp_ptr, a->npost + 1);
  if (!rc) rc = up(h->owned, &b.ecp_slot, a->ecp_slot, necp);
  if (!rc) rc = up(h->owned, &b.ecp_off, a->ecp_off, necp);
  if (!rc) rc = up(h->owned, &b.schur_off, a->schur_off, (size_t)a->nfins * nift);
  if (!rc) rc = up(h->owned, &b.iface_elim, a->iface_elim, (size_t)a->nfins * a->nif);
  if (!rc) rc = up(h->owned, &b.perm, a->perm, n);
  if (!rc) rc = up(h->owned, &b.obs_ptr, a->obs_ptr, d.n_obs + 1);
  if (!rc) rc = up(h->owned, &b.obs_idx, a->obs_idx, nobsnz);
  if (!rc) rc = up(h->owned, &b.obs_w, a->obs_w, nobsnz);
  if (a->qoi_FgQ != nullptr && getenv("FINROM_BAND_NO_QOI_ONLY") == nullptr) {
    const int nqz = a->qoi_obs_ptr[d.n_obs];
    if (!rc) rc = up(h->owned, &b.FgQ, a->qoi_FgQ, G);
    if (!rc) rc = up(h->owned, &b.row_fin, a->qoi_row_fin, d.n_obs);
    if (!rc) rc = up(h->owned, &b.qobs_ptr, a->qoi_obs_ptr, d.n_obs + 1);
    if (!rc) rc = up(h->owned, &b.qobs_idx, a->qoi_obs_idx, nqz);
    if (!rc) rc = up(h->owned, &b.qobs_w, a->qoi_obs_w, nqz);
    b.qo = 1;
  }
  if (rc) return rc;
  b.on = getenv("FINROM_BAND_TIMING") != nullptr ? 2 : 1;      // 2: block 0 reports its phase clocks in sample 0's QoI (diagnostic)
  if (getenv("FINROM_BAND_NOMEM") != nullptr) b.on |= 4;       // timing experiment (m <= 12 kernel): no L / y / w traffic, garbage results
  if (getenv("FINROM_BAND_PRIO") != nullptr) b.on |= 8 * (atoi(getenv("FINROM_BAND_PRIO")) & 3);      // experiment: s_setprio 1..3 for the sweep's waves (bits 3-4)
  h->band = b;
  return 0;
}

int finrom_fom_set_band_gradient(finrom_fom_t h, const finrom_fom_band_grad_desc* a) {
  if (!h || !a || !a->bt_ptr || !a->g_ptr) { set_error("fom_set_band_gradient: null argument"); return FINROM_ERR_ARG; }
  if (!h->band.on) { set_error("fom_set_band_gradient: finrom_fom_set_band has not been called"); return FINROM_ERR_ARG; }
  const FomDev& d = h->d;
  const int n = d.n;
  auto bad = [&](const char* what) { set_error(std::string("fom_set_band_gradient: invalid ") + what); return FINROM_ERR_ARG; };
  if (a->bt_ptr[0] != 0) return bad("bt_ptr");
  for (int i = 0; i < n; ++i) if (a->bt_ptr[i + 1] < a->bt_ptr[i]) return bad("bt_ptr");
  const int nbt = a->bt_ptr[n];
  if (nbt > 0 && (!a->bt_obs || !a->bt_w)) return bad("table pointer (null)");
  for (int t = 0; t < nbt; ++t) if (a->bt_obs[t] < 0 || a->bt_obs[t] >= d.n_obs) return bad("bt_obs");
  if (a->g_ptr[0] != 0) return bad("g_ptr");
  for (int j = 0; j < d.xdim; ++j) if (a->g_ptr[j + 1] < a->g_ptr[j]) return bad("g_ptr");
  const int ng = a->g_ptr[d.xdim];
  if (ng > 0 && (!a->g_a || !a->g_b || !a->g_w)) return bad("table pointer (null)");
  for (int t = 0; t < ng; ++t) if (a->g_a[t] < 0 || a->g_a[t] >= n || a->g_b[t] < 0 || a->g_b[t] >= n) return bad("g_a / g_b");
  BandGradDev g;
  int rc = 0;
  if (!rc) rc = up(h->owned, &g.bt_ptr, a->bt_ptr, n + 1);
  if (!rc) rc = up(h->owned, &g.bt_obs, a->bt_obs, nbt);
  if (!rc) rc = up(h->owned, &g.bt_w, a->bt_w, nbt);
  if (!rc) rc = up(h->owned, &g.g_ptr, a->g_ptr, d.xdim + 1);
  if (!rc) rc = up(h->owned, &g.g_a, a->g_a, ng);
  if (!rc) rc = up(h->owned, &g.g_b, a->g_b, ng);
  if (!rc) rc = up(h->owned, &g.g_w, a->g_w, ng);
  if (rc) return rc;
  g.on = 1;
  h->band_grad = g;
  return 0;
}

int finrom_fom_solve_rhs(finrom_fom_t h, const double* x, int64_t S, const double* rhs, int32_t nrhs, double* out, int32_t* info,
                         void* stream) {
  CallGuard cg((hipStream_t)stream);
  if (!h || S < 0 || nrhs < 0 || (S > 0 && nrhs > 0 && (!x || !rhs || !out))) { set_error("fom_solve_rhs: bad argument"); return FINROM_ERR_ARG; }
  if (!h->band.on) { set_error("fom_solve_rhs: needs the band sweep (finrom_fom_set_band)"); return FINROM_ERR_UNSUPPORTED; }
  if (S == 0 || nrhs == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const FomDev& d = h->d;
  const BandDev& b = h->band;
  h->last_path = band_path(b, false);
  const int64_t dr = (int64_t)nrhs * d.n;
  const size_t per_sample = ((size_t)b.gsize + d.xdim + 2 * (size_t)dr) * sizeof(double);
  const int64_t limit = std::max<int64_t>(64, (int64_t)((size_t)48 << 30) / (int64_t)per_sample / 64 * 64);
  for (int64_t s0 = 0; s0 < S; s0 += limit) {
    const int64_t Sc = std::min(limit, S - s0), nblk = (Sc + 63) / 64;
    int rc;
    if ((rc = h->xT.reserve((size_t)nblk * d.xdim * 64 * sizeof(double)))) return rc;
    if ((rc = h->Gw.reserve((size_t)nblk * b.gsize * 64 * sizeof(double)))) return rc;
    if ((rc = h->gradT.reserve((size_t)nblk * 2 * dr * 64 * sizeof(double)))) return rc;      // rhsT | outT
    if ((rc = h->qtmp.reserve((size_t)Sc * std::max(d.n_obs, 1) * sizeof(double)))) return rc;
    double* rhsT = (double*)h->gradT.p;
    double* outT = rhsT + (size_t)nblk * dr * 64;
    if ((rc = launch_pack(x + s0 * d.xdim, Sc, d.xdim, (double*)h->xT.p, st))) return rc;
    if ((rc = launch_fom_assemble(h->band_asm, (const double*)h->xT.p, nblk, (double*)h->Gw.p, st))) return rc;
    if ((rc = launch_fom_band(b, (double*)h->Gw.p, nblk, Sc, (double*)h->qtmp.p, info ? info + s0 : nullptr, st, false))) return rc;
    if ((rc = launch_pack(rhs + s0 * dr, Sc, (int)dr, rhsT, st))) return rc;
    if ((rc = launch_fom_band_resolve(b, (double*)h->Gw.p, nblk, nrhs, rhsT, outT, st))) return rc;
    if ((rc = launch_unpack(outT, Sc, (int)dr, dr, 0, nullptr, out + s0 * dr, st))) return rc;
  }
  return 0;
}

int finrom_fom_set_gradient(finrom_fom_t h, const finrom_fom_grad_desc* a) {
  if (!h || !a || a->nops_res < VM_CHUNK || a->nops_res % VM_CHUNK) { set_error("fom_set_gradient: bad argument"); return FINROM_ERR_ARG; }
  FomDev& d = h->d;
  const int n = d.n, nnzL = d.nnzL, gsize = nnzL + 3 * n;
  auto bad = [&](const char* what) { set_error(std::string("fom_set_gradient: invalid ") + what); return FINROM_ERR_ARG; };
  {
    std::vector<int> stored(gsize, -10);
    auto need1 = [&](int g, int c) { return g >= 0 && g < gsize && stored[g] + 1 <= c; };
    for (int t = 0; t < a->nops_res; ++t) {
      const int k = a->res_kind[t], A = a->res_a[t], B = a->res_b[t], D = a->res_d[t], c = t / VM_CHUNK;
      if (A < -1 || A >= gsize || B < -1 || B >= gsize) return bad("resolve op operand index");
      switch (k) {
        case 0: break;
        case 1: if (!need1(A, c) || !need1(B, c)) return bad("WFMA op"); break;
        case 3: if (!need1(A, c)) return bad("WSET op"); break;
        case 5: if (!need1(A, c) || D < nnzL + 2 * n || D >= gsize) return bad("WFIN op"); stored[D] = c; break;
        default: return bad("op kind");
      }
    }
  }
  if (a->bt_ptr[0] != 0) return bad("bt_ptr");
  for (int i = 0; i < n; ++i) if (a->bt_ptr[i + 1] < a->bt_ptr[i]) return bad("bt_ptr");
  for (int t = 0; t < a->bt_ptr[n]; ++t) if (a->bt_obs[t] < 0 || a->bt_obs[t] >= d.n_obs) return bad("bt_obs");
  if (a->g_ptr[0] != 0) return bad("g_ptr");
  for (int j = 0; j < d.xdim; ++j) if (a->g_ptr[j + 1] < a->g_ptr[j]) return bad("g_ptr");
  for (int t = 0; t < a->g_ptr[d.xdim]; ++t) if (a->g_a[t] < 0 || a->g_a[t] >= n || a->g_b[t] < 0 || a->g_b[t] >= n) return bad("g_a/g_b");
  std::vector<int> rkb(a->nops_res);
  for (int t = 0; t < a->nops_res; ++t) rkb[t] = a->res_kind[t] | ((a->res_b[t] + 1) << 8);
  int rc = 0;
  if (!rc) rc = up(h->owned, &d.r_a, a->res_a, a->nops_res);
  if (!rc) rc = up(h->owned, &d.r_kb, rkb.data(), rkb.size());
  if (!rc) rc = up(h->owned, &d.r_d, a->res_d, a->nops_res);
  if (!rc) rc = up(h->owned, &d.bt_ptr, a->bt_ptr, n + 1);
  if (!rc) rc = up(h->owned, &d.bt_obs, a->bt_obs, a->bt_ptr[n]);
  if (!rc) rc = up(h->owned, &d.bt_w, a->bt_w, a->bt_ptr[n]);
  if (!rc) rc = up(h->owned, &d.g_ptr, a->g_ptr, d.xdim + 1);
  if (!rc) rc = up(h->owned, &d.g_a, a->g_a, a->g_ptr[d.xdim]);
  if (!rc) rc = up(h->owned, &d.g_b, a->g_b, a->g_ptr[d.xdim]);
  if (!rc) rc = up(h->owned, &d.g_w, a->g_w, a->g_ptr[d.xdim]);
  if (rc) return rc;
  d.nchunks_res = a->nops_res / VM_CHUNK;
  d.has_grad = 1;
  d.gsize = gsize;                       // value vector grows by the adjoint region
  return 0;
}

int finrom_fom_gradient(finrom_fom_t h, const double* x, const double* data, int32_t data_per_sample, int64_t S,
                        double* grad, double* J, double* qoi, int32_t* info, void* stream) {
  CallGuard cg((hipStream_t)stream);
  if (!h || S < 0 || (S > 0 && (!x || !data || !grad || !J))) { set_error("fom_gradient: bad argument"); return FINROM_ERR_ARG; }
  if (!h->d.has_grad && !h->band_grad.on) { set_error("fom_gradient: finrom_fom_set_gradient has not been called"); return FINROM_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const FomDev& d = h->d;
  static const bool env_no_band_grad = getenv("FINROM_NO_BAND_GRAD") != nullptr || getenv("FINROM_NO_BAND") != nullptr;
  const bool small = h->small.small_max > 0 && S <= h->small.small_max && d.n_obs <= 64 && d.has_grad;
  if (!small && h->band.on && h->band_grad.on && !env_no_band_grad) {
    // the full band sweep (the factor and w stay in the workspace), then the adjoint solve and the contraction on the same layout
    const BandDev& b = h->band;
    h->last_path = band_path(b, false);
    const size_t per_sample = ((size_t)b.gsize + 2 * d.xdim) * sizeof(double);
    const int64_t limit = std::max<int64_t>(64, (int64_t)((size_t)48 << 30) / (int64_t)per_sample / 64 * 64);
    const int64_t npieces = (S + limit - 1) / limit;
    const int64_t chunk = npieces <= 1 ? limit : ((S + npieces - 1) / npieces + 63) / 64 * 64;
    for (int64_t s0 = 0; s0 < S; s0 += chunk) {
      const int64_t Sc = std::min(chunk, S - s0), nblk = (Sc + 63) / 64;
      int rc;
      if ((rc = h->xT.reserve((size_t)nblk * d.xdim * 64 * sizeof(double)))) return rc;
      if ((rc = h->Gw.reserve((size_t)nblk * b.gsize * 64 * sizeof(double)))) return rc;
      if ((rc = h->gradT.reserve((size_t)nblk * d.xdim * 64 * sizeof(double)))) return rc;
      double* q = qoi ? qoi + s0 * d.n_obs : nullptr;
      if (!q) { if ((rc = h->qtmp.reserve((size_t)Sc * d.n_obs * sizeof(double)))) return rc; q = (double*)h->qtmp.p; }
      if ((rc = launch_pack(x + s0 * d.xdim, Sc, d.xdim, (double*)h->xT.p, st))) return rc;
      if ((rc = launch_fom_assemble(h->band_asm, (const double*)h->xT.p, nblk, (double*)h->Gw.p, st))) return rc;
      if ((rc = launch_fom_band(b, (double*)h->Gw.p, nblk, Sc, q, info ? info + s0 : nullptr, st, false))) return rc;
      if ((rc = launch_fom_band_adjoint(b, h->band_grad, (double*)h->Gw.p, nblk, Sc, q, data + (data_per_sample ? s0 * d.n_obs : 0),
                                        data_per_sample ? d.n_obs : 0, (double*)h->gradT.p, J + s0, st))) return rc;
      if ((rc = launch_unpack((const double*)h->gradT.p, Sc, d.xdim, d.xdim, 0, nullptr, grad + s0 * d.xdim, st))) return rc;
    }
    return 0;
  }
  if (!d.has_grad) { set_error("fom_gradient: finrom_fom_set_gradient has not been called (the band tables serve large batches only)"); return FINROM_ERR_ARG; }
  if (h->small.small_max > 0 && S <= h->small.small_max && d.n_obs <= 64) {      // small batch: value and gradient in one workgroup per sample
    int rc;
    if (!h->small.in_lds && (rc = h->Gw.reserve((size_t)S * d.gsize * sizeof(double)))) return rc;
    double* q = qoi;
    if (!q) { if ((rc = h->qtmp.reserve((size_t)S * d.n_obs * sizeof(double)))) return rc; q = (double*)h->qtmp.p; }
    FomSmallGrad g;
    g.data = data; g.data_stride = data_per_sample ? d.n_obs : 0; g.grad = grad; g.J = J;
    h->last_path = h->small.in_lds ? FINROM_FOM_PATH_SMALL_LDS : FINROM_FOM_PATH_SMALL_GLOBAL;
    return launch_fom_small(d, h->small, x, S, (double*)h->Gw.p, q, nullptr, info, st, g);
  }
  h->last_path = FINROM_FOM_PATH_INTERPRETER;
  const size_t per_sample = ((size_t)d.gsize + 2 * d.xdim) * sizeof(double);
  int64_t chunk = (int64_t)((size_t)48 << 30) / (int64_t)per_sample;
  chunk = std::max<int64_t>(64, chunk / 64 * 64);
  for (int64_t s0 = 0; s0 < S; s0 += chunk) {
    const int64_t Sc = std::min(chunk, S - s0);
    const int64_t nblk = (Sc + 63) / 64;
    int rc;
    if ((rc = h->xT.reserve((size_t)nblk * d.xdim * 64 * sizeof(double)))) return rc;
    if ((rc = h->Gw.reserve((size_t)nblk * d.gsize * 64 * sizeof(double)))) return rc;
    if ((rc = h->gradT.reserve((size_t)nblk * d.xdim * 64 * sizeof(double)))) return rc;
    double* q = qoi ? qoi + s0 * d.n_obs : nullptr;
    if (!q) { if ((rc = h->qtmp.reserve((size_t)Sc * d.n_obs * sizeof(double)))) return rc; q = (double*)h->qtmp.p; }
    if ((rc = launch_pack(x + s0 * d.xdim, Sc, d.xdim, (double*)h->xT.p, st))) return rc;
    if (!d.fused && (rc = launch_fom_assemble(d, (const double*)h->xT.p, nblk, (double*)h->Gw.p, st))) return rc;
    if ((rc = launch_fom(d, (const double*)h->xT.p, nblk, Sc, (double*)h->Gw.p, q, info ? info + s0 : nullptr, st))) return rc;
    if ((rc = launch_fom_adjoint(d, nblk, Sc, (double*)h->Gw.p, q, data + (data_per_sample ? s0 * d.n_obs : 0),
                                 data_per_sample ? d.n_obs : 0, (double*)h->gradT.p, J + s0, st))) return rc;
    if ((rc = launch_unpack((const double*)h->gradT.p, Sc, d.xdim, d.xdim, 0, nullptr, grad + s0 * d.xdim, st))) return rc;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
// ROM
// ---------------------------------------------------------------------------------------
namespace {
int validate_rom_desc(const finrom_rom_desc* a) {
  if (a->n <= 0 || a->r <= 0 || a->P < 0 || a->P > 31 || a->n_obs < 0 || a->nterms < 0) { set_error("rom_create: inconsistent sizes"); return FINROM_ERR_ARG; }
  if (!a->row_ptr || (a->nterms > 0 && (!a->term_p || !a->term_val))) { set_error("rom_create: null table"); return FINROM_ERR_ARG; }
  if (a->row_ptr[0] != 0 || a->row_ptr[a->n] != a->nterms) { set_error("rom_create: invalid row_ptr"); return FINROM_ERR_ARG; }
  for (int i = 0; i < a->n; ++i) if (a->row_ptr[i + 1] < a->row_ptr[i]) { set_error("rom_create: invalid row_ptr"); return FINROM_ERR_ARG; }
  for (int t = 0; t < a->nterms; ++t) if (a->term_p[t] < 0 || a->term_p[t] > a->P) { set_error("rom_create: invalid term_p"); return FINROM_ERR_ARG; }
  // a row lists each theta index at most once: the pattern-uniform k-step tables below fill a slot with the SUM of the row's
  // terms that carry the slot's index, so a repeated index would be counted once per repetition
  for (int i = 0; i < a->n; ++i) {
    unsigned seen = 0;
    for (int t = a->row_ptr[i]; t < a->row_ptr[i + 1]; ++t) {
      if (seen & (1u << a->term_p[t])) { set_error("rom_create: a row of psi lists the same theta index twice (merge the terms)"); return FINROM_ERR_ARG; }
      seen |= 1u << a->term_p[t];
    }
  }
  return 0;
}

// rows with a term, sorted by term count (descending)
std::vector<int> rows_by_term_count(const finrom_rom_desc* a) {
  auto cnt = [&](int row) { return a->row_ptr[row + 1] - a->row_ptr[row]; };
  std::vector<int> order;
  for (int i = 0; i < a->n; ++i) if (cnt(i) > 0) order.push_back(i);
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cnt(x) > cnt(y); });
  return order;
}

// pattern-uniform k-steps (RomDev::tvu): four rows that share ONE list of theta indices; sorted by term count, descending
struct KStep { std::vector<int> pat; int rows[4]; };
std::vector<KStep> uniform_ksteps(const finrom_rom_desc* a, const std::vector<int>& order) {
  std::vector<KStep> ksteps;
  auto pattern = [&](int row) { return std::vector<int>(a->term_p + a->row_ptr[row], a->term_p + a->row_ptr[row + 1]); };
  std::vector<int> byp(order);
  std::stable_sort(byp.begin(), byp.end(), [&](int x, int y) { return pattern(x) < pattern(y); });
  std::vector<int> left;
  for (size_t i0 = 0; i0 < byp.size();) {
    size_t i1 = i0;
    while (i1 < byp.size() && pattern(byp[i1]) == pattern(byp[i0])) ++i1;
    size_t i = i0;
    for (; i + 4 <= i1; i += 4) ksteps.push_back({pattern(byp[i]), {byp[i], byp[i + 1], byp[i + 2], byp[i + 3]}});
    for (; i < i1; ++i) left.push_back(byp[i]);
    i0 = i1;
  }
  // leftovers: consecutive rows share a k-step as long as the union of their patterns has at most ROM_MAX_NT entries
  KStep cur{{}, {-1, -1, -1, -1}};
  int fill = 0;
  auto flush = [&]() { if (fill) ksteps.push_back(cur); cur = KStep{{}, {-1, -1, -1, -1}}; fill = 0; };
  for (int row : left) {
    std::vector<int> u = cur.pat;
    for (int pp : pattern(row)) if (std::find(u.begin(), u.end(), pp) == u.end()) u.push_back(pp);
    if (fill == 4 || (int)u.size() > 4) { flush(); u = pattern(row); }
    cur.pat = u; cur.rows[fill++] = row;
  }
  flush();
  std::stable_sort(ksteps.begin(), ksteps.end(), [](const KStep& x, const KStep& y) { return x.pat.size() > y.pat.size(); });
  return ksteps;
}

// ---- the same k-steps GROUPED by their leading parameter (proj_main_grouped, NB <= 5) ----------------------------------
// A k-step whose rows carry {theta_d} or {1, theta_d} belongs to group d and is accumulated DIVIDED by theta_d: its slab is
// T_d (no arithmetic at all) or T_d + (1/theta_d) T_0 (one multiply-add per block); the accumulators are rescaled where the
// group changes -- by (theta_d / theta_e)^2 between groups d and e, by theta_d^2 after the last -- so that the sum is the
// one the ungrouped loop forms, up to rounding.  Everything else (rows that touch two parameters, merged leftovers) comes
// first, at scale 1, with the usual coefficients.  The per-sample scalars (1, theta, 1 / theta, the rescale factors) are
// one row of RomDev::ext, filled by rom_ext_kernel ahead of the projection kernel; the records name them by index.
struct GroupedTables { std::vector<double> tvg; std::vector<int> kmg, def; int nkg = 0, n_ext = 0, ext_final = 0; };
bool build_grouped_tables(const finrom_rom_desc* a, const std::vector<KStep>& ksteps, int rp, GroupedTables& out) {
  const int r = a->r;
  struct GStep { int group; std::vector<int> pat; const int* rows; };
  std::vector<GStep> gs;
  for (const KStep& ks : ksteps) {
    if (ks.rows[0] < 0 && ks.rows[1] < 0 && ks.rows[2] < 0 && ks.rows[3] < 0) continue;      // (padding k-steps of the ungrouped list)
    int group = 0;
    std::vector<int> pat = ks.pat;
    if (pat.size() == 1 && pat[0] != 0) group = pat[0];
    else if (pat.size() == 2 && (pat[0] == 0) != (pat[1] == 0)) { group = pat[0] ? pat[0] : pat[1]; pat = {group, 0}; }
    gs.push_back({group, pat, ks.rows});
  }
  std::stable_sort(gs.begin(), gs.end(), [](const GStep& x, const GStep& y) { return x.group < y.group; });
  std::vector<int> groups;
  for (const GStep& g : gs) if (g.group && (groups.empty() || groups.back() != g.group)) groups.push_back(g.group);
  const int P = a->P, n_ext = 1 + 2 * P + (int)groups.size() + 1;
  if (groups.empty() || n_ext > 64) return false;
  // ext[0] = 1, ext[p] = theta_p, ext[P + p] = 1 / theta_p, then one factor per change of group; as (numerator, denominator, squared)
  std::vector<int> def(3 * (size_t)n_ext, 0);
  for (int pp = 1; pp <= P; ++pp) { def[3 * pp] = pp; def[3 * (P + pp) + 1] = pp; }
  std::vector<int> factor_of(groups.size() + 1);
  for (size_t g = 0; g <= groups.size(); ++g) {
    const int e = 1 + 2 * P + (int)g;
    def[3 * e] = g == 0 ? 0 : groups[g - 1]; def[3 * e + 1] = g == groups.size() ? 0 : groups[g]; def[3 * e + 2] = 1;
    factor_of[g] = e;
  }
  std::vector<double>& tvg = out.tvg; std::vector<int>& kmg = out.kmg;
  tvg.clear(); kmg.clear();
  int slot = 0, prev = 0;
  bool first = true;
  for (const GStep& g : gs) {
    const int nt = (int)g.pat.size();
    int rec[8] = {slot, nt, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < nt; ++t) rec[4 + t] = g.group ? (t == 0 ? 0 : P + g.group) : g.pat[t];
    if (rec[4] == 0) rec[2] |= 1;                       // the first coefficient is 1: its rows are the slab as they are
    if (g.group != prev && !first) {                   // (nothing to rescale in front of the very first k-step)
      const size_t gi = std::find(groups.begin(), groups.end(), g.group) - groups.begin();
      rec[2] |= 2; rec[3] = factor_of[gi];             // (from scale 1: (1 / theta_d)^2)
    }
    prev = g.group; first = false;
    kmg.insert(kmg.end(), rec, rec + 8);
    for (int t = 0; t < nt; ++t)
      for (int q = 0; q < 4; ++q) {
        const size_t base = tvg.size();
        tvg.resize(base + rp, 0.0);
        const int row = g.rows[q];
        if (row < 0) continue;
        for (int tt = a->row_ptr[row]; tt < a->row_ptr[row + 1]; ++tt)
          if (a->term_p[tt] == g.pat[t]) for (int col = 0; col < r; ++col) tvg[base + col] += a->term_val[(size_t)tt * r + col];
      }
    slot += nt;
  }
  // zero k-steps fill the list up to a multiple of three (the loop rotates three slab buffers) and serve the pipeline's reads
  // of the records / rows of the k-steps behind the last one
  int nkg = (int)gs.size();
  const int zslot = slot;
  tvg.resize(tvg.size() + (size_t)4 * 4 * rp, 0.0);
  while (nkg % 3) { int rec[8] = {zslot, 1, 1, 0, 0, 0, 0, 0}; kmg.insert(kmg.end(), rec, rec + 8); ++nkg; }
  for (int k = 0; k < 8; ++k) { int rec[8] = {zslot, 1, 1, 0, 0, 0, 0, 0}; kmg.insert(kmg.end(), rec, rec + 8); }
  out.def = def; out.nkg = nkg; out.n_ext = n_ext; out.ext_final = factor_of[groups.size()];
  return true;
}
}  // namespace

// Host only (no device needed): the grouped tables finrom_rom_create builds for r <= 80, for tests of the host logic.
int finrom_rom_grouped_tables(const finrom_rom_desc* a, int32_t* nkg, int32_t* n_ext, int32_t* ext_final, int64_t* n_slots,
                              int32_t* kmg, double* tvg, int32_t* ext_def) {
  if (!a || !nkg || !n_ext || !ext_final || !n_slots) { set_error("rom_grouped_tables: null argument"); return FINROM_ERR_ARG; }
  if (int rc = validate_rom_desc(a)) return rc;
  const int NB = (a->r + 15) / 16, rp = 16 * NB;
  GroupedTables g;
  if (NB > 5 || !build_grouped_tables(a, uniform_ksteps(a, rows_by_term_count(a)), rp, g)) { *nkg = 0; *n_ext = 0; *ext_final = 0; *n_slots = 0; return 0; }
  *nkg = g.nkg; *n_ext = g.n_ext; *ext_final = g.ext_final; *n_slots = (int64_t)(g.tvg.size() / ((size_t)4 * rp));
  if (kmg) std::memcpy(kmg, g.kmg.data(), g.kmg.size() * sizeof(int));
  if (tvg) std::memcpy(tvg, g.tvg.data(), g.tvg.size() * sizeof(double));
  if (ext_def) std::memcpy(ext_def, g.def.data(), g.def.size() * sizeof(int));
  return 0;
}

int finrom_rom_create(const finrom_rom_desc* a, finrom_rom_t* out) {
  if (!a || !out) { set_error("rom_create: null argument"); return FINROM_ERR_ARG; }
  *out = nullptr;
  if (int rc = validate_rom_desc(a)) return rc;
  const int r = a->r, NB = (r + 15) / 16, rp = 16 * NB;
  if (NB > 13) { set_error("rom_create: basis size > 208 not supported"); return FINROM_ERR_UNSUPPORTED; }

  // rows sorted by term count (descending) so that the 4 rows of a k-step need the same number
  // of slots and the k-steps fall into a few phases of constant term count NT
  auto cnt = [&](int row) { return a->row_ptr[row + 1] - a->row_ptr[row]; };
  const std::vector<int> order = rows_by_term_count(a);
  const int nrows = (int)order.size();
  const int nk = (nrows + 3) / 4;
  auto* h = new finrom_rom_s();
  RomDev& d = h->d;
  d.n = a->n; d.r = r; d.rp = rp; d.NB = NB; d.P = a->P; d.n_obs = a->n_obs;
  d.solve_in_lds = rp <= 176 ? 1 : 0;
  d.clock_probe = getenv("FINROM_CLOCK_PROBE") != nullptr ? std::max(1, atoi(getenv("FINROM_CLOCK_PROBE"))) : 0; d.trace = nullptr;
  d.n_phases = 0;
  std::vector<double> tv; std::vector<int> pidx;
  auto push_slot = [&](std::vector<double>& T, std::vector<int>& Pi, const std::vector<int>& rows4, int t) {
    for (int q = 0; q < 4; ++q) {
      const size_t base = T.size();
      T.resize(base + rp, 0.0);
      int pi = 0;
      if (q < (int)rows4.size() && rows4[q] >= 0) {
        const int row = rows4[q], tt = a->row_ptr[row] + t;
        if (tt < a->row_ptr[row + 1]) { std::memcpy(&T[base], a->term_val + (size_t)tt * r, r * sizeof(double)); pi = a->term_p[tt]; }
      }
      Pi.push_back(pi);
    }
  };
  int nslots = 0;
  for (int ks = 0; ks < nk; ++ks) {
    std::vector<int> rows4;
    int nt = 0;
    for (int q = 0; q < 4; ++q) {
      const int idx = ks * 4 + q;
      rows4.push_back(idx < nrows ? order[idx] : -1);
      if (idx < nrows) nt = std::max(nt, cnt(order[idx]));
    }
    if (nt > 4) { delete h; set_error("rom_create: a row of psi has more than 4 terms"); return FINROM_ERR_UNSUPPORTED; }
    if (d.n_phases == 0 || d.phase_nt[d.n_phases - 1] != nt) {
      if (d.n_phases == ROM_MAX_PHASES) { delete h; set_error("rom_create: too many phases"); return FINROM_ERR_UNSUPPORTED; }
      d.phase_nt[d.n_phases] = nt; d.phase_ks0[d.n_phases] = ks; d.phase_slot0[d.n_phases] = nslots;
      ++d.n_phases;
    }
    d.phase_ks1[d.n_phases - 1] = ks + 1;
    for (int t = 0; t < nt; ++t) push_slot(tv, pidx, rows4, t);
    nslots += nt;
  }
  // ---- pattern-uniform k-steps for the single-wave kernels (see RomDev::tvu) ----------------------------------------------
  std::vector<double> tvu; std::vector<int> kpat, kmeta;
  d.n_uphases = 0;
  {
    std::vector<KStep> ksteps = uniform_ksteps(a, order);
    // every term-count group gets an even number of k-steps (a k-step of zero rows pads it): the multi-wave main loop is
    // unrolled twice and specialised by term count, so that its phases start at even k-steps
    for (size_t k = 0; k < ksteps.size();) {
      size_t k1 = k;
      while (k1 < ksteps.size() && ksteps[k1].pat.size() == ksteps[k].pat.size()) ++k1;
      if ((k1 - k) % 2) { ksteps.insert(ksteps.begin() + k1, KStep{ksteps[k].pat, {-1, -1, -1, -1}}); ++k1; }
      k = k1;
    }
    for (int t = 0; t < 4; ++t) {
      d.uend[t] = 0;
      for (const KStep& ks : ksteps) if ((int)ks.pat.size() > t) ++d.uend[t];
    }
    int uslots = 0;
    for (size_t k = 0; k < ksteps.size(); ++k) {
      const int nt = (int)ksteps[k].pat.size();
      if (nt > 4) { delete h; set_error("rom_create: a row of psi has more than 4 terms"); return FINROM_ERR_UNSUPPORTED; }
      if (d.n_uphases == 0 || d.uphase_nt[d.n_uphases - 1] != nt) {
        if (d.n_uphases == ROM_MAX_PHASES) { delete h; set_error("rom_create: too many phases"); return FINROM_ERR_UNSUPPORTED; }
        d.uphase_nt[d.n_uphases] = nt; d.uphase_ks0[d.n_uphases] = (int)k; d.uphase_slot0[d.n_uphases] = uslots;
        ++d.n_uphases;
      }
      d.uphase_ks1[d.n_uphases - 1] = (int)k + 1;
      for (int t = 0; t < nt; ++t) {
        const int pp = ksteps[k].pat[t];
        kpat.push_back(pp);
        for (int q = 0; q < 4; ++q) {
          const size_t base = tvu.size();
          tvu.resize(base + rp, 0.0);
          const int row = ksteps[k].rows[q];
          if (row < 0) continue;
          for (int tt = a->row_ptr[row]; tt < a->row_ptr[row + 1]; ++tt)
            if (a->term_p[tt] == pp) for (int col = 0; col < r; ++col) tvu[base + col] += a->term_val[(size_t)tt * r + col];
        }
      }
      uslots += nt;
    }
    // per-k-step records for the interleaved main loop: {first slot, term count, -, -, theta index of slots 0..3}; eight padding
    // k-steps point at the zero slots behind the table
    d.nku = (int)ksteps.size();
    {
      int sl = 0;
      for (size_t k = 0; k < ksteps.size(); ++k) {
        const int nt = (int)ksteps[k].pat.size();
        int rec[8] = {sl, nt, 0, 0, 0, 0, 0, 0};
        for (int t = 0; t < nt; ++t) rec[4 + t] = ksteps[k].pat[t];
        kmeta.insert(kmeta.end(), rec, rec + 8);
        sl += nt;
      }
      for (int k = 0; k < 8; ++k) { int rec[8] = {uslots, 1, 0, 0, 0, 0, 0, 0}; kmeta.insert(kmeta.end(), rec, rec + 8); }
    }
    tvu.resize(tvu.size() + (size_t)4 * 4 * rp, 0.0);       // one k-step of padding for the prefetch
    kpat.resize(kpat.size() + 16, 0);                        // the scalar pipeline reads up to two k-steps ahead
    d.tvu_bytes = (int)std::min<size_t>(tvu.size() * sizeof(double), (size_t)0x7FFFFFF0);

    // ---- the same k-steps grouped by their leading parameter (build_grouped_tables above; proj_main_grouped, NB <= 5) ------
    d.n_ext = 0; d.nkg = 0; d.ext_final = 0; d.ext = nullptr;
    if (NB <= 5 && getenv("FINROM_PROJ_UNGROUPED") == nullptr) {
      GroupedTables gt;
      if (build_grouped_tables(a, ksteps, rp, gt)) {
        d.nkg = gt.nkg; d.n_ext = gt.n_ext; d.ext_final = gt.ext_final;
        d.tvg_bytes = (int)std::min<size_t>(gt.tvg.size() * sizeof(double), (size_t)0x7FFFFFF0);
        int grc = up(h->owned, &d.tvg, gt.tvg.data(), gt.tvg.size());
        if (!grc) grc = up(h->owned, &d.kmg, gt.kmg.data(), gt.kmg.size());
        if (!grc) grc = up(h->owned, &d.ext_def, gt.def.data(), gt.def.size());
        if (grc) { finrom_rom_destroy(h); return grc; }
      }
    }
  }
  // chunk images for the LDS-staged kernel: whole k-steps of one phase, <= 32 KiB, padded to 1 KiB
  std::vector<int> ch_nt, ch_nks, ch_off, ch_bytes;
  std::vector<double> tvc;
  d.n_chunks = 0;
  if (NB <= 5) {
    const size_t cap = 24 * 1024;   // = ROM_CHUNK_BYTES (rom_kernels.hip): 2 buffers + theta = 50 KiB per workgroup
    for (int ph = 0; ph < d.n_phases; ++ph) {
      const int nt = d.phase_nt[ph];
      if (nt == 0) continue;
      const size_t per_ks = (size_t)nt * 4 * rp * 8 + (size_t)nt * 4 * 4;
      const int max_ks = std::max<int>(1, (int)(cap / per_ks));
      if (per_ks > cap) { d.n_chunks = 0; ch_nt.clear(); break; }
      for (int ks = d.phase_ks0[ph]; ks < d.phase_ks1[ph]; ks += max_ks) {
        const int nks = std::min(max_ks, d.phase_ks1[ph] - ks);
        const int slot = d.phase_slot0[ph] + (ks - d.phase_ks0[ph]) * nt;
        const size_t nrows = (size_t)nks * nt * 4;
        const size_t off = tvc.size();
        tvc.insert(tvc.end(), tv.begin() + (size_t)slot * 4 * rp, tv.begin() + (size_t)slot * 4 * rp + nrows * rp);
        const size_t nint = nrows;                      // theta indices, two per double
        tvc.resize(tvc.size() + (nint + 1) / 2, 0.0);
        std::memcpy(reinterpret_cast<char*>(tvc.data()) + (off + nrows * rp) * 8, &pidx[(size_t)slot * 4], nint * 4);
        size_t bytes = (tvc.size() - off) * 8;
        bytes = (bytes + 1023) / 1024 * 1024;
        tvc.resize(off + bytes / 8, 0.0);
        ch_nt.push_back(nt); ch_nks.push_back(nks); ch_off.push_back((int)off); ch_bytes.push_back((int)bytes);
      }
    }
    d.n_chunks = (int)ch_nt.size();
  }
  tv.resize(tv.size() + (size_t)4 * 4 * rp, 0.0);          // one k-step of padding for the prefetch
  pidx.resize(pidx.size() + 16, 0);

  // root rows (F != 0) for B_r = psi^T F
  std::vector<int> frows;
  for (int i = 0; i < a->n; ++i) if (a->rhs[i] != 0.0 && cnt(i) > 0) frows.push_back(i);
  d.rhs_nk = ((int)frows.size() + 3) / 4;
  d.rhs_nt = 0;
  for (int row : frows) d.rhs_nt = std::max(d.rhs_nt, cnt(row));
  std::vector<double> rtv, rf; std::vector<int> rpi;
  for (int ks = 0; ks < d.rhs_nk; ++ks) {
    std::vector<int> rows4;
    for (int q = 0; q < 4; ++q) {
      const int idx = ks * 4 + q;
      rows4.push_back(idx < (int)frows.size() ? frows[idx] : -1);
      rf.push_back(idx < (int)frows.size() ? a->rhs[frows[idx]] : 0.0);
    }
    for (int t = 0; t < d.rhs_nt; ++t) push_slot(rtv, rpi, rows4, t);
  }
  if (rtv.empty()) { rtv.assign(4 * rp, 0.0); rpi.assign(4, 0); rf.assign(4, 0.0); }

  // h_p = Psi_p^T F for the offline/online form of B_r
  std::vector<double> hp((size_t)(a->P + 1) * rp, 0.0);
  for (int i = 0; i < a->n; ++i) {
    if (a->rhs[i] == 0.0) continue;
    for (int t = a->row_ptr[i]; t < a->row_ptr[i + 1]; ++t)
      for (int col = 0; col < r; ++col) hp[(size_t)a->term_p[t] * rp + col] += a->rhs[i] * a->term_val[(size_t)t * r + col];
  }

  int rc = 0;
  if (!rc) rc = up(h->owned, &h->gram.h, hp.data(), hp.size());
  if (d.n_chunks > 0) {
    if (!rc) rc = up(h->owned, &d.ch_nt, ch_nt.data(), ch_nt.size());
    if (!rc) rc = up(h->owned, &d.ch_nks, ch_nks.data(), ch_nks.size());
    if (!rc) rc = up(h->owned, &d.ch_off, ch_off.data(), ch_off.size());
    if (!rc) rc = up(h->owned, &d.ch_bytes, ch_bytes.data(), ch_bytes.size());
    if (!rc) rc = up(h->owned, &d.tvc, tvc.data(), tvc.size());
  }
  if (!rc) rc = up(h->owned, &d.tvu, tvu.data(), tvu.size());
  if (!rc) rc = up(h->owned, &d.kpat, kpat.data(), kpat.size());
  if (!rc) rc = up(h->owned, &d.kmeta, kmeta.data(), kmeta.size());
  if (!rc) rc = up(h->owned, &d.tv, tv.data(), tv.size());
  if (!rc) rc = up(h->owned, &d.pidx, pidx.data(), pidx.size());
  if (!rc) rc = up(h->owned, &d.rhs_tv, rtv.data(), rtv.size());
  if (!rc) rc = up(h->owned, &d.rhs_pidx, rpi.data(), rpi.size());
  if (!rc) rc = up(h->owned, &d.rhs_f, rf.data(), rf.size());
  if (!rc) rc = up(h->owned, &d.obs_phi, a->obs_phi, (size_t)a->n_obs * r);
  if (rc) { finrom_rom_destroy(h); return rc; }
  *out = h;
  return 0;
}

void finrom_rom_destroy(finrom_rom_t h) {
  if (!h) return;
  for (void* p : h->owned) dev_free(p);
  h->Ar.release(); h->Br.release(); h->theta.release(); h->qtmp.release(); h->vw.release(); h->ticket.release(); h->grad_ticket.release();
  h->part.release(); h->ext.release();
  if (h->side) defer_or_run([](void* s) { (void)hipStreamDestroy((hipStream_t)s); }, h->side);
  if (h->fom_side) defer_or_run([](void* s) { (void)hipStreamDestroy((hipStream_t)s); }, h->fom_side);
  if (h->ev_join_fom) defer_or_run([](void* e) { (void)hipEventDestroy((hipEvent_t)e); }, h->ev_join_fom);
  if (h->ev_fork) defer_or_run([](void* e) { (void)hipEventDestroy((hipEvent_t)e); }, h->ev_fork);
  if (h->ev_join) defer_or_run([](void* e) { (void)hipEventDestroy((hipEvent_t)e); }, h->ev_join);
  delete h;
}

int finrom_rom_set_gram(finrom_rom_t h, int32_t npairs, const int32_t* pair_p, const int32_t* pair_q, const double* G) {
  if (!h || npairs <= 0 || !pair_p || !pair_q || !G) { set_error("rom_set_gram: bad argument"); return FINROM_ERR_ARG; }
  if (npairs > ROM_GRAM_MAX_PAIRS) { set_error("rom_set_gram: more than 64 pairs"); return FINROM_ERR_UNSUPPORTED; }
  const RomDev& d = h->d;
  const int r = d.r, R = d.rp, NB = d.NB;
  for (int t = 0; t < npairs; ++t)
    if (pair_p[t] < 0 || pair_q[t] > d.P || pair_p[t] > pair_q[t]) { set_error("rom_set_gram: invalid pair (need 0 <= p <= q <= P)"); return FINROM_ERR_ARG; }
  for (int t = 0; t < npairs; ++t)
    for (int u = 0; u < t; ++u)
      if (pair_p[t] == pair_p[u] && pair_q[t] == pair_q[u]) { set_error("rom_set_gram: duplicate pair"); return FINROM_ERR_ARG; }
  auto at = [&](int t, int i, int j) { return (i < r && j < r) ? G[((size_t)t * r + i) * r + j] : 0.0; };
  for (int t = 0; t < npairs; ++t)
    for (int i = 0; i < r; ++i)
      for (int j = 0; j < i; ++j) {
        const double x = at(t, i, j), y = at(t, j, i);
        if (!(std::fabs(x - y) <= 1e-12 * (std::fabs(x) + std::fabs(y)) + 1e-300)) { set_error("rom_set_gram: block is not symmetric"); return FINROM_ERR_ARG; }
      }
  int rc = 0;
  RomGramDev gm = h->gram;
  if (!rc) rc = up(h->owned, &gm.pair_p, pair_p, npairs);
  if (!rc) rc = up(h->owned, &gm.pair_q, pair_q, npairs);
  if (NB <= 6) {       // tile images: lane (q, c), register g of tile (ti <= tj) holds entry (16 ti + q + 4 g, 16 tj + c)
    const int NT = NB * (NB + 1) / 2;
    std::vector<double> Gt((size_t)npairs * NT * 256, 0.0);
    for (int t = 0; t < npairs; ++t) {
      int tile = 0;
      for (int ti = 0; ti < NB; ++ti)
        for (int tj = ti; tj < NB; ++tj, ++tile)
          for (int lane = 0; lane < 64; ++lane)
            for (int g = 0; g < 4; ++g)
              Gt[(((size_t)t * NT + tile) * 64 + lane) * 4 + g] = at(t, 16 * ti + (lane >> 4) + 4 * g, 16 * tj + (lane & 15));
    }
    if (!rc) rc = up(h->owned, &gm.Gt, Gt.data(), Gt.size());
  } else {             // packed upper triangle, row by row (= the lower triangle by columns the Cholesky kernels read)
    const size_t np = (size_t)R * (R + 1) / 2;
    std::vector<double> Gp((size_t)npairs * np, 0.0);
    for (int t = 0; t < npairs; ++t)
      for (int row = 0; row < r; ++row)
        for (int col = row; col < r; ++col) Gp[t * np + (size_t)row * R - ((size_t)row * (row - 1)) / 2 + col - row] = at(t, row, col);
    if (!rc) rc = up(h->owned, &gm.Gp, Gp.data(), Gp.size());
  }
  if (rc) return rc;
  gm.npairs = npairs;
  h->gram = gm;
  return 0;
}

int finrom_rom_set_projection(finrom_rom_t h, int32_t mode) {
  if (!h || (mode != FINROM_PROJECTION_DIRECT && mode != FINROM_PROJECTION_GRAM)) { set_error("rom_set_projection: bad argument"); return FINROM_ERR_ARG; }
  if (mode == FINROM_PROJECTION_GRAM && h->gram.npairs == 0) { set_error("rom_set_projection: finrom_rom_set_gram has not been called"); return FINROM_ERR_ARG; }
  h->projection = mode;
  return 0;
}

// A_r, B_r (or, with factor > 0, the factor / the solution) by the form of the reduced operator the handle is set to
static int rom_project(finrom_rom_t h, const double* theta, int64_t S, int factor, int* info, hipStream_t st,
                       double* w_r = nullptr, double* qoi_r = nullptr) {
  if (h->projection == FINROM_PROJECTION_GRAM)
    return launch_rom_gram(h->d, h->gram, theta, S, (double*)h->Ar.p, (double*)h->Br.p, factor, info, st, w_r, qoi_r);
  if (h->ticket.reserve(4096 * sizeof(int))) return FINROM_ERR_NOMEM;
  RomDev d = h->d;
  d.ext = nullptr;
  if (d.nkg > 0 && S > 0 && !rom_splitk_applies(d, S)) {      // the grouped main loop: room for the samples' scalars
    if (int rc = h->ext.reserve((size_t)S * d.n_ext * sizeof(double))) return rc;
    d.ext = (double*)h->ext.p;
  }
  return launch_rom_proj(d, theta, S, (double*)h->Ar.p, (double*)h->Br.p, factor, info, st, w_r, qoi_r, (int*)h->ticket.p);
}

int finrom_rom_solve(finrom_rom_t h, const double* theta, int64_t S, double* w_r, double* qoi_r, double* A_r,
                     double* B_r, int32_t* info, void* stream) {
  CallGuard cg((hipStream_t)stream);
  if (!h || S < 0 || (S > 0 && (!theta || (!qoi_r && h->d.n_obs > 0)))) { set_error("rom_solve: bad argument"); return FINROM_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const RomDev& d = h->d;
  const size_t per_sample = ((size_t)d.rp * (d.rp + 1) / 2 + d.rp) * sizeof(double);
  int64_t chunk = std::max<int64_t>(4, (int64_t)(((size_t)16 << 30) / per_sample) / 4 * 4);
  if (S > chunk) { const int64_t np = (S + chunk - 1) / chunk; chunk = ((S + np - 1) / np + 3) / 4 * 4; }      // equal pieces
  for (int64_t s0 = 0; s0 < S; s0 += chunk) {
    const int64_t Sc = std::min(chunk, S - s0);
    int rc;
    // one-sample call patterns (MAP / HMC: a handful of samples, latency): contraction over several workgroups per sample, then
    // the MFMA-form factorisation + substitutions (rom_onesample.hip)
    if (A_r == nullptr && B_r == nullptr && h->projection == FINROM_PROJECTION_DIRECT && rom_onesample_applies(d, Sc) &&
        getenv("FINROM_NO_FUSED_SOLVE") == nullptr && getenv("FINROM_NO_FUSED_CHOL") == nullptr) {
      if ((rc = h->part.reserve(rom_onesample_scratch_bytes(d, Sc)))) return rc;
      if ((rc = launch_rom_onesample(d, theta + s0 * d.P, Sc, (double*)h->part.p, 0, RomGradArgs(), w_r ? w_r + s0 * d.r : nullptr,
                                     qoi_r ? qoi_r + s0 * d.n_obs : nullptr, info ? info + s0 : nullptr, st))) return rc;
      continue;
    }
    if ((rc = h->Ar.reserve((size_t)Sc * (d.rp * (d.rp + 1) / 2) * sizeof(double)))) return rc;
    if ((rc = h->Br.reserve((size_t)Sc * d.rp * sizeof(double)))) return rc;
    // A_r is factored inside the projection kernel (in registers, MFMA trailing updates) unless the caller
    // wants A_r itself back (the state the reference's gradients use) or the basis needs more than one wave
    const bool want_factor = (A_r == nullptr || getenv("FINROM_DEBUG_RETURN_FACTOR") != nullptr) &&
                             getenv("FINROM_NO_FUSED_CHOL") == nullptr;   // debug: A_r output then holds L
    int factor = (want_factor && d.NB <= 6) ? 1 : 0;                  // in-register, inside the projection kernel
    // ... and when neither A_r nor B_r is wanted back (r <= 80), the two substitutions and the reduced QoI happen there
    // as well: no packed factor in memory, no second kernel
    if (factor && (d.NB <= 5 || (d.NB == 6 && h->projection == FINROM_PROJECTION_DIRECT && rom_splitk_applies(d, Sc))) && A_r == nullptr && B_r == nullptr && getenv("FINROM_NO_FUSED_SOLVE") == nullptr &&
        getenv("FINROM_PROJ_LDS") == nullptr) factor = 2;
    // wider bases, only the reduced QoI wanted (the sample-pair path): factorisation and QoI stay in the registers of the
    // projection kernel's waves (fused_solve_mw); A_r never reaches memory
    if (d.NB > 6 && want_factor && h->projection == FINROM_PROJECTION_DIRECT && A_r == nullptr && B_r == nullptr && w_r == nullptr &&
        qoi_r != nullptr && d.n_obs + 1 <= 16 * 3 && getenv("FINROM_NO_FUSED_SOLVE") == nullptr) factor = 3;
    if ((rc = rom_project(h, theta + s0 * d.P, Sc, factor, info ? info + s0 : nullptr, st,
                          w_r ? w_r + s0 * d.r : nullptr, qoi_r ? qoi_r + s0 * d.n_obs : nullptr))) return rc;
    if (factor >= 2) continue;
    int factored = factor;
    if (want_factor && d.NB > 6) {                                     // wider bases: blocked MFMA Cholesky kernel
      if ((rc = launch_rom_chol_blocked(d, (double*)h->Ar.p, Sc, info ? info + s0 : nullptr, st))) return rc;
      factored = 1;
    }
    if ((rc = launch_rom_solve(d, (const double*)h->Ar.p, (const double*)h->Br.p, Sc, w_r ? w_r + s0 * d.r : nullptr,
                               qoi_r ? qoi_r + s0 * d.n_obs : nullptr, A_r ? A_r + s0 * (int64_t)d.r * d.r : nullptr,
                               B_r ? B_r + s0 * d.r : nullptr, info ? info + s0 : nullptr, factored, st))) return rc;
  }
  return 0;
}

int finrom_rom_set_gradient(finrom_rom_t h, int32_t npairs, const int32_t* pair_p, const int32_t* pair_i, const double* G) {
  if (!h || npairs < 0 || (npairs > 0 && (!pair_p || !pair_i || !G))) { set_error("rom_set_gradient: bad argument"); return FINROM_ERR_ARG; }
  for (int t = 0; t < npairs; ++t)
    if (pair_p[t] < 0 || pair_p[t] > h->d.P || pair_i[t] < 0 || pair_i[t] >= h->d.P) { set_error("rom_set_gradient: invalid pair"); return FINROM_ERR_ARG; }
  int rc = 0;
  if (!rc) rc = up(h->owned, &h->g_pair_p, pair_p, npairs);
  if (!rc) rc = up(h->owned, &h->g_pair_i, pair_i, npairs);
  if (!rc) rc = up(h->owned, &h->g_Gt, G, (size_t)npairs * h->d.r * h->d.r);
  if (rc) return rc;
  h->g_npairs = npairs;
  return 0;
}

int finrom_rom_grad(finrom_rom_t h, const double* theta, const double* data, int32_t data_per_sample, int64_t S,
                    double* J, double* g, double* w_r, double* qoi_r, int32_t* info, void* stream) {
  CallGuard cg((hipStream_t)stream);
  if (!h || S < 0 || (S > 0 && (!theta || !data || !J || !g))) { set_error("rom_grad: bad argument"); return FINROM_ERR_ARG; }
  if (h->g_npairs == 0) { set_error("rom_grad: finrom_rom_set_gradient has not been called"); return FINROM_ERR_ARG; }
  const RomDev& d = h->d;
  hipStream_t st = (hipStream_t)stream;
  const size_t per_sample = ((size_t)d.rp * (d.rp + 1) / 2 + d.rp) * sizeof(double);
  int64_t chunk = std::max<int64_t>(4, (int64_t)(((size_t)16 << 30) / per_sample) / 4 * 4);
  for (int64_t s0 = 0; s0 < S; s0 += chunk) {
    const int64_t Sc = std::min(chunk, S - s0);
    int rc;
    if ((rc = h->Ar.reserve((size_t)Sc * (d.rp * (d.rp + 1) / 2) * sizeof(double)))) return rc;
    if ((rc = h->Br.reserve((size_t)Sc * d.rp * sizeof(double)))) return rc;
    double* q = qoi_r ? qoi_r + s0 * d.n_obs : nullptr;
    if (!q) { if ((rc = h->qtmp.reserve((size_t)Sc * d.n_obs * sizeof(double)))) return rc; q = (double*)h->qtmp.p; }
    // one-sample call patterns (MAP / HMC), 48 < r <= 96, the per-sample contraction: projection split over four waves, both
    // solves in registers, the gradient contraction dealt over the same waves -- one kernel, nothing but the outputs written
    if (h->projection == FINROM_PROJECTION_DIRECT && rom_onesample_applies(d, Sc) && getenv("FINROM_OLD_SUBST") == nullptr) {
      // contraction over several workgroups per sample, MFMA-form factorisation + forward / adjoint solves (rom_onesample.hip),
      // then the gradient contraction dealt over 36 workgroups per sample
      RomGradArgs ga;
      ga.data = data + (data_per_sample ? s0 * d.n_obs : 0); ga.data_stride = data_per_sample ? d.n_obs : 0;
      ga.theta = theta + s0 * d.P; ga.J = J + s0; ga.g = g + s0 * d.P;
      ga.npairs = h->g_npairs; ga.pair_p = h->g_pair_p; ga.pair_i = h->g_pair_i; ga.Gt = h->g_Gt;
      const size_t vw_bytes = (size_t)Sc * 2 * d.rp * sizeof(double), gp_bytes = (size_t)Sc * ROM_GRAD_SMALL_NG * 32 * sizeof(double);
      if ((rc = h->vw.reserve(vw_bytes + gp_bytes))) return rc;
      if ((rc = h->part.reserve(rom_onesample_scratch_bytes(d, Sc)))) return rc;
      if (!h->grad_ticket.p) {                          // arrival counters: zero once, the kernel leaves them zero
        if ((rc = h->grad_ticket.reserve(ROM_SPLITK_MAX_S * sizeof(int)))) return rc;
        FR_HIP(hipMemset(h->grad_ticket.p, 0, ROM_SPLITK_MAX_S * sizeof(int)));
      }
      ga.vw = (double*)h->vw.p; ga.gpart = (double*)((char*)h->vw.p + vw_bytes); ga.ticket = (int*)h->grad_ticket.p;
      ga.defer_sum = h->defer_gsum && s0 == 0 && Sc == S;
      ga.info_store = ga.defer_sum && h->info_store;
      h->last_gpart = ga.defer_sum ? ga.gpart : nullptr;
      if ((rc = launch_rom_onesample(d, theta + s0 * d.P, Sc, (double*)h->part.p, 1, ga, w_r ? w_r + s0 * d.r : nullptr, q,
                                     info ? info + s0 : nullptr, st, ga.defer_sum ? h->fuse : nullptr))) return rc;
      if ((rc = launch_rom_grad_contract_small(d, Sc, ga, st, ga.defer_sum ? h->back : nullptr))) return rc;
      continue;
    }
    if (h->projection == FINROM_PROJECTION_DIRECT && rom_splitk_applies(d, Sc) && d.n_obs <= 64 && getenv("FINROM_OLD_SUBST") == nullptr) {
      RomGradArgs ga;
      ga.data = data + (data_per_sample ? s0 * d.n_obs : 0); ga.data_stride = data_per_sample ? d.n_obs : 0;
      ga.theta = theta + s0 * d.P; ga.J = J + s0; ga.g = g + s0 * d.P;
      ga.npairs = h->g_npairs; ga.pair_p = h->g_pair_p; ga.pair_i = h->g_pair_i; ga.Gt = h->g_Gt;
      const size_t vw_bytes = (size_t)Sc * 2 * d.rp * sizeof(double), gp_bytes = (size_t)Sc * ROM_GRAD_SMALL_NG * 32 * sizeof(double);
      if ((rc = h->vw.reserve(vw_bytes + gp_bytes))) return rc;
      if (!h->grad_ticket.p) {                          // arrival counters: zero once, the kernel leaves them zero
        if ((rc = h->grad_ticket.reserve(ROM_SPLITK_MAX_S * sizeof(int)))) return rc;
        FR_HIP(hipMemset(h->grad_ticket.p, 0, ROM_SPLITK_MAX_S * sizeof(int)));      // synchronous: ordered before ANY stream's first use
      }
      ga.vw = (double*)h->vw.p; ga.gpart = (double*)((char*)h->vw.p + vw_bytes); ga.ticket = (int*)h->grad_ticket.p;
      {
        ScopedKernelTimer t(K_ROM_PROJ, st);
        if ((rc = launch_rom_proj_splitk(d, theta + s0 * d.P, Sc, (double*)h->Ar.p, (double*)h->Br.p, 4, info ? info + s0 : nullptr, st,
                                         w_r ? w_r + s0 * d.r : nullptr, q, ga))) return rc;
      }
      if ((rc = launch_rom_grad_contract_small(d, Sc, ga, st))) return rc;
      continue;
    }
    // the factor of A_r: inside the projection kernel for r <= 96, by the blocked MFMA Cholesky kernel for wider bases
    if ((rc = rom_project(h, theta + s0 * d.P, Sc, d.NB <= 6 ? 1 : 0, info ? info + s0 : nullptr, st))) return rc;
    if (d.NB > 6 && (rc = launch_rom_chol_blocked(d, (double*)h->Ar.p, Sc, info ? info + s0 : nullptr, st))) return rc;
    RomGradArgs ga;
    ga.data = data + (data_per_sample ? s0 * d.n_obs : 0); ga.data_stride = data_per_sample ? d.n_obs : 0;
    ga.theta = theta + s0 * d.P; ga.J = J + s0; ga.g = g + s0 * d.P;
    ga.npairs = h->g_npairs; ga.pair_p = h->g_pair_p; ga.pair_i = h->g_pair_i; ga.Gt = h->g_Gt;
    // batches: the contraction with the blocks G_pi runs on the matrix cores for 16 samples at a time; a handful of samples
    // (the one-sample call pattern) keep it inside the substitution kernel (one launch less)
    const bool batched = Sc >= 64 && d.n_obs <= 64 && getenv("FINROM_OLD_SUBST") == nullptr && getenv("FINROM_GRAD_INLINE") == nullptr;
    if (batched) {
      if ((rc = h->vw.reserve((size_t)Sc * 2 * d.rp * sizeof(double)))) return rc;
      ga.vw = (double*)h->vw.p;
    }
    if ((rc = launch_rom_grad(d, (const double*)h->Ar.p, (const double*)h->Br.p, Sc, w_r ? w_r + s0 * d.r : nullptr, q,
                              info ? info + s0 : nullptr, ga, st))) return rc;
    if (batched && (rc = launch_rom_grad_contract(d, Sc, ga, st))) return rc;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
int finrom_subfin_avg(const double* Sop, int32_t P, int32_t n, const double* k, int64_t S, double* theta, void* stream) {
  CallGuard cg((hipStream_t)stream);
  if (!Sop || P <= 0 || n <= 0 || S < 0 || (S > 0 && (!k || !theta))) { set_error("subfin_avg: bad argument"); return FINROM_ERR_ARG; }
  return launch_subfin_avg(Sop, P, n, k, S, theta, (hipStream_t)stream);
}

// FINROM_TRACE=<prefix>: every finrom_solve_pairs call records, per workgroup of the FOM interpreter and of the
// projection kernel, {start, end, HW_ID, XCC_ID} and rewrites <prefix>.fom.bin / <prefix>.proj.bin (int64 x 6 per
// workgroup) after a device synchronisation.  Diagnostic for the co-residency of the two halves.
static long long* g_trace_buf[2] = {nullptr, nullptr};
static constexpr size_t kTraceWg = 1u << 18;
static const char* trace_prefix() { static const char* p = getenv("FINROM_TRACE"); return p; }
static int trace_dump(const char* kind, long long* dbuf, size_t nwg) {
  std::vector<long long> h(nwg * 6);
  FR_HIP(hipDeviceSynchronize());
  FR_HIP(hipMemcpy(h.data(), dbuf, nwg * 6 * sizeof(long long), hipMemcpyDeviceToHost));
  const std::string path = std::string(trace_prefix()) + "." + kind + ".bin";
  if (FILE* f = fopen(path.c_str(), "wb")) { fwrite(h.data(), sizeof(long long), h.size(), f); fclose(f); }
  return 0;
}

int finrom_solve_pairs(finrom_fom_t fom, finrom_rom_t rom, const double* Sop, const double* x, int64_t S,
                       double* qoi, double* qoi_r, double* err, double* w, double* w_r, double* theta,
                       int32_t* info, void* stream) {
  CallGuard cg((hipStream_t)stream);
  if (!fom || !rom || !Sop || S < 0 || (S > 0 && (!x || !qoi || !qoi_r))) { set_error("solve_pairs: bad argument"); return FINROM_ERR_ARG; }
  if (fom->d.n_obs != rom->d.n_obs) { set_error("solve_pairs: FOM and ROM observation operators differ in size"); return FINROM_ERR_ARG; }
  if (S == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if ((rc = rom->ensure_side())) return rc;
  if (!theta) {
    if ((rc = rom->theta.reserve((size_t)S * rom->d.P * sizeof(double)))) return rc;
    theta = (double*)rom->theta.p;
  }
  // The two halves run on two streams (FINROM_NO_OVERLAP / finrom_set_overlap(0): in turn, for per-kernel profiling).  With the
  // band sweep as the FOM half (HBM-bound, one 300-register wave per SIMD, vector units 80 % idle) a projection wave fits beside
  // every sweep wave, and since the projection hides its non-MFMA work behind its own MFMAs a lone wave per SIMD is efficient:
  // measured 26.3 ms side by side against 29.1 ms in turn per 100k samples (r = 80).
  static const bool env_no_overlap = getenv("FINROM_NO_OVERLAP") != nullptr;      // (read once, not per call)
  // (The four-wave band sweep of m = 16, 20 takes a whole CU's LDS -- 157 KB per workgroup -- and the wide-basis projection 43 KB
  // per workgroup: the two kernels cannot share a CU, side by side they take turns CU by CU at the dispatcher's discretion.  That
  // still beats running them in turn by 1-5 % (r = 200: 68.6 against 72.0 ms per 20k samples, 420 against 425 ms per 125k), so
  // those sizes fork too; their event-timed kernel durations then include waiting for CUs -- FINROM_NO_OVERLAP=1 gives clean ones.)
  const bool overlap = g_overlap && !env_no_overlap;
  hipStream_t side = overlap ? rom->side : st;
  const bool tracing = trace_prefix() != nullptr;
  if (tracing) {
    for (int i = 0; i < 2; ++i)
      if (!g_trace_buf[i]) FR_HIP(hipMalloc((void**)&g_trace_buf[i], kTraceWg * 6 * sizeof(long long)));
    FR_HIP(hipMemset(g_trace_buf[0], 0, kTraceWg * 6 * sizeof(long long)));
    FR_HIP(hipMemset(g_trace_buf[1], 0, kTraceWg * 6 * sizeof(long long)));
    FR_HIP(hipDeviceSynchronize());
    fom->d.trace = (size_t)((S + 63) / 64) <= kTraceWg ? g_trace_buf[0] : nullptr;
    rom->d.trace = (size_t)S <= kTraceWg ? g_trace_buf[1] : nullptr;
  }
  // (the fork below: everything already queued on the caller's stream -- inputs, zeroed info -- precedes both halves)
  // ROM half first, on a high-priority stream: its 4-wave, 160-VGPR workgroups need large contiguous
  // register ranges, so they claim their slots before the small FOM waves fill the remaining ones
  // The FOM's short bandwidth-bound pre-pass (pack + assembly) used to run first, alone: beside the projection kernel it crawls and
  // the interpreter of round 1, whose waves then started late, WAS the critical path.  Since the band sweep the FOM half ends
  // long before the ROM half (6.8 against 21.8 ms side by side at the headline), so the pre-pass now runs after the fork, on the
  // FOM side, and only the sub-fin averages -- the head of the ROM half -- precede it.
  // Every workspace the two halves need is reserved BEFORE the fork: an allocation failure after it would return while the
  // side stream still writes the caller's outputs.  Any other failure after the fork joins the side stream first.
  const bool split = S <= fom_chunk_samples(fom->d, &fom->band);
  // Round 4: beside the ONE-WAVE projection kernel (r <= 80, two 200-register waves per SIMD) the register band sweep (m <= 12, one
  // 312-register wave per SIMD) runs on a stream masked to THREE CUs of every shader engine (96 of 256): what the projection loses is
  // proportional to the CU-time it shares with sweep waves, and confined to fewer CUs the sweep's waves -- fewer at a time, less
  // contention for HBM -- need less of it in total (256 CUs x 6.9 ms -> 96 x 15.3 ms; the sweep still ends long before the projection).
  // Measured, headline: step 21.54 -> 20.92 ms; two CUs per engine make the sweep the longer half (22.4), four give 21.06.  NOT for the
  // multi-wave projection kernels (r = 120: 49.7 -> 56.7 ms, their workgroups leave the confined sweep too few slots), the four-wave
  // sweep (LDS-exclusive: no change) or the offline/online form (FOM-bound: 9.1 -> 14.3).  FINROM_FOM_CUS=<n> forces n CUs (0: off).
  static const int env_fom_cus = getenv("FINROM_FOM_CUS") != nullptr ? atoi(getenv("FINROM_FOM_CUS")) : -1;
  static const bool env_no_band2 = getenv("FINROM_NO_BAND") != nullptr;
  int fom_cus = env_fom_cus;
  if (fom_cus < 0)
    fom_cus = (rom->projection == FINROM_PROJECTION_DIRECT && rom->d.NB <= 5 && fom->band.on && fom->band.NSP <= 14 && !env_no_band2 &&
               w == nullptr && S >= 16384 && getenv("FINROM_PROJ_LDS") == nullptr) ? -3 : 0;
  hipStream_t fst = st;
  if (overlap && fom_cus != 0) {
    if ((rc = rom->ensure_fom_side(fom_cus))) return rc;
    if (rom->fom_side) fst = rom->fom_side;
  }
  // With the sweep on its masked stream and its pre-pass off it (below), the ROM half can stay on the CALLER's stream: sub-fin
  // averages, per-sample scalars, projection and the difference kernel then follow each other in stream order, and the library's
  // unmasked side stream carries the pre-pass instead -- no cross-stream wait on the step's critical path (each costs 40-80 us:
  // between two steps of the headline the GPU waited 0.33 ms).  FINROM_ROM_ON_SIDE=1: the ROM half on the side stream, as for
  // every other caller's stream.
  static const bool env_rom_on_side = getenv("FINROM_ROM_ON_SIDE") != nullptr;
  static const bool env_pre_masked = getenv("FINROM_FOM_PREPASS_MASKED") != nullptr;
  const bool pre_unmasked = fst != st && split && !env_pre_masked;
  // Only when the caller's stream is a non-blocking one: hipExtStreamCreateWithCUMask has no flags argument, the masked stream is a
  // BLOCKING stream, and the null stream (torch's default) serialises with it -- measured: the sweep then starts when the
  // projection ends (29.0 instead of 20.2 ms per step).
  unsigned st_flags = 0;
  const bool st_nonblocking = st != nullptr && hipStreamGetFlags(st, &st_flags) == hipSuccess && (st_flags & hipStreamNonBlocking);
  const bool rom_on_caller = pre_unmasked && !env_rom_on_side && st_nonblocking;
  const hipStream_t rst = rom_on_caller ? st : side;      // the ROM half
  const hipStream_t pst = rom_on_caller ? side : st;      // the FOM half's unmasked pre-pass
  auto join = [&]() {
    if (overlap && !rom_on_caller && hipEventRecord(rom->ev_join, side) == hipSuccess) (void)hipStreamWaitEvent(st, rom->ev_join, 0);
    if (fst != st && hipEventRecord(rom->ev_join_fom, fst) == hipSuccess) (void)hipStreamWaitEvent(st, rom->ev_join_fom, 0);      // (the sweep waited for the pre-pass)
  };
  auto fail = [&](int code) { if (overlap) (void)hipStreamSynchronize(side); if (fst != st) (void)hipStreamSynchronize(fst); return code; };
  if (split && (rc = fom_solve_stages(fom, x, S, qoi, w, info, st, 4))) return rc;          // (reserves the FOM workspace)
  // the sub-fin averages (bandwidth-bound, short) run BEFORE the fork, alone: beside the sweep they were starved (0.19 -> 0.39 ms
  // at the headline, 0.6 -> 5.8 ms at m = 20) and they head the ROM half's critical path
  if ((rc = launch_subfin_avg(Sop, rom->d.P, fom->d.xdim, x, S, theta, st))) return fail(rc);
  if (overlap) {                          // the ROM half starts after the averages
    FR_HIP(hipEventRecord(rom->ev_fork, st));
    FR_HIP(hipStreamWaitEvent(side, rom->ev_fork, 0));
    if (fst != st) FR_HIP(hipStreamWaitEvent(fst, rom->ev_fork, 0));
  }
  if ((rc = finrom_rom_solve(rom, theta, S, w_r, qoi_r, nullptr, nullptr, info, rst))) return fail(rc);
  // With a masked FOM stream only the SWEEP runs on it: pack + assembly go to an unmasked stream (they take 0.1 ms there against
  // 2.0 ms on 96 CUs beside the projection's start, and since the grouped projection loop the FOM half is the one that ends
  // last: step 20.46 -> 20.26 ms).  FINROM_FOM_PREPASS_MASKED=1: the whole FOM half on the masked stream, as before.
  if (pre_unmasked) {
    if ((rc = fom_solve_stages(fom, x, S, qoi, w, info, pst, 1))) return fail(rc);
    if (hipEventRecord(rom->ev_join_fom, pst) != hipSuccess || hipStreamWaitEvent(fst, rom->ev_join_fom, 0) != hipSuccess) return fail(FINROM_ERR_HIP);
    if ((rc = fom_solve_stages(fom, x, S, qoi, w, info, fst, 2))) return fail(rc);
  } else if ((rc = fom_solve_stages(fom, x, S, qoi, w, info, fst, 3))) return fail(rc);
  join();
  if (err && (rc = launch_sub(qoi, qoi_r, S * (int64_t)fom->d.n_obs, err, st))) return rc;
  if (tracing) {
    if (fom->d.trace) trace_dump("fom", g_trace_buf[0], (size_t)((S + 63) / 64));
    if (rom->d.trace) trace_dump("proj", g_trace_buf[1], (size_t)S);
    fom->d.trace = nullptr; rom->d.trace = nullptr;
  }
  return 0;
}

int finrom_sampler_create(const double* U, int32_t n, finrom_sampler_t* out) {
  if (!U || n <= 0 || !out) { set_error("sampler_create: bad argument"); return FINROM_ERR_ARG; }
  // the kernel only visits the K-range of an UPPER factor (scipy.linalg.cholesky's default, gaussian_field.py:30): a lower
  // factor such as np.linalg.cholesky(C) would silently give wrong fields
  for (int i = 1; i < n; ++i)
    for (int j = 0; j < i; ++j)
      if (U[(size_t)i * n + j] != 0.0) { set_error("sampler_create: U must be upper triangular (pass the transpose of a lower factor)"); return FINROM_ERR_ARG; }
  auto* h = new finrom_sampler_s();
  h->n = n;
  int rc = upload(&h->U, U, (size_t)n * n);
  if (rc) { delete h; return rc; }
  *out = h;
  return 0;
}
void finrom_sampler_destroy(finrom_sampler_t h) { if (!h) return; dev_free(h->U); h->xi.release(); delete h; }
int finrom_sampler_draw(finrom_sampler_t h, const double* xi, int64_t S, double* k, void* stream) {
  CallGuard cg((hipStream_t)stream);
  if (!h || S < 0 || (S > 0 && (!xi || !k))) { set_error("sampler_draw: bad argument"); return FINROM_ERR_ARG; }
  return launch_sampler(h->U, h->n, xi, S, k, (hipStream_t)stream);
}

int finrom_sampler_draw_seeded(finrom_sampler_t h, uint64_t seed, int64_t first_global_sample, int64_t S, double* k, double* xi_out,
                               void* stream) {
  CallGuard cg((hipStream_t)stream);
  if (!h || S < 0 || first_global_sample < 0 || (S > 0 && !k)) { set_error("sampler_draw_seeded: bad argument"); return FINROM_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  // pieces of <= 32k samples bound the xi scratch (1 GiB at n = 4101) when the caller does not want xi back
  const int64_t piece = xi_out ? S : std::min<int64_t>(S, 32768);
  for (int64_t s0 = 0; s0 < S; s0 += piece) {
    const int64_t Sc = std::min(piece, S - s0);
    double* xi = xi_out ? xi_out + s0 * h->n : nullptr;
    if (!xi) { int rc = h->xi.reserve((size_t)Sc * h->n * sizeof(double)); if (rc) return rc; xi = (double*)h->xi.p; }
    int rc = launch_philox_normal(seed, first_global_sample + s0, Sc, h->n, xi, st);
    if (!rc) rc = launch_sampler(h->U, h->n, xi, Sc, k + s0 * h->n, st);
    if (rc) return rc;
  }
  return 0;
}

int finrom_mlp_create(const finrom_mlp_desc* a, finrom_mlp_t* out) {
  if (!a || !out) { set_error("mlp_create: null argument"); return FINROM_ERR_ARG; }
  *out = nullptr;
  if (a->n_in <= 0 || a->n_w <= 0 || a->n_w > 64 || a->n_layers < 0 || a->n_layers > 64 || a->n_out <= 0 || a->n_out > 64 ||
      !a->W0 || !a->b0 || !a->scale || !a->shift || !a->Wh || !a->bh || (a->n_layers > 0 && (!a->W || !a->b))) {
    set_error("mlp_create: inconsistent sizes (n_w, n_out <= 64)"); return FINROM_ERR_ARG;
  }
  auto* h = new finrom_mlp_s();
  MlpDev& d = h->d;
  d.n_in = a->n_in; d.n_w = a->n_w; d.n_layers = a->n_layers; d.n_out = a->n_out;
  int rc = 0;
  const size_t nw = a->n_w, L = a->n_layers;
  if (!rc) rc = up(h->owned, &d.W0, a->W0, (size_t)a->n_in * nw);
  if (!rc) rc = up(h->owned, &d.b0, a->b0, nw);
  if (!rc) rc = up(h->owned, &d.scale, a->scale, (L + 1) * nw);
  if (!rc) rc = up(h->owned, &d.shift, a->shift, (L + 1) * nw);
  if (!rc) rc = up(h->owned, &d.W, a->W, L * nw * nw);
  if (!rc) rc = up(h->owned, &d.b, a->b, L * nw);
  if (!rc) rc = up(h->owned, &d.Wh, a->Wh, nw * a->n_out);
  if (!rc) rc = up(h->owned, &d.bh, a->bh, (size_t)a->n_out);
  if (rc) { finrom_mlp_destroy(h); return rc; }
  *out = h;
  return 0;
}
void finrom_mlp_destroy(finrom_mlp_t h) {
  if (!h) return;
  for (void* p : h->owned) dev_free(p);
  h->tape.release(); h->theta.release(); h->gth.release(); h->shift.release(); h->qtmp.release(); h->etmp.release(); h->g0.release(); h->y0p.release(); h->thp.release();
  delete h;
}
int finrom_mlp_predict(finrom_mlp_t h, const double* k, int64_t S, double* e, void* stream) {
  CallGuard cg((hipStream_t)stream);
  if (!h || S < 0 || (S > 0 && (!k || !e))) { set_error("mlp_predict: bad argument"); return FINROM_ERR_ARG; }
  int rc = h->tape.reserve((size_t)S * (h->d.n_layers + 1) * h->d.n_w * sizeof(float));
  if (rc) return rc;
  return launch_mlp_forward(h->d, k, S, nullptr, 0, (float*)h->tape.p, e, nullptr, (hipStream_t)stream);
}
// (hs: the call is a leapfrog step -- finrom_hmc_leapfrog: position update in front, momentum update behind; one-sample form only)
struct HmcStep { const double* mom; double eps; double* k_out; HmcTail tail; const double* theta_parts_in; };
static int romml_grad_impl(finrom_rom_t rom, finrom_mlp_t mlp, const double* Sop, const double* k, const double* data,
                           int32_t data_per_sample, int64_t S, double* grad, double* loss, double* qoi_r, double* e_nn,
                           int32_t* info, void* stream, const HmcStep* hs) {
  if (!rom || !mlp || !Sop || S < 0 || (S > 0 && (!k || !data || (!grad && !hs) || !loss))) { set_error("romml_grad: bad argument"); return FINROM_ERR_ARG; }
  const MlpDev& m = mlp->d;
  if (m.n_out != rom->d.n_obs) { set_error("romml_grad: the error model's outputs are not the ROM's observables"); return FINROM_ERR_ARG; }
  if (S == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int P = rom->d.P, no = m.n_out;
  int rc;
  if ((rc = mlp->tape.reserve((size_t)S * (m.n_layers + 1) * m.n_w * sizeof(float)))) return rc;
  if ((rc = mlp->theta.reserve(((size_t)S * P + (size_t)S * 16 * 16) * sizeof(double)))) return rc;      // theta | the contraction workgroups' own copies (MlpFuse)
  if ((rc = mlp->gth.reserve((size_t)S * P * sizeof(double)))) return rc;
  if ((rc = mlp->shift.reserve((size_t)S * no * sizeof(double)))) return rc;
  if (!qoi_r) { if ((rc = mlp->qtmp.reserve((size_t)S * no * sizeof(double)))) return rc; qoi_r = (double*)mlp->qtmp.p; }
  if (!e_nn) { if ((rc = mlp->etmp.reserve((size_t)S * no * sizeof(double)))) return rc; e_nn = (double*)mlp->etmp.p; }
  const int64_t stride = data_per_sample ? no : 0;
  // (the sub-fin averages theta = S k are formed inside the network's forward kernel: it reads k anyway)
  // One-sample form (the ROM's contraction + solve kernels, S <= 64): the gradient contraction leaves its partial sums to the
  // error model's backward kernel instead of fencing and drawing tickets for a last workgroup (-7 us per call); the error
  // model's forward pass is one more workgroup per sample of the ROM's contraction kernel (MlpFuse), behind the sub-fin averages.
  // (Tried and removed: under stream capture, the error model's forward kernel on a side stream beside the ROM's contraction
  // kernel -- the solve kernel is the first to need its output.  One fork / join inside a replayed graph cost ~240 us per call
  // on this runtime: 113 -> 355 us, tools/graph_call_cost.py.)
  const bool one = rom->projection == FINROM_PROJECTION_DIRECT && rom_onesample_applies(rom->d, S) && getenv("FINROM_OLD_SUBST") == nullptr &&
                   rom->g_npairs > 0;
  if (hs != nullptr && !(one && P <= 16)) {
    set_error("hmc_leapfrog: needs the one-sample form of the reduced model (<= 64 chains, direct projection, r <= 96, <= 16 averages, finrom_rom_set_gradient)");
    return FINROM_ERR_UNSUPPORTED;
  }
  MlpFuse fm;
  if (hs != nullptr) { fm.mom = hs->mom; fm.eps = hs->eps; fm.k_out = hs->k_out; fm.theta_parts = hs->theta_parts_in; }
  if (one && P <= 16) {                                // the network's forward pass and theta = S k ride in the ROM's contraction kernel
    fm.on = 1; fm.m = m; fm.k = k; fm.data = data; fm.data_stride = stride; fm.tape = (float*)mlp->tape.p; fm.e_out = e_nn;
    fm.data_shift = (double*)mlp->shift.p;
    if ((rc = mlp->y0p.reserve((size_t)S * fm.nw0 * 64 * sizeof(float)))) return rc;
    fm.y0_part = (float*)mlp->y0p.p;
    fm.Sop = Sop; fm.P = P; fm.theta_out = (double*)mlp->theta.p; fm.theta_scr = (double*)mlp->theta.p + (size_t)S * P;
  } else if (P <= 16) {
    if ((rc = launch_mlp_forward(m, k, S, data, stride, (float*)mlp->tape.p, e_nn, (double*)mlp->shift.p, st, Sop, P, (double*)mlp->theta.p))) return rc;
  } else {
    if ((rc = launch_subfin_avg(Sop, P, m.n_in, k, S, (double*)mlp->theta.p, st))) return rc;
    if ((rc = launch_mlp_forward(m, k, S, data, stride, (float*)mlp->tape.p, e_nn, (double*)mlp->shift.p, st))) return rc;
  }
  MlpBackFuse bf;
  if (fm.on) {
    if ((rc = mlp->g0.reserve((size_t)S * 64 * sizeof(float)))) return rc;
    bf.on = 1; bf.m = m; bf.tape = (const float*)mlp->tape.p; bf.data = data; bf.data_stride = stride; bf.qoi_r = qoi_r; bf.e_nn = e_nn;
    bf.g0_out = (float*)mlp->g0.p;
  }
  // info is this call's alone: stored by the one-sample solve kernel, cleared here for the other forms (whose kernels or flags in)
  if (info != nullptr && !one) FR_HIP(hipMemsetAsync(info, 0, (size_t)S * sizeof(int32_t), st));
  rom->defer_gsum = one; rom->last_gpart = nullptr; rom->fuse = fm.on ? &fm : nullptr; rom->back = bf.on ? &bf : nullptr;
  rom->info_store = one;
  rc = finrom_rom_grad(rom, (const double*)mlp->theta.p, (const double*)mlp->shift.p, 1, S, loss, (double*)mlp->gth.p, nullptr,
                       qoi_r, info, st);
  const double* gparts = rom->last_gpart;
  rom->defer_gsum = false; rom->last_gpart = nullptr; rom->fuse = nullptr; rom->back = nullptr; rom->info_store = false;
  if (rc) return rc;
  if (fm.on && gparts == nullptr) {                    // (the predicate above and finrom_rom_grad's own must agree)
    set_error("romml_grad: internal: the one-sample form was announced but not taken");
    return FINROM_ERR_UNSUPPORTED;
  }
  return launch_mlp_backward(m, S, (const float*)mlp->tape.p, data, stride, qoi_r, e_nn, (const double*)mlp->gth.p, Sop, P, grad, st,
                             gparts, gparts ? ROM_GRAD_SMALL_NG : 0, bf.on && gparts ? (const float*)mlp->g0.p : nullptr,
                             hs != nullptr ? &hs->tail : nullptr);
}

int finrom_romml_grad(finrom_rom_t rom, finrom_mlp_t mlp, const double* Sop, const double* k, const double* data,
                      int32_t data_per_sample, int64_t S, double* grad, double* loss, double* qoi_r, double* e_nn,
                      int32_t* info, void* stream) {
  CallGuard cg((hipStream_t)stream);
  return romml_grad_impl(rom, mlp, Sop, k, data, data_per_sample, S, grad, loss, qoi_r, e_nn, info, stream, nullptr);
}

static int hmc_dev(const finrom_hmc_state* a, HmcDev* h, const char* who) {
  if (!a || a->C < 0 || a->n <= 0 || !a->mean || !a->K || !a->U || !a->dU || !a->Kq[0] || !a->Kq[1] || !a->P || !a->dUq || !a->H0 ||
      !a->P_block || !a->lu_block || !a->jt || !a->pt || !a->accept || !a->loss || !a->info) {
    set_error(std::string(who) + ": null field in finrom_hmc_state"); return FINROM_ERR_ARG;
  }
  h->C = a->C; h->n = a->n; h->eps = a->eps; h->c_lik = a->c_lik; h->c_pri = a->c_pri;
  h->mean = a->mean; h->K = a->K; h->U = a->U; h->dU = a->dU; h->Kq0 = a->Kq[0]; h->P = a->P; h->dUq = a->dUq; h->H0 = a->H0;
  h->P_block = a->P_block; h->lu_block = a->lu_block; h->jt = (long long*)a->jt; h->pt = (long long*)a->pt;
  h->accept = (long long*)a->accept; h->trace = a->trace; h->loss = a->loss; h->info = a->info;
  return 0;
}
int finrom_hmc_begin(const finrom_hmc_state* a, void* stream) {
  CallGuard cg((hipStream_t)stream);
  HmcDev h;
  if (int rc = hmc_dev(a, &h, "hmc_begin")) return rc;
  return launch_hmc_begin(h, (hipStream_t)stream);
}
int finrom_hmc_end(const finrom_hmc_state* a, int32_t n_steps, void* stream) {
  CallGuard cg((hipStream_t)stream);
  HmcDev h;
  if (int rc = hmc_dev(a, &h, "hmc_end")) return rc;
  if (n_steps < 0) { set_error("hmc_end: n_steps < 0"); return FINROM_ERR_ARG; }
  return launch_hmc_end(h, a->Kq[n_steps & 1], (hipStream_t)stream);       // (the position ping-pongs once per step)
}
int finrom_hmc_leapfrog(finrom_rom_t rom, finrom_mlp_t mlp, const double* Sop, const finrom_hmc_state* a, int32_t step,
                        const double* data, int32_t data_per_sample, double* grad_out, double* qoi_r, double* e_nn, void* stream) {
  CallGuard cg((hipStream_t)stream);
  HmcDev h;
  if (int rc = hmc_dev(a, &h, "hmc_leapfrog")) return rc;
  if (!mlp || a->n != mlp->d.n_in || step < 0) { set_error("hmc_leapfrog: bad argument"); return FINROM_ERR_ARG; }
  if (a->c_pri == 0.0) { set_error("hmc_leapfrog: c_pri = 0"); return FINROM_ERR_ARG; }
  HmcStep hs;
  hs.mom = a->P; hs.eps = a->eps; hs.k_out = a->Kq[(step + 1) & 1];
  // theta of THIS step's field from the previous step's momentum update, when this is that field's step (not the first step of a
  // trajectory: finrom_hmc_begin has no averaging operator at hand); and this step leaves the next one's
  if (int rc = mlp->thp.reserve((size_t)a->C * HMC_THETA_PARTS * 16 * sizeof(double))) return rc;
  const bool carried = step > 0 && mlp->carry_k == a->Kq[step & 1] && mlp->carry_mom == a->P && mlp->carry_eps == a->eps &&
                       mlp->carry_S == a->C && getenv("FINROM_HMC_NO_CARRY") == nullptr;
  hs.theta_parts_in = carried ? (const double*)mlp->thp.p : nullptr;
  hs.tail.eps = a->eps; hs.tail.theta_parts = rom->d.P <= HMC_THETA_MAXP ? (double*)mlp->thp.p : nullptr;
  mlp->carry_k = nullptr;
  hs.tail.on = 1; hs.tail.kq = hs.k_out; hs.tail.mean = a->mean; hs.tail.mom = a->P; hs.tail.dU = a->dUq;
  hs.tail.coef = a->c_lik / a->c_pri; hs.tail.eps_cpri = a->eps * a->c_pri; hs.tail.info = a->info;
  const int rc = romml_grad_impl(rom, mlp, Sop, a->Kq[step & 1], data, data_per_sample, a->C, grad_out, a->loss, qoi_r, e_nn, a->info, stream, &hs);
  if (rc == 0 && hs.tail.theta_parts != nullptr) { mlp->carry_k = hs.k_out; mlp->carry_mom = a->P; mlp->carry_eps = a->eps; mlp->carry_S = a->C; }
  return rc;
}

int finrom_sub(const double* a, const double* b, int64_t count, double* out, void* stream) {
  CallGuard cg((hipStream_t)stream);
  if (count < 0 || (count > 0 && (!a || !b || !out))) { set_error("sub: bad argument"); return FINROM_ERR_ARG; }
  return launch_sub(a, b, count, out, (hipStream_t)stream);
}

}  // extern "C"
