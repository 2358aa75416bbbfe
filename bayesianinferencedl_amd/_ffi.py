"""ctypes binding of libfinrom_hip.so (include/finrom.h).  Thin by design: every batched
operation of the hot path is one call into the HIP library.  There is NO CPU fallback --
if the library is missing or a call fails, this module raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)


class FinromError(RuntimeError):
    pass


class FomDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("nnzL", C.c_int32), ("xdim", C.c_int32), ("n_obs", C.c_int32),
                ("nasm", C.c_int32), ("n_alist", C.c_int32), ("cache_slots", C.c_int32),
                ("fwd_chunk", C.c_int32), ("nops_fwd", C.c_int32), ("nops_bwd", C.c_int32),
                ("a_list", c_i32p), ("asm_c0", c_f64p), ("asm_ptr", c_i32p), ("asm_idx", c_i32p),
                ("asm_w", c_f64p), ("rhs", c_f64p),
                ("fwd_kind", c_i32p), ("fwd_a", c_i32p), ("fwd_b", c_i32p), ("fwd_d", c_i32p),
                ("bwd_kind", c_i32p), ("bwd_a", c_i32p), ("bwd_b", c_i32p), ("bwd_d", c_i32p),
                ("obs_ptr", c_i32p), ("obs_idx", c_i32p), ("obs_w", c_f64p), ("perm", c_i32p),
                ("n_imm", C.c_int32), ("imm", c_f64p)]


class FomGradDesc(C.Structure):
    _fields_ = [("nops_res", C.c_int32),
                ("res_kind", c_i32p), ("res_a", c_i32p), ("res_b", c_i32p), ("res_d", c_i32p),
                ("bt_ptr", c_i32p), ("bt_obs", c_i32p), ("bt_w", c_f64p),
                ("g_ptr", c_i32p), ("g_a", c_i32p), ("g_b", c_i32p), ("g_w", c_f64p)]


class FomBandGradDesc(C.Structure):
    _fields_ = [("bt_ptr", c_i32p), ("bt_obs", c_i32p), ("bt_w", c_f64p),
                ("g_ptr", c_i32p), ("g_a", c_i32p), ("g_b", c_i32p), ("g_w", c_f64p)]


class FomSmallDesc(C.Structure):
    _fields_ = [("small_max", C.c_int32), ("npairs", C.c_int32), ("nasm", C.c_int32), ("nlev_f", C.c_int32), ("nlev_b", C.c_int32),
                ("row_ptr", c_i32p), ("ent_col", c_i32p),
                ("pair_ptr", c_i32p), ("pair_a", c_i32p), ("pair_b", c_i32p),
                ("asm_c0", c_f64p), ("asm_ptr", c_i32p), ("asm_idx", c_i32p), ("asm_w", c_f64p),
                ("col_ptr", c_i32p), ("col_ent", c_i32p), ("col_row", c_i32p),
                ("lev_ptr_f", c_i32p), ("lev_rows_f", c_i32p), ("lev_ptr_b", c_i32p), ("lev_rows_b", c_i32p)]


class FomBandDesc(C.Structure):
    _fields_ = [("NSF", C.c_int32), ("NSP", C.c_int32), ("NX", C.c_int32), ("nfins", C.c_int32), ("npf", C.c_int32),
                ("nif", C.c_int32), ("npost", C.c_int32), ("nAB", C.c_int32), ("nterms", C.c_int32), ("nLx", C.c_int32),
                ("ab_c0", c_f64p), ("ab_ptr", c_i32p), ("ab_idx", c_i32p), ("ab_w", c_f64p), ("abmap", c_i32p), ("Fg", c_f64p),
                ("act", c_i32p), ("lx_ptr", c_i32p), ("ent_extra", c_i32p),
                ("ecp_ptr", c_i32p), ("ecp_slot", c_i32p), ("ecp_off", c_i32p),
                ("schur_off", c_i32p), ("iface_elim", c_i32p), ("perm", c_i32p),
                ("obs_ptr", c_i32p), ("obs_idx", c_i32p), ("obs_w", c_f64p),
                ("qoi_FgQ", c_f64p), ("qoi_row_fin", c_i32p), ("qoi_obs_ptr", c_i32p), ("qoi_obs_idx", c_i32p), ("qoi_obs_w", c_f64p)]


c_f32p = C.POINTER(C.c_float)


class MlpDesc(C.Structure):
    _fields_ = [("n_in", C.c_int32), ("n_w", C.c_int32), ("n_layers", C.c_int32), ("n_out", C.c_int32),
                ("W0", c_f32p), ("b0", c_f32p), ("scale", c_f32p), ("shift", c_f32p), ("W", c_f32p), ("b", c_f32p),
                ("Wh", c_f32p), ("bh", c_f32p)]


class HmcState(C.Structure):
    _fields_ = [("C", C.c_int64), ("n", C.c_int32), ("eps", C.c_double), ("c_lik", C.c_double), ("c_pri", C.c_double),
                ("mean", C.c_void_p), ("K", C.c_void_p), ("U", C.c_void_p), ("dU", C.c_void_p),
                ("Kq", C.c_void_p * 2), ("P", C.c_void_p), ("dUq", C.c_void_p), ("H0", C.c_void_p),
                ("P_block", C.c_void_p), ("lu_block", C.c_void_p), ("jt", C.c_void_p), ("pt", C.c_void_p),
                ("accept", C.c_void_p), ("trace", C.c_void_p), ("loss", C.c_void_p), ("info", C.c_void_p)]


class RomDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("r", C.c_int32), ("P", C.c_int32), ("n_obs", C.c_int32),
                ("nterms", C.c_int32),
                ("row_ptr", c_i32p), ("term_p", c_i32p), ("term_val", c_f64p), ("rhs", c_f64p),
                ("obs_phi", c_f64p)]


# name -> (restype, argtypes); tests/test_abi.py checks every symbol of include/finrom.h is here
SIGNATURES = {
    "finrom_version": (C.c_int, []),
    "finrom_last_error": (C.c_char_p, []),
    "finrom_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "finrom_set_device": (C.c_int, [C.c_int]),
    "finrom_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "finrom_free": (C.c_int, [C.c_void_p]),
    "finrom_free_async": (C.c_int, [C.c_void_p, C.c_void_p]),
    "finrom_note_stream": (C.c_int, [C.c_void_p]),
    "finrom_deferred_count": (C.c_int, []),
    "finrom_flush_deferred": (C.c_int, []),
    "finrom_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "finrom_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "finrom_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "finrom_stream_sync": (C.c_int, [C.c_void_p]),
    "finrom_profile_enable": (C.c_int, [C.c_int]),
    "finrom_set_overlap": (C.c_int, [C.c_int]),
    "finrom_profile_reset": (C.c_int, []),
    "finrom_profile_slots": (C.c_int, []),
    "finrom_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "finrom_fom_create": (C.c_int, [C.POINTER(FomDesc), C.POINTER(C.c_void_p)]),
    "finrom_fom_destroy": (None, [C.c_void_p]),
    "finrom_fom_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "finrom_fom_set_small": (C.c_int, [C.c_void_p, C.POINTER(FomSmallDesc)]),
    "finrom_fom_set_gradient": (C.c_int, [C.c_void_p, C.POINTER(FomGradDesc)]),
    "finrom_fom_set_band": (C.c_int, [C.c_void_p, C.POINTER(FomBandDesc)]),
    "finrom_fom_set_band_gradient": (C.c_int, [C.c_void_p, C.POINTER(FomBandGradDesc)]),
    "finrom_fom_band_validate": (C.c_int, [C.POINTER(FomBandDesc), C.c_int32, C.c_int32, C.c_int32]),
    "finrom_fom_solve_rhs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "finrom_fom_last_path": (C.c_int, [C.c_void_p]),
    "finrom_fom_set_small_max": (C.c_int, [C.c_void_p, C.c_int32]),
    "finrom_fom_gradient": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64] + [C.c_void_p] * 5),
    "finrom_rom_create": (C.c_int, [C.POINTER(RomDesc), C.POINTER(C.c_void_p)]),
    "finrom_rom_grouped_tables": (C.c_int, [C.POINTER(RomDesc), c_i32p, c_i32p, c_i32p, C.POINTER(C.c_int64), c_i32p, c_f64p, c_i32p]),
    "finrom_rom_destroy": (None, [C.c_void_p]),
    "finrom_rom_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "finrom_rom_set_gradient": (C.c_int, [C.c_void_p, C.c_int32, c_i32p, c_i32p, c_f64p]),
    "finrom_rom_set_gram": (C.c_int, [C.c_void_p, C.c_int32, c_i32p, c_i32p, c_f64p]),
    "finrom_rom_set_projection": (C.c_int, [C.c_void_p, C.c_int32]),
    "finrom_rom_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64] + [C.c_void_p] * 6),
    "finrom_subfin_avg": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "finrom_sampler_create": (C.c_int, [c_f64p, C.c_int32, C.POINTER(C.c_void_p)]),
    "finrom_sampler_destroy": (None, [C.c_void_p]),
    "finrom_sampler_draw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "finrom_sampler_draw_seeded": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "finrom_mlp_create": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(C.c_void_p)]),
    "finrom_mlp_destroy": (None, [C.c_void_p]),
    "finrom_mlp_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "finrom_romml_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64] + [C.c_void_p] * 6),
    "finrom_hmc_begin": (C.c_int, [C.POINTER(HmcState), C.c_void_p]),
    "finrom_hmc_leapfrog": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(HmcState), C.c_int32, C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "finrom_hmc_end": (C.c_int, [C.POINTER(HmcState), C.c_int32, C.c_void_p]),
    "finrom_solve_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 8),
    "finrom_sub": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "finrom_comm_unique_id": (C.c_int, [C.c_void_p]),
    "finrom_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]),
    "finrom_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "finrom_comm_destroy": (C.c_int, [C.c_void_p]),
}

ABI_VERSION = 12
# finrom_fom_last_path codes (include/finrom.h)
FOM_PATHS = {0: "none", 1: "small_lds", 2: "small_global", 3: "interpreter", 4: "band_registers", 5: "band_lds_4wave",
             6: "band_lds_1wave", 7: "band_registers_qoi", 8: "band_lds_4wave_qoi"}

_lib = None


def lib_path() -> str:
    return os.environ.get("FINROM_LIB", _build.LIB)


def _hip_runtime_loaded() -> bool:
    """Is a HIP runtime (any libamdhip64) already mapped into this process -- torch's, /opt/rocm's under rocprofv3's preloaded
    tool library, or another ROCm-linked module's?"""
    try:
        with open("/proc/self/maps") as f:
            return any("libamdhip64" in line for line in f)
    except OSError:
        pass
    try:                                                  # no /proc: ask the loader without loading anything (RTLD_NOLOAD = 4)
        C.CDLL("libamdhip64.so.7", mode=4)
        return True
    except OSError:
        return False


def _share_torch_hip_runtime():
    """ONE HIP / HSA runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so.7 + libhsa-runtime64.so.1; this
    library is linked against /opt/rocm's.  With torch imported first the loader resolves our DT_NEEDED to torch's copy (same
    SONAME) and everything shares one runtime; with OUR library loaded first the process ends up with two HSA runtimes and the
    second one to initialise (torch) finds no GPU ("No HIP GPUs are available").  So: if torch is installed and not yet imported,
    its two runtime libraries are loaded first, without importing torch -- unless a HIP runtime is ALREADY mapped (rocprofv3's
    preloaded tool library, another /opt/rocm-linked module): loading torch's copy by path on top of it would itself create the
    two-runtime state.  FINROM_SYSTEM_HIP=1 keeps /opt/rocm's."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("FINROM_SYSTEM_HIP") or _hip_runtime_loaded():
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(d, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    """Load (once) and return the shared library; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        _share_torch_hip_runtime()
        path = lib_path()
        if not os.path.exists(path):
            raise FinromError(
                f"{path} not found: the HIP extension is not built "
                "(run `python -m bayesianinferencedl_amd._build` or __graft_entry__.build()); "
                "there is no CPU fallback for the hot path")
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.finrom_version() != ABI_VERSION:
            raise FinromError("libfinrom_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().finrom_last_error().decode(errors="replace")
        raise FinromError(f"{what or 'finrom call'} failed (status {rc}): {msg}")


def i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_i32p)


def f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_f64p)


def note_current_stream():
    """Tell the library about torch's current stream before anything is freed or destroyed: a finaliser may run inside somebody's
    `torch.cuda.graph` capture (a generational GC pass between two captured calls), where hipFree would invalidate the capture or
    wait on it.  The library queues the work while that stream captures (include/finrom.h, finrom_note_stream).  Returns the
    stream handle (or None when torch has not initialised CUDA in this process)."""
    import sys
    t = sys.modules.get("torch")
    if t is None or _lib is None:
        return None
    try:
        if not t.cuda.is_initialized():
            return None
        st = t.cuda.current_stream().cuda_stream
        _lib.finrom_note_stream(st)
        return st
    except Exception:                                     # interpreter shutdown: torch half torn down
        return None


def destroy_handle(fn_name, handle):
    """finrom_*_destroy from a close() / finaliser: capture-safe (see note_current_stream)."""
    note_current_stream()
    getattr(lib(), fn_name)(handle)


class DeviceBuffer:
    """Device allocation owned by Python (finrom_malloc / finrom_free).  Small buffers are recycled by the library's pool
    (finrom_malloc: <= 16 MiB size classes; hipMalloc / hipFree cost more than the kernels of a one-sample call, the MAP / HMC call
    pattern of SURVEY configs[4]).  `used_on(stream)`: a launch on a non-default stream consumed the buffer -- it is then handed back
    with finrom_free_async, which parks it behind an event on that stream."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        self._stream = None
        p = C.c_void_p()
        check(lib().finrom_malloc(C.byref(p), max(self.nbytes, 8)), "finrom_malloc")
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a, stream=None):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        check(lib().finrom_memcpy_h2d(b.ptr, a.ctypes.data, a.nbytes, stream), "memcpy_h2d")
        return b

    def to_numpy(self, shape, dtype=np.float64, stream=None):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= max(self.nbytes, 8)
        check(lib().finrom_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, stream), "memcpy_d2h")
        return out

    def zero(self, stream=None):
        check(lib().finrom_memset(self.ptr, 0, self.nbytes, stream), "memset")

    def used_on(self, stream):
        if stream:
            self._stream = stream

    def free(self):
        if getattr(self, "ptr", None):
            ptr, self.ptr = self.ptr, None
            note_current_stream()
            if self._stream:
                lib().finrom_free_async(ptr, self._stream)
            else:
                lib().finrom_free(ptr)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def profile_read():
    """-> {kernel name: (launches, total_ms)} from the library's HIP-event timers."""
    L = lib()
    out = {}
    for s in range(L.finrom_profile_slots()):
        name = C.c_char_p(); cnt = C.c_int64(); ms = C.c_double()
        check(L.finrom_profile_read(s, C.byref(name), C.byref(cnt), C.byref(ms)))
        out[name.value.decode()] = (cnt.value, ms.value)
    return out
