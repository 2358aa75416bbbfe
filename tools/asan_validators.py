"""Drive the create-time validators of libfinrom_hip.so through an AddressSanitizer / UBSan build of the C-ABI layer, on a box
WITHOUT a GPU: every engine's descriptor is built by the product's own Python code, corrupted in one place at a time and handed
to the library; a corrupt descriptor must come back as an error code before any device call (and without a sanitizer report).
Run by tests/test_host_and_abi.py::test_validators_under_address_sanitizer with the ASan runtime preloaded:
    LD_PRELOAD=<libclang_rt.asan> FINROM_LIB=<lib/libfinrom_hip_asan.so> python tools/asan_validators.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianinferencedl_amd import _ffi                                    # noqa: E402
from bayesianinferencedl_amd.fom.thermal_fin import get_space                # noqa: E402
from bayesianinferencedl_amd.fom.forward_solve import Fin                    # noqa: E402
from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin     # noqa: E402

lib = _ffi.lib()
cases = []


def expect_error(name, rc):
    cases.append((name, rc))
    assert rc != 0, f"{name}: a corrupt descriptor was accepted"
    assert lib.finrom_last_error(), name


def tweak(arr, idx, value, call, name):
    old = arr[idx]
    arr[idx] = value
    try:
        expect_error(name, call())
    finally:
        arr[idx] = old


real_fom_create, real_set_small, real_set_band, real_rom_create = (lib.finrom_fom_create, lib.finrom_fom_set_small,
                                                                   lib.finrom_fom_set_band, lib.finrom_rom_create)


def spy_fom_create(dref, href):
    d = dref._obj
    h = C.c_void_p()
    call = lambda: real_fom_create(C.byref(d), C.byref(h))
    tweak(d.fwd_a, 5, d.nnzL + 3 * d.n + 7, call, "fom_create: operand index outside the value vector")
    tweak(d.fwd_a, 9, -5, call, "fom_create: negative operand index")
    tweak(d.perm, 0, d.perm[1], call, "fom_create: perm is not a permutation")
    tweak(d.asm_idx, 0, d.xdim, call, "fom_create: parameter index out of range")
    tweak(d.obs_idx, 0, d.n, call, "fom_create: observation index out of range")
    tweak(d.bwd_b, 3, 1 << 30, call, "fom_create: backward operand out of range")
    old = d.fwd_chunk; d.fwd_chunk = 12; expect_error("fom_create: chunk size", call()); d.fwd_chunk = old
    old = d.nops_fwd; d.nops_fwd = old - 1; expect_error("fom_create: stream length", call()); d.nops_fwd = old
    return real_fom_create(dref, href)          # no device here: an error code, not a crash


def spy_rom_create(dref, href):
    d = dref._obj
    h = C.c_void_p()
    call = lambda: real_rom_create(C.byref(d), C.byref(h))
    tweak(d.term_p, 0, d.P + 1, call, "rom_create: theta index out of range")
    tweak(d.row_ptr, 1, -1, call, "rom_create: row_ptr not monotone")
    old = d.r; d.r = 209; expect_error("rom_create: basis too wide", call()); d.r = old
    return real_rom_create(dref, href)


lib.finrom_fom_create, lib.finrom_rom_create = spy_fom_create, spy_rom_create
V = get_space(None, m=4)
try:
    Fin(V)._engine("nine")
except _ffi.FinromError as e:                   # the valid descriptor fails at the first device allocation: an error, no crash
    cases.append(("fom_create on a box without a GPU", str(e)[:60]))
try:
    AffineROMFin(V, None, np.linalg.qr(np.random.default_rng(0).standard_normal((V.dim(), 8)))[0])
except _ffi.FinromError as e:
    cases.append(("rom_create on a box without a GPU", str(e)[:60]))
# entry points that take plain arrays
h = C.c_void_p()
expect_error("sampler_create: lower factor", lib.finrom_sampler_create(np.tril(np.ones((4, 4))).ctypes.data_as(_ffi.c_f64p), 4, C.byref(h)))
expect_error("fom_solve: null handle", lib.finrom_fom_solve(None, None, 4, None, None, None, None))
expect_error("rom_solve: null handle", lib.finrom_rom_solve(None, None, 4, None, None, None, None, None, None))
expect_error("solve_pairs: null handles", lib.finrom_solve_pairs(None, None, None, None, 1, None, None, None, None, None, None, None, None))
expect_error("fom_set_band: null", lib.finrom_fom_set_band(None, None))
print(f"ASAN-VALIDATORS-OK {len(cases)} cases")
