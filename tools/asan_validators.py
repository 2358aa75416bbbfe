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
    row = next(i for i in range(d.n) if d.row_ptr[i + 1] - d.row_ptr[i] >= 2)
    tweak(d.term_p, d.row_ptr[row] + 1, d.term_p[d.row_ptr[row]], call, "rom_create: a row lists the same theta index twice")
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

# ---- the band descriptor (finrom_fom_set_band's host-only validator, exported as finrom_fom_band_validate): the product's own
# descriptors for every mesh with a band plan and every parameter kind must pass, one-field corruptions must be rejected -------
import scipy.sparse as sp                                                    # noqa: E402
from bayesianinferencedl_amd.engine import FomEngine                         # noqa: E402


def band_cases(m, kinds, corrupt):
    Vm = get_space(None, m=m)
    ops = Vm.operators()
    bp = ops.band_plan()
    assert bp is not None, m
    fin = Fin(Vm)
    for kind in kinds:
        W = {"field": ops.W_field, "nine": sp.csr_matrix(ops.W_field @ sp.csr_matrix(ops.N9)),
             "five": sp.csr_matrix(ops.W_field @ sp.csr_matrix(ops.N9 @ ops.E59))}[kind]
        xdim = W.shape[1]
        d, keep, _ = FomEngine.band_descriptor(bp, xdim, ops.robin_vals, W, ops.F, fin.B_obs)
        call = lambda: lib.finrom_fom_band_validate(C.byref(d), ops.n, xdim, fin.n_obs)
        rc = call()
        assert rc == 0, (m, kind, lib.finrom_last_error())
        cases.append((f"band descriptor m={m} {kind}: accepted", rc))
        if not corrupt:
            continue
        G = bp.G
        nift = d.nif * (d.nif + 1) // 2
        tweak(d.abmap, 7, d.nAB, call, "band: abmap out of range")
        tweak(d.abmap, 3 * G - 1, -1, call, "band: abmap negative")
        tweak(d.act, 5, 1 << d.NX, call, "band: act has a bit beyond NX")
        p_act = next(p_ for p_ in range(d.npost) if d.act[p_] != 0)
        tweak(d.act, p_act, 0, call, "band: act disagrees with lx_ptr")
        tweak(d.lx_ptr, d.npost, d.nLx + 1, call, "band: lx_ptr does not end at nLx")
        tweak(d.ent_extra, 3, d.NX + 1, call, "band: ent_extra beyond NX")
        tweak(d.schur_off, 1, d.nAB + 3, call, "band: schur_off out of range")
        tweak(d.schur_off, nift, d.schur_off[0], call, "band: two fins share a Schur slot")
        tweak(d.schur_off, 1, d.schur_off[0], call, "band: Schur slot repeated within a fin")
        tweak(d.iface_elim, 0, 0, call, "band: interface node is not a post node")
        tweak(d.iface_elim, 1, ops.n, call, "band: iface_elim out of range")
        tweak(d.perm, 0, d.perm[1], call, "band: perm is not a permutation")
        tweak(d.obs_idx, 0, ops.n, call, "band: observation index out of range")
        tweak(d.ab_idx, 0, xdim, call, "band: parameter index out of range")
        tweak(d.ecp_off, 0, d.nAB, call, "band: ecp_off out of range")
        tweak(d.ecp_slot, 0, d.NX, call, "band: ecp_slot out of range")
        # a logical value slot de-duplicated onto a slot a fin writes to would be a second reader of that slot
        tweak(d.abmap, 0, d.schur_off[0], call, "band: a Schur slot is also somebody's plain value slot")
        for field in ("abmap", "ab_ptr", "ab_c0", "Fg", "act", "lx_ptr", "ent_extra", "ecp_ptr", "ecp_off", "schur_off", "iface_elim",
                      "perm", "obs_ptr", "obs_idx", "qoi_FgQ", "qoi_obs_ptr"):
            ptype = type(getattr(d, field))
            addr = C.cast(getattr(d, field), C.c_void_p).value      # (the attribute is a VIEW of the field: keep the address)
            setattr(d, field, ptype())                  # NULL pointer
            try:
                expect_error(f"band: {field} is NULL", call())
            finally:
                setattr(d, field, C.cast(addr, ptype))
        # the QoI-only tables must describe the same operator as obs_*
        assert bool(d.qoi_FgQ) and bool(d.qoi_row_fin), "QoI-only tables missing for the sub-fin averages"
        o_fin = next(o for o in range(fin.n_obs) if d.qoi_row_fin[o] >= 0)
        f_ = d.qoi_row_fin[o_fin]
        tweak(d.qoi_FgQ, f_ * (d.npf + d.nif) + 3, 0.125, call, "band: QoI-only weight differs from B_obs")
        tweak(d.qoi_FgQ, bp.G - 1, d.Fg[bp.G - 1] + 1.0, call, "band: QoI-only load differs from Fg on the post")
        tweak(d.qoi_row_fin, o_fin, -1, call, "band: a row lost its fin")
        o_post = next(o for o in range(fin.n_obs) if d.qoi_row_fin[o] < 0)
        tweak(d.qoi_row_fin, o_post, f_, call, "band: two rows claim one fin")
        tweak(d.qoi_obs_idx, 0, 0, call, "band: QoI-only remainder points into a fin")
        tweak(d.qoi_obs_w, 0, d.qoi_obs_w[0] * 2, call, "band: QoI-only remainder weight differs")
        tweak(d.Fg, 5, 1.0, call, "band: load on a fin node")
        for field, val in (("npf", d.NSF - 1), ("npost", d.NSP - 1), ("nif", d.nif + 1), ("NSP", 15), ("nAB", 0)):
            old = getattr(d, field); setattr(d, field, val)
            try:
                expect_error(f"band: {field} = {val}", call())
            finally:
                setattr(d, field, old)
        assert call() == 0, ("descriptor not restored", lib.finrom_last_error())


band_cases(12, ("five",), corrupt=True)
band_cases(20, ("field",), corrupt=True)
for m_ in (4, 8, 16):
    band_cases(m_, ("field", "nine", "five"), corrupt=False)
print(f"ASAN-VALIDATORS-OK {len(cases)} cases")
