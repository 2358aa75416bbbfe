"""CPU tests of the host logic and of the C-ABI library surface (no compute calls: no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from oracle import fin_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    from bayesianinferencedl_amd import _build, _ffi
    path = _build.build()
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "finrom.h")).read()
    declared = set(re.findall(r"\b(finrom_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in finrom.h but not exported"
    assert declared == set(_ffi.SIGNATURES), declared ^ set(_ffi.SIGNATURES)
    lib.finrom_version.restype = ctypes.c_int
    assert lib.finrom_version() == _ffi.ABI_VERSION == 12


def test_missing_library_fails_loudly(monkeypatch):
    from bayesianinferencedl_amd import _ffi
    monkeypatch.setenv("FINROM_LIB", "/nonexistent/libfinrom_hip.so")
    monkeypatch.setattr(_ffi, "_lib", None)
    with pytest.raises(_ffi.FinromError, match="no CPU fallback"):
        _ffi.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bayesianinferencedl_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(d, f)


@pytest.mark.parametrize("m", [4, 12])
def test_package_operators_match_oracle(problems, spaces, m):
    prob = problems(m); ops = spaces(m).operators()
    assert np.array_equal(prob.coords, ops.mesh.coords)
    rng = np.random.default_rng(0)
    k = np.exp(0.3 * rng.standard_normal(prob.n)); th = rng.uniform(0.1, 10, 9)
    assert abs(prob.assemble_fom_loops(k) - ops.csr(ops.fom_values(k))).max() < 1e-13
    assert abs(prob.assemble_affine(th) - ops.csr(ops.affine_values(th))).max() < 1e-13
    assert np.allclose(prob.S, ops.S, atol=1e-16) and np.array_equal(prob.B, ops.F)
    fo = O.FinOracle(prob)
    k9 = rng.uniform(0.1, 10, 9); k5 = rng.uniform(0.1, 10, 5)
    assert np.array_equal(fo.nine_param_to_function(k9), ops.N9 @ k9)
    assert np.array_equal(fo.five_param_to_function(k5), ops.N9 @ (ops.E59 @ k5))
    assert sorted(map(tuple, ops.mesh.robin_facets.tolist())) == sorted(prob.robin)


@pytest.mark.parametrize("ordering", ["auto", "md", "rcm", "natural"])
def test_cholesky_plan_schedule_reproduces_the_solve(spaces, ordering):
    """Execute the device schedule in NumPy (same order of operations as fom_kernels.hip)."""
    from bayesianinferencedl_amd.symbolic import CholeskyPlan
    ops = spaces(4).operators()
    plan = CholeskyPlan(ops.indptr, ops.indices, ops.n, ordering)
    rng = np.random.default_rng(1)
    k = np.exp(0.3 * rng.standard_normal(ops.n))
    c0, ptr, idx, w = plan.entry_table(ops.robin_vals, ops.W_field)
    L = np.zeros(plan.nnzL); invd = np.zeros(ops.n); y = np.zeros(ops.n)
    F = ops.F[plan.perm]
    for i in range(ops.n):
        for e in range(plan.row_ptr[i], plan.row_ptr[i + 1]):
            acc = c0[e] + w[ptr[e]:ptr[e + 1]] @ k[idx[ptr[e]:ptr[e + 1]]]
            q = slice(plan.pair_ptr[e], plan.pair_ptr[e + 1])
            acc -= L[plan.pair_a[q]] @ L[plan.pair_b[q]]
            if e == plan.row_ptr[i + 1] - 1:
                L[e] = np.sqrt(acc); invd[i] = 1 / L[e]
            else:
                L[e] = acc * invd[plan.ent_col[e]]
        off = slice(plan.row_ptr[i], plan.row_ptr[i + 1] - 1)
        y[i] = (F[i] - L[off] @ y[plan.ent_col[off]]) * invd[i]
    for i in range(ops.n - 1, -1, -1):
        c = slice(plan.col_ptr[i], plan.col_ptr[i + 1])
        y[i] = (y[i] - L[plan.col_ent[c]] @ y[plan.col_row[c]]) * invd[i]
    wsol = np.empty(ops.n); wsol[plan.perm] = y
    ref = spl.spsolve(ops.csr(ops.fom_values(k)).tocsc(), ops.F)
    assert np.linalg.norm(wsol - ref) < 1e-12 * np.linalg.norm(ref)
    assert plan.npairs == sum(len(a) for a in [plan.pair_a])


@pytest.mark.parametrize("m,cache,fwd_chunk", [(4, 36, 8), (4, 3, 8), (12, 36, 8), (4, 5, 16), (12, 40, 16)])
def test_op_streams_replay_with_prefetch_semantics(spaces, m, cache, fwd_chunk):
    """The device interpreter fetches the operands of chunk c+1 before executing chunk c; the
    NumPy replay does the same, so a scheduling/padding bug shows up as a wrong solution."""
    from bayesianinferencedl_amd.symbolic import CholeskyPlan, build_op_streams, replay_op_streams, CHUNK
    ops = spaces(m).operators()
    plan = CholeskyPlan(ops.indptr, ops.indices, ops.n)
    st = build_op_streams(plan, cache, None, fwd_chunk)
    for name, ch in (("fwd", fwd_chunk), ("bwd", CHUNK)):
        k = st[name][0]
        assert len(k) % (2 * ch) == 0 and (k[-2 * ch:] == 0).all()
    rng = np.random.default_rng(2)
    kf = np.exp(0.3 * rng.standard_normal(ops.n))
    vals = ops.fom_values(kf)
    Aent = np.zeros(plan.nnzL); has = plan.a_ent >= 0; Aent[has] = vals[plan.a_ent[has]]
    wp = replay_op_streams(plan, st, Aent, ops.F[plan.perm], cache)
    w = np.empty(ops.n); w[plan.perm] = wp
    ref = spl.spsolve(ops.csr(vals).tocsc(), ops.F)
    assert np.linalg.norm(w - ref) < 1e-12 * np.linalg.norm(ref)
    k, a, b, d = st["fwd"]
    real = ((k != 0) | (b != cache + 1)).sum()            # everything but the ZERO-slot padding
    n_a = (plan.a_ent >= 0).sum()
    n_ldx = (k == 3).sum()                                # spilled row entries cost one extra op each
    assert real - n_ldx == n_a + plan.npairs + plan.nnzL + (plan.nnzL - ops.n) + 2 * ops.n


def test_function_space_shim(spaces):
    from bayesianinferencedl_amd.fem import Function
    V = spaces(4)
    assert V.dim() == len(V.dofmap().dofs()) == V.mesh().num_vertices() == 245
    assert V.tabulate_dof_coordinates().shape == (245, 2)
    f = Function(V); g = Function(V)
    f.vector().set_local(np.arange(245.0)); g.assign(f)
    g.vector().axpy(2.0, f.vector())
    assert np.array_equal(g.vector()[:], 3 * np.arange(245.0))
    v = g.vector()[:]; v[0] = -1
    assert g.vector()[0] == 0.0                      # slices are copies, as with dolfin vectors


def test_get_space_resolution_mapping():
    from bayesianinferencedl_amd.fom.thermal_fin import get_space
    assert get_space(40).dim() == 1597               # reference: resolution 40 -> 1446 (mshr)
    assert get_space(40) is get_space(40)


def test_gaussian_field_matches_oracle(problems, spaces):
    from bayesianinferencedl_amd.bayesian_inference.gaussian_field import make_cov_chol
    from bayesianinferencedl_amd.fem import deterministic_blas
    for kern in ("m52", "sq_exp", "m32"):
        with deterministic_blas():       # the product pins LAPACK to one thread (same factor in every rank of a multi-GPU run)
            ref = O.make_cov_chol(problems(4).coords, kern, 1.6)
        assert np.array_equal(make_cov_chol(spaces(4), kern, 1.6), ref)


def test_shard_bounds_cover_everything():
    from bayesianinferencedl_amd.distributed import shard_bounds
    for total in (0, 1, 7, 100000, 1000003):
        for world in (1, 2, 4, 8):
            b = [shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))


def _bench_json(argv, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    import json
    return json.loads(lines[0])


def test_bench_gpus2_launches_two_ranks_and_matches_single_process():
    """N > 1 on CPU through the SAME entry the driver uses: `python bench.py --gpus 2` must itself start two rank processes
    (children of a parent that never touched a GPU), each rank builds its shard of the globally keyed input stream, the shards
    are gathered (gloo) and must reproduce the one-process stream bit for bit (SURVEY 8(e) determinism).  --dry-run replaces
    the HIP solve by a stand-in (no GPU here); tests/test_gpu_multirank.py runs the real solve the same way on the GPU box."""
    two = _bench_json(["--gpus", "2", "--dry-run", "--samples", "5000"])
    one = _bench_json(["--gpus", "1", "--dry-run", "--samples", "10000"])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["dry_run"] is True
    assert two["gathered_sha256"] == one["gathered_sha256"]
    nine = _bench_json(["--gpus", "2", "--dry-run", "--samples", "1000", "--params", "field"])
    assert nine["n_gpus"] == 2


def test_hmc_entry_points_validate_their_state_before_any_device_call():
    """finrom_hmc_begin / _leapfrog / _end reject a state with null fields, a negative step and a field size that is not the error
    model's with FINROM_ERR_ARG and a message -- host-side checks, no GPU needed."""
    import ctypes as C
    from bayesianinferencedl_amd import _ffi
    L = _ffi.lib()
    st = _ffi.HmcState(C=2, n=8, eps=0.1, c_lik=1.0, c_pri=1.0)
    assert L.finrom_hmc_begin(C.byref(st), None) == -1 and b"null field" in L.finrom_last_error()
    assert L.finrom_hmc_end(C.byref(st), 10, None) == -1
    assert L.finrom_hmc_leapfrog(None, None, None, C.byref(st), 0, None, 0, None, None, None, None) == -1
    assert L.finrom_hmc_begin(None, None) == -1
    assert L.finrom_deferred_count() == 0 and L.finrom_flush_deferred() == 0 and L.finrom_note_stream(None) == 0


def test_build_dependencies_follow_the_includes():
    """_build._deps: a source is rebuilt when a header it includes (transitively) changes -- and only then: the band sweep's
    translation units (minutes each) do not depend on the reduced model's headers."""
    import os
    from bayesianinferencedl_amd import _build as B
    deps = {s_: {os.path.basename(d) for d in B._deps(os.path.join(B.CSRC, s_))} for s_ in B.SOURCES}
    assert {"finrom_core.h", "fom_band_device.h", "finrom.h"} <= deps["fom_band.hip"] and "finrom_internal.h" not in deps["fom_band.hip"]
    assert "fom_band.hip" in deps["fom_band_wide.hip"]
    assert {"finrom_internal.h", "finrom_core.h", "rom_proj_device.h", "mlp_device.h"} <= deps["rom_onesample.hip"]
    assert all("finrom_core.h" in d for d in deps.values())
    assert set(B.SOURCES) == {f for f in os.listdir(B.CSRC) if f.endswith(".hip")}


def test_bench_hmc_chain_deal_and_gather_over_two_ranks():
    """configs[4]'s multi-rank form on CPU (`--workload hmc --gpus 2 --dry-run`): chains dealt round robin (rank g owns g, g + N, ...),
    end states / accept counts / traces gathered on rank 0 in chain order; the checksums must equal the one-process run's, also when
    the chains do not divide by the ranks (4 chains on 3 ranks).  tests/test_gpu_multirank.py runs the real chains the same way."""
    one = _bench_json(["--workload", "hmc", "--gpus", "1", "--dry-run", "--steps", "40", "--chains", "4"])
    for n in (2, 3):
        got = _bench_json(["--workload", "hmc", "--gpus", str(n), "--dry-run", "--steps", "40", "--chains", "4"])
        assert got["n_gpus"] == n and got["chains_gathered"] == 4 and got["dry_run"] is True
        assert got["chains_sha256"] == one["chains_sha256"] and got["trace_sha256"] == one["trace_sha256"] is not None


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=120, env=dict(os.environ, WORLD_SIZE="3", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stdout + r.stderr)


def test_global_input_streams_do_not_depend_on_the_shard_cut():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    for gen, dim in ((b.global_uniform, 5), (b.global_normal, 7)):
        full = gen(3, 0, 10000, dim)
        for cut in (1, 4095, 4096, 4097, 9999):
            assert np.array_equal(np.concatenate([gen(3, 0, cut, dim), gen(3, cut, 10000, dim)]), full)
        assert gen(3, 500, 500, dim).shape == (0, dim)


def test_reference_import_paths_resolve():
    """`from fom.forward_solve import Fin` etc. (the reference's import paths) resolve to the HIP-backed classes."""
    from fom.forward_solve import Fin
    from fom.thermal_fin import get_space
    from rom.averaged_affine_ROM import AffineROMFin
    from bayesian_inference.gaussian_field import make_cov_chol
    from deep_learning.generate_fin_dataset import gen_affine_avg_rom_dataset
    import bayesianinferencedl_amd.fom.forward_solve as impl
    assert Fin is impl.Fin and callable(get_space) and callable(make_cov_chol) and callable(gen_affine_avg_rom_dataset)
    assert AffineROMFin.__module__ == "bayesianinferencedl_amd.rom.averaged_affine_ROM"


def test_error_model_vjp_matches_finite_differences():
    """deep_learning/dl_model.py::ResBnFcModel (stand-in for the reference's res_bn_fc_model): the vector-Jacobian product
    behind tf.gradients(loss, model.input) against central differences (fp32 network: loose tolerance)."""
    from bayesianinferencedl_amd.deep_learning.dl_model import ResBnFcModel
    rng = np.random.default_rng(0)
    m = ResBnFcModel(n_in=30, n_out=9, n_layers=3, n_weights=16, seed=3)
    for u in m.units + [m.head]:                       # non-trivial batch-norm statistics
        u["gamma"] = rng.uniform(0.5, 1.5, u["gamma"].shape).astype(np.float32)
        u["beta"] = rng.normal(0, 0.3, u["beta"].shape).astype(np.float32)
        u["mean"] = rng.normal(0, 0.3, u["mean"].shape).astype(np.float32)
        u["var"] = rng.uniform(0.5, 2.0, u["var"].shape).astype(np.float32)
    x = rng.normal(0, 1, (2, 30))
    up = rng.normal(0, 1, (2, 9))
    g = m.vjp(x, up)
    assert g.shape == (2, 30) and m.predict([[x[0]]]).shape == (1, 9)
    eps = 1e-2
    for j in (0, 7, 29):
        e = np.zeros(30); e[j] = eps
        fd = ((m.predict(x + e).astype(np.float64) - m.predict(x - e).astype(np.float64)) / (2 * eps) * up).sum(axis=1)
        assert np.allclose(fd, g[:, j], rtol=3e-2, atol=3e-3)


@pytest.mark.parametrize("params", ["five", "nine"])
def test_fused_assembly_stream_replays_to_the_same_solution(spaces, params):
    """Short parameter vectors: the op stream assembles A_e = c0_e + sum_t w_t x[idx_t] itself (XFMA / CADD ops with
    immediates) instead of reading the pre-pass's output; the NumPy replay with prefetch semantics must still solve A w = F."""
    import scipy.sparse as sp
    from bayesianinferencedl_amd.symbolic import CholeskyPlan, build_op_streams, replay_op_streams, OP_XFMA, OP_CADD
    ops = spaces(4).operators()
    plan = CholeskyPlan(ops.indptr, ops.indices, ops.n)
    lift = ops.N9 if params == "nine" else ops.N9 @ ops.E59
    W = sp.csr_matrix(ops.W_field @ sp.csr_matrix(lift))
    tab = plan.entry_table(ops.robin_vals, W)
    cache = 12
    st = build_op_streams(plan, cache, None, 8, tab)
    k = st["fwd"][0]
    assert len(st["a_list"]) == 0 and (k == OP_XFMA).sum() == len(tab[3]) and (k == OP_CADD).sum() == (tab[0] != 0).sum()
    # the immediates are de-duplicated (a lattice stiffness has a few dozen distinct weights): one small table
    used = st["fwd"][3][(k == OP_XFMA) | (k == OP_CADD)]
    assert len(st["imm"]) == len(set(st["imm"].tolist())) and set(used.tolist()) == set(range(len(st["imm"])))
    rng = np.random.default_rng(6)
    x = rng.uniform(0.1, 10.0, lift.shape[1])
    wp = replay_op_streams(plan, st, np.zeros(plan.nnzL), ops.F[plan.perm], cache, x=x)
    w = np.empty(ops.n); w[plan.perm] = wp
    ref = spl.spsolve(ops.csr(ops.fom_values(lift @ x)).tocsc(), ops.F)
    assert np.linalg.norm(w - ref) < 1e-12 * np.linalg.norm(ref)


@pytest.mark.parametrize("m", [4, 8, 12, 20, 24, 28])      # 20: the plan of the four-wave variant (8 extras); 24: ten extras; 28: twelve
def test_band_plan_replay_solves_the_fom(spaces, m):
    """bandplan.py: fin-by-fin, then up the post -- the tables of the frontal band sweep (windows of NS slots renamed
    cyclically, fin Schur complements added to the post's entries, long-range couplings carried as extras) replayed in NumPy
    with the device kernel's data flow must solve A(k) w = F for every operator table the engines use."""
    import scipy.sparse as sp
    ops = spaces(m).operators()
    bp = ops.band_plan()
    assert bp is not None and (bp.NSF, bp.NSP) == (m // 4 + 2, m + 2) and bp.NX <= (4 if m <= 12 else 8 if m <= 20 else 10 if m <= 24 else 12)
    assert sorted(bp.perm.tolist()) == list(range(ops.n))
    assert bp.lx_ptr[-1] == bp.nLx and bp.nLx == sum(bin(int(a)).count("1") for a in bp.act)
    rng = np.random.default_rng(m)
    for W, x in ((ops.W_field, np.exp(0.5 * rng.standard_normal(ops.n))),
                 (sp.csr_matrix(ops.W_field @ sp.csr_matrix(ops.N9 @ ops.E59)), rng.uniform(0.1, 10.0, 5)),
                 (sp.csr_matrix(ops.sub_vals.T), rng.uniform(0.1, 10.0, 9))):
        c0, ptr, idx, w = bp.ab_table(ops.robin_vals, W)
        AB = c0 + np.array([(w[ptr[e]:ptr[e + 1]] * x[idx[ptr[e]:ptr[e + 1]]]).sum() for e in range(bp.nAB)])
        sol = bp.replay(AB, ops.F)
        ref = spl.spsolve(ops.csr(ops.robin_vals + sp.csr_matrix(W) @ x).tocsc(), ops.F)
        assert np.linalg.norm(sol - ref) < 1e-12 * np.linalg.norm(ref)
    with pytest.raises(np.linalg.LinAlgError):
        bp.replay(-AB, ops.F)


@pytest.mark.parametrize("m", [4, 12, 20, 24])
def test_band_plan_qoi_only_form_gives_the_same_observables(spaces, m):
    """The QoI-only form of the band sweep (csrc/fom_band.hip, finrom_fom_band_desc::qoi_*): an observation row's weights on a
    fin ride through that fin's forward sweep as its right-hand side and come out as a functional of the fin's interface
    values; no factor, y or backward sweep for the fins.  Replayed in NumPy with the kernel's data flow, the observables must
    equal B_obs w of the full sweep; the 40 point observations of external_obs do not split this way (two on one fin) and get
    no tables."""
    import scipy.sparse as sp
    from bayesianinferencedl_amd.engine import FomEngine
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    V = spaces(m)
    ops = V.operators()
    bp = ops.band_plan()
    fin = Fin(V)
    Fg = np.zeros(bp.G)
    for seg in bp.fin_segs + [bp.post_seg]:
        Fg[seg.g0:seg.g0 + seg.npiv] = ops.F[bp.perm[seg.e0:seg.e0 + seg.npiv]]
    Bp = sp.csr_matrix(np.asarray(fin.B_obs)[:, bp.perm])
    qo = FomEngine.qoi_only_tables(bp, Bp, Fg)
    assert qo is not None and sorted(int(f) for f in qo[1] if f >= 0) == list(range(8))      # eight fin rows + the centre row
    rng = np.random.default_rng(m + 1)
    x = np.exp(0.5 * rng.standard_normal(ops.n))
    c0, ptr, idx, w = bp.ab_table(ops.robin_vals, ops.W_field)
    AB = c0 + np.array([(w[ptr[e]:ptr[e + 1]] * x[idx[ptr[e]:ptr[e + 1]]]).sum() for e in range(bp.nAB)])
    q = bp.replay(AB, ops.F, qoi_only=qo)
    ref = np.asarray(fin.B_obs) @ bp.replay(AB, ops.F)
    assert np.linalg.norm(q - ref) < 1e-12 * np.linalg.norm(ref)
    B40 = sp.csr_matrix(np.asarray(Fin(V, external_obs=True).B_obs)[:, bp.perm])
    if m >= 12:
        assert FomEngine.qoi_only_tables(bp, B40, Fg) is None


def test_sampler_rejects_a_lower_triangular_factor():
    """finrom_sampler_create only accepts the UPPER factor scipy.linalg.cholesky returns (gaussian_field.py:30); a lower factor
    (np.linalg.cholesky) would silently give wrong fields, so it is an argument error (checked before any device call)."""
    import ctypes as C
    from bayesianinferencedl_amd import _ffi
    L = np.linalg.cholesky(np.eye(5) + 0.1 * np.ones((5, 5)))
    h = C.c_void_p()
    rc = _ffi.lib().finrom_sampler_create(np.ascontiguousarray(L).ctypes.data_as(_ffi.c_f64p), 5, C.byref(h))
    assert rc == -1 and b"upper" in _ffi.lib().finrom_last_error()


def test_validators_under_address_sanitizer():
    """SURVEY 5 (race / memory checking of the host layer): the C-ABI's create-time validators run under an AddressSanitizer +
    UBSan build of csrc/finrom_api.hip (host code only) on the CPU box, fed corrupt descriptors by tools/asan_validators.py."""
    from bayesianinferencedl_amd import _build
    lib = _build.build_asan()
    rt = _build.asan_runtime()
    assert rt, "clang's shared ASan runtime not found"
    env = dict(os.environ, LD_PRELOAD=":".join(rt), FINROM_LIB=lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_validators.py")], capture_output=True, text=True,
                       timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ASAN-VALIDATORS-OK" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_function_writes_invalidate_the_cached_observables(spaces):
    """Fin.forward attaches the kernel's B_obs w to the returned Function; qoi_operator may only use it while the
    state is untouched -- set_local / axpy / __setitem__ / assign must drop it (the reference always computes B_obs @ w)."""
    from bayesianinferencedl_amd.fem import Function
    V = spaces(4)
    for mutate in (lambda f: f.vector().set_local(np.ones(V.dim())), lambda f: f.vector().axpy(2.0, np.ones(V.dim())),
                   lambda f: f.vector().__setitem__(3, 7.0), lambda f: f.assign(np.arange(V.dim(), dtype=float))):
        f = Function(V, np.zeros(V.dim()))
        f._qoi = np.arange(9.0)
        mutate(f)
        assert f._qoi is None
    f = Function(V); f._qoi = np.arange(9.0)
    _ = f.vector()[:]; _ = f.vector().get_local()            # reads keep it
    assert f._qoi is not None


def test_band_value_slots_are_shared_only_when_their_records_are_equal(spaces):
    """bandplan.compact_slots: logical value slots with the same affine record share a physical slot (370 instead of 4887 for
    the five fin conductivities at m = 12), the slots the fins write to stay private, and expanding the physical values
    through the map reproduces every logical value exactly."""
    import scipy.sparse as sp
    ops = spaces(12).operators()
    bp = ops.band_plan()
    W = sp.csr_matrix(ops.W_field @ sp.csr_matrix(ops.N9 @ ops.E59))
    c0, ptr, idx, w = bp.ab_table(ops.robin_vals, W)
    abmap, c0p, ptrp, idxp, wp = bp.compact_slots(c0, ptr, idx, w)
    assert len(c0p) < 600 < bp.nAB and abmap.max() == len(c0p) - 1
    x = np.random.default_rng(0).uniform(0.1, 10.0, 5)
    val = lambda c, p_, i_, w_, e: c[e] + (w_[p_[e]:p_[e + 1]] * x[i_[p_[e]:p_[e + 1]]]).sum()
    phys = np.array([val(c0p, ptrp, idxp, wp, e) for e in range(len(c0p))])
    logical = np.array([val(c0, ptr, idx, w, e) for e in range(bp.nAB)])
    assert np.array_equal(phys[abmap], logical)
    targets = [off for tg in bp.schur_target for _, _, off in tg]
    assert len(set(abmap[targets].tolist())) == len(targets)                      # private
    shared = np.bincount(abmap)
    assert all(shared[abmap[t]] == 1 for t in targets)
    Wf = ops.W_field                                                              # a nodal field: hardly any duplicates
    assert len(bp.compact_slots(*bp.ab_table(ops.robin_vals, Wf))[1]) > 0.9 * 3 * bp.G


def test_hessian_action_without_a_band_plan_runs_on_the_host(problems, spaces, monkeypatch):
    """A handle without a band plan (meshes beyond the built-in windows, FINROM_NO_BAND=1) keeps the reference's property that
    `hessian_action` works on any mesh (fom/forward_solve.py:344-368 is a host routine): SciPy SuperLU on the host, announced by
    a RuntimeWarning, no GPU needed.  Same checks as the device path's test below: derivative of the ORACLE's adjoint gradient
    at second order, symmetry."""
    from oracle import fin_oracle as O
    import bayesianinferencedl_amd.engine as E
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    monkeypatch.setattr(E, "USE_BAND", False)
    m = 4
    fin = Fin(spaces(m)); fo = O.FinOracle(problems(m))
    rng = np.random.default_rng(0)
    k = np.exp(0.3 * rng.standard_normal(fin.dofs)); u = rng.standard_normal(fin.dofs); u2 = rng.standard_normal(fin.dofs)
    d = rng.uniform(0.1, 1.0, fin.n_obs)
    with pytest.warns(RuntimeWarning, match="on the host"):
        H = fin.hessian_action(k, u, d)
    err = []
    for eps in (1e-3, 1e-4):
        fd = (fo.gradient(k + eps * u, d) - fo.gradient(k - eps * u, d)) / (2 * eps)
        err.append(np.linalg.norm(H - fd) / np.linalg.norm(fd))
    assert err[0] < 1e-5 and err[1] < 1e-7 and err[1] < err[0] / 50          # O(eps^2)
    with pytest.warns(RuntimeWarning):
        H2 = fin.hessian_action(k, u2, d)
    assert abs(u2 @ H - u @ H2) < 1e-10 * np.linalg.norm(H) * np.linalg.norm(u2)


@pytest.mark.gpu
def test_hessian_action_is_the_derivative_of_the_gradient(problems, spaces):
    """`Fin.hessian_action` (reference fom/forward_solve.py:344-368, here consistent with the k-linear forward model; its four
    solves on the device since round 3): central differences of the ORACLE's adjoint gradient converge to it at second order,
    and the action is symmetric."""
    from oracle import fin_oracle as O
    from bayesianinferencedl_amd.fom.forward_solve import Fin
    m = 4
    fin = Fin(spaces(m)); fo = O.FinOracle(problems(m))
    rng = np.random.default_rng(0)
    k = np.exp(0.3 * rng.standard_normal(fin.dofs)); u = rng.standard_normal(fin.dofs); u2 = rng.standard_normal(fin.dofs)
    d = rng.uniform(0.1, 1.0, fin.n_obs)
    H = fin.hessian_action(k, u, d)
    err = []
    for eps in (1e-3, 1e-4):
        fd = (fo.gradient(k + eps * u, d) - fo.gradient(k - eps * u, d)) / (2 * eps)
        err.append(np.linalg.norm(H - fd) / np.linalg.norm(fd))
    assert err[0] < 1e-5 and err[1] < 1e-7 and err[1] < err[0] / 50          # O(eps^2)
    assert abs(u2 @ H - u @ fin.hessian_action(k, u2, d)) < 1e-10 * np.linalg.norm(H) * np.linalg.norm(u2)


@pytest.mark.parametrize("m,r", [(4, 24), (4, 80), (12, 33)])
def test_grouped_projection_tables_reproduce_psi_t_psi(m, r):
    """Host logic of the grouped projection loop (finrom_rom_create -> RomDev::kmg / tvg / ext_def), no GPU: the
    tables from the host-only entry finrom_rom_grouped_tables, walked in NumPy exactly as proj_main_grouped walks them (slab =
    first term's rows as they are where the record says so, multiply-adds for the others, accumulators rescaled where a record
    opens a group and behind the last k-step), must give psi^T psi with psi = (A_robin + sum_i theta_i A_i) Phi -- what the
    reference forms (rom/averaged_affine_ROM.py:291-297) -- for conductivities over the dataset's two decades."""
    import ctypes as C

    from bayesianinferencedl_amd import _ffi
    from bayesianinferencedl_amd.engine import RomEngine
    from bayesianinferencedl_amd.fom.thermal_fin import get_space
    V = get_space(None, m=m); ops = V.operators(); n = V.dim()
    rng = np.random.default_rng(3)
    phi = np.linalg.qr(rng.standard_normal((n, r)))[0]
    tables = [ops.csr(ops.robin_vals) @ phi] + [ops.csr(ops.sub_vals[i]) @ phi for i in range(9)]
    row_ptr, term_p, tv = RomEngine.pack_terms(n, r, list(enumerate(tables)))
    a1, p1 = _ffi.i32(row_ptr); a2, p2 = _ffi.i32(term_p); a3, p3 = _ffi.f64(tv)
    d = _ffi.RomDesc(n=n, r=r, P=9, n_obs=0, nterms=len(term_p), row_ptr=p1, term_p=p2, term_val=p3, rhs=None, obs_phi=None)
    L = _ffi.lib()
    nkg, n_ext, ext_final, n_slots = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    _ffi.check(L.finrom_rom_grouped_tables(C.byref(d), C.byref(nkg), C.byref(n_ext), C.byref(ext_final), C.byref(n_slots), None, None, None))
    nkg, n_ext, ext_final, n_slots = nkg.value, n_ext.value, ext_final.value, n_slots.value
    rp = (r + 15) // 16 * 16
    assert nkg > 0 and nkg % 3 == 0 and 0 < n_ext <= 64 and 0 < ext_final < n_ext
    kmg = np.zeros((nkg + 8) * 8, np.int32); tvg = np.zeros(n_slots * 4 * rp); ext_def = np.zeros(n_ext * 3, np.int32)
    o = [C.c_int32() for _ in range(3)] + [C.c_int64()]
    _ffi.check(L.finrom_rom_grouped_tables(C.byref(d), *[C.byref(x) for x in o], kmg.ctypes.data_as(_ffi.c_i32p),
                                           tvg.ctypes.data_as(_ffi.c_f64p), ext_def.ctypes.data_as(_ffi.c_i32p)))
    kmg = kmg.reshape(-1, 8); tvg = tvg.reshape(n_slots, 4, rp); ext_def = ext_def.reshape(n_ext, 3)
    assert (kmg[nkg:, 1] == 1).all() and (kmg[nkg:, 2] == 1).all()          # the records behind the list: zero k-steps
    assert not tvg[kmg[nkg, 0]].any()
    unit = (kmg[:nkg, 2] & 1) != 0
    assert unit.mean() > 0.9                                # almost every k-step's first term goes in as loaded
    if m == 12:                                             # (the survey's mesh: 62 % of the rows lie inside a sub-domain)
        assert ((kmg[:nkg, 1] == 1) & unit).mean() > 0.5  # ... and more than half of the k-steps need no arithmetic at all
    assert ((kmg[:nkg, 2] & 2) != 0).sum() <= 10          # one rescaling per sub-domain at most
    for trial in range(3):
        theta = np.exp(rng.uniform(np.log(0.1), np.log(10.0), 9))
        th1 = np.concatenate([[1.0], theta])
        ext = np.array([(th1[a] / th1[b]) ** (2 if sq else 1) for a, b, sq in ext_def])
        assert ext[0] == 1.0
        acc = np.zeros((rp, rp))
        for slot, nt, flags, fidx, *cf in kmg[:nkg]:
            if flags & 2:
                acc *= ext[fidx]
            slab = tvg[slot].copy() if flags & 1 else ext[cf[0]] * tvg[slot]
            for t in range(1, nt):
                slab += ext[cf[t]] * tvg[slot + t]
            acc += slab.T @ slab
        acc *= ext[ext_final]
        psi = sum(th1[p] * tables[p] for p in range(10))
        want = psi.T @ psi
        assert np.max(np.abs(acc[:r, :r] - want)) <= 1e-12 * np.abs(want).max()
        assert not acc[r:].any() and not acc[:, r:].any()
