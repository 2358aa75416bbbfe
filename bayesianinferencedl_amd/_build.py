"""Build recipe for libfinrom_hip.so (hipcc, gfx950 only, in-tree output)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfinrom_hip.so")
SOURCES = ["finrom_api.hip", "fom_kernels.hip", "fom_band.hip", "fom_band_wide.hip", "fom_band_adjoint.hip", "rom_kernels.hip", "rom_proj_wide.hip", "rom_proj_half.hip", "rom_proj_single.hip", "rom_onesample.hip", "rom_gram.hip", "util_kernels.hip", "mlp_kernels.hip", "hmc_kernels.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("finrom_core.h", "finrom_internal.h", "rom_proj_device.h", "fom_band_device.h", "mlp_device.h")] + [os.path.join(ROOT, "include", "finrom.h")]


def _deps(path, seen=None):
    """The source and every project header / source it includes with #include "...", recursively (csrc/ and include/): a change of
    rom_proj_device.h does not rebuild the band sweep's translation units (minutes each)."""
    import re
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path) as f:
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', f.read(), flags=re.M):
            for d in (os.path.dirname(path), CSRC, os.path.join(ROOT, "include")):
                q = os.path.join(d, inc)
                if os.path.exists(q):
                    _deps(q, seen)
                    break
    return seen
FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC] + \
    os.environ.get("FINROM_EXTRA_FLAGS", "").split()      # (A/B builds of tuning constants, e.g. -DADJ_RING=2)
# rom_proj_single.hip is built at -O2: at -O3 hipcc's extra passes inflate the register pressure of the r = 80 projection kernel
# (chol_tiles + solve_tiles around inline-asm MFMA tuples) from 188 VGPRs to 256 + 388 B of scratch, and the kernel
# must stay at <= 192 to share SIMDs with the FOM interpreter.  The MFMA main loop is inline asm either way.
OPT = {"rom_proj_single.hip": "-O2"}
EXTRA_DEPS = {"fom_band_wide.hip": ["fom_band.hip"]}       # (includes it: same templates, other instantiations)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, sorted(_deps(sp)) + [os.path.join(CSRC, d) for d in EXTRA_DEPS.get(src, [])]):
            jobs.append([hipcc, *FLAGS, OPT.get(src, "-O3"), "-c", sp, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))


def build_asan() -> str:
    """Host-side AddressSanitizer + UBSan build of the C-ABI layer (csrc/finrom_api.hip: the create-time validators of every
    descriptor) for the CPU box: only the HOST code of finrom_api.hip is instrumented (-Xarch_host), the kernels' objects are
    the product's.  GPU sanitizers are not available on this pool; tools/asan_validators.py drives the corrupt-descriptor
    cases through this library without a GPU (a corrupt descriptor is rejected before any device call)."""
    build()
    hipcc = _hipcc()
    obj = os.path.join(LIBDIR, "finrom_api_asan.o")
    lib = os.path.join(LIBDIR, "libfinrom_hip_asan.so")
    san = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer"]
    src = os.path.join(CSRC, "finrom_api.hip")
    if _stale(obj, sorted(_deps(src))):
        r = subprocess.run([hipcc, *FLAGS, "-O1", "-g", *san, "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc (asan) failed:\n" + r.stdout + r.stderr)
    others = [os.path.join(LIBDIR, s_.replace(".hip", ".o")) for s_ in SOURCES if s_ != "finrom_api.hip"]
    if _stale(lib, [obj] + others):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *san[:2], "-o", lib, obj, *others], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc (asan link) failed:\n" + r.stdout + r.stderr)
    return lib


def asan_runtime():
    """clang's shared ASan runtime to LD_PRELOAD into the Python process that loads the ASan build.  It carries the UBSan
    handlers too; preloading libclang_rt.ubsan_standalone beside it hangs the process at start-up."""
    import glob
    base = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(_hipcc()))), "lib", "llvm", "lib", "clang")
    out = []
    for pat in ("libclang_rt.asan-x86_64.so",):
        hits = sorted(glob.glob(os.path.join(base, "*", "lib", "linux", pat)))
        if hits:
            out.append(hits[-1])
    return out
