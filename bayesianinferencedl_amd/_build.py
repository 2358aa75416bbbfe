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
SOURCES = ["finrom_api.hip", "fom_kernels.hip", "fom_band.hip", "rom_kernels.hip", "rom_proj_single.hip", "rom_gram.hip", "util_kernels.hip"]
HEADERS = [os.path.join(CSRC, "finrom_internal.h"), os.path.join(CSRC, "rom_proj_device.h"), os.path.join(ROOT, "include", "finrom.h")]
FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC]
# rom_proj_single.hip is built at -O2: at -O3 hipcc's extra passes inflate the register pressure of the r = 80 projection kernel
# (chol_tiles + solve_tiles around inline-asm MFMA tuples) from 188 VGPRs to 256 + 388 B of scratch, and the kernel
# must stay at <= 192 to share SIMDs with the FOM interpreter.  The MFMA main loop is inline asm either way.
OPT = {"rom_proj_single.hip": "-O2"}


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
        if force or _stale(obj, [sp] + HEADERS):
            jobs.append([hipcc, *FLAGS, OPT.get(src, "-O3"), "-c", sp, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
