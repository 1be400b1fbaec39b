"""Builds circl_amd/libcirclhip.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m circl_amd.build [--force]

hipcc cross-compiles without a GPU.  The translation units are compiled in parallel (one hipcc per .hip file) and linked
into one shared object.  The .so is git-ignored but travels to the GPU box with the tree.  Intermediate files (objects,
the gfx950 assembly of every kernel) go to build/ (also git-ignored).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcirclhip.so")
COMMON = ["host_common.h", "keccak_dev.h"]
# translation unit -> the headers it includes (besides COMMON and include/circl_hip.h)
UNITS = {
    "host_runtime.hip": [],
    "api_mlkem.hip": ["kyber_dev.h", "mlkem_kernels.h"],
    "api_mldsa.hip": ["kyber_dev.h", "dilithium_dev.h", "mlkem_kernels.h", "mldsa_kernels.h", "mldsa_sign_batched.h"],
    "api_prims.hip": ["kyber_dev.h", "dilithium_dev.h", "prim_kernels.h", "sampler_prims.h", "mlkem_kernels.h", "mldsa_kernels.h"],
    "api_x25519.hip": ["x25519_dev.h", "x25519_base_table.h", "x25519_kernels.h"],
    "api_hybrid.hip": ["hybrid_kernels.h"],
}
ARCH = "gfx950"
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def _bdir():
    d = os.path.join(ROOT, "build")
    os.makedirs(d, exist_ok=True)
    return d


def _obj(unit):
    return os.path.join(_bdir(), unit.replace(".hip", ".o"))


def _stale(unit):
    o = _obj(unit)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    deps = [os.path.join(CSRC, unit)] + [os.path.join(CSRC, h) for h in COMMON + UNITS[unit]] + [os.path.join(ROOT, "include", "circl_hip.h")]
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in deps)


def needs_build():
    return not os.path.exists(LIB) or any(_stale(u) for u in UNITS) or any(os.path.getmtime(_obj(u)) > os.path.getmtime(LIB) for u in UNITS)


def _compile(unit, verbose):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-save-temps=obj", "-c", os.path.join(CSRC, unit), "-o", _obj(unit)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=_bdir())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    todo = [u for u in UNITS if force or _stale(u)]
    with ThreadPoolExecutor(max_workers=max(1, len(todo))) as ex:
        list(ex.map(lambda u: _compile(u, verbose), todo))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = os.path.join(_bdir(), "libcirclhip.so")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + [_obj(u) for u in UNITS] + ["-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=_bdir())
    shutil.copy2(out, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
