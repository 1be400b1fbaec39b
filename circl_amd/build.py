"""Builds circl_amd/libcirclhip.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m circl_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
tree.  Intermediate files go to build/ (also git-ignored).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcirclhip.so")
SOURCES = ["circl_hip.hip"]
HEADERS = ["keccak_dev.h", "kyber_dev.h", "dilithium_dev.h", "mlkem_kernels.h", "mldsa_kernels.h", "mldsa_sign_batched.h", "prim_kernels.h"]
ARCH = "gfx950"


def _deps():
    d = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    d.append(os.path.join(ROOT, "include", "circl_hip.h"))
    return [p for p in d if os.path.exists(p)]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    bdir = os.path.join(ROOT, "build")
    os.makedirs(bdir, exist_ok=True)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall",
           "-Wno-unused-function", "-Wno-unused-variable",
           "-save-temps=obj", "-o", os.path.join(bdir, "libcirclhip.so")]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=bdir)
    shutil.copy2(os.path.join(bdir, "libcirclhip.so"), LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
