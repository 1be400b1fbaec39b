"""Builds circl_amd/libcirclhip.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m circl_amd.build [--force]

hipcc cross-compiles without a GPU.  The translation units are compiled in parallel (one hipcc per .hip file) and linked
into one shared object.  The .so is git-ignored but travels to the GPU box with the tree.  Intermediate files (objects,
the gfx950 assembly of every kernel) go to build/ (also git-ignored).
"""
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcirclhip.so")
UNITS = ["host_runtime.hip", "host_coalesce.hip", "api_mlkem.hip", "api_mldsa.hip", "api_prims.hip", "api_x25519.hip", "api_hybrid.hip"]
_INCLUDE = re.compile(r'^\s*#\s*include\s*"([^"]+)"', re.M)


def deps(unit):
    """Every file `unit` depends on: the transitive closure of its #include "..." graph over csrc/ and include/ (read from the
    sources at every call, so a new header can never be forgotten in a table)."""
    seen, todo = set(), [os.path.join(CSRC, unit)]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(f):
            continue
        seen.add(f)
        with open(f, errors="replace") as fh:
            for inc in _INCLUDE.findall(fh.read()):
                for base in (os.path.dirname(f), CSRC, os.path.join(ROOT, "include")):
                    cand = os.path.normpath(os.path.join(base, inc))
                    if os.path.exists(cand):
                        todo.append(cand)
                        break
    return sorted(seen)


ARCH = "gfx950"
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def _bdir():
    d = os.path.join(ROOT, "build")
    os.makedirs(d, exist_ok=True)
    return d


def _obj(unit):
    return os.path.join(_bdir(), unit.replace(".hip", ".o"))


def _stale(unit):
    """True when `unit` must be recompiled.  The object's time stamp decides; where build/ did not travel with the tree (a GPU box
    gets the sources and the built .so, not the objects) the library's own time stamp stands in for it, so an up-to-date
    library is never rebuilt -- and never replaced under a process that has it loaded."""
    o = _obj(unit)
    if os.path.exists(o):
        t = os.path.getmtime(o)
    elif os.path.exists(LIB):
        t = os.path.getmtime(LIB)
    else:
        return True
    return any(os.path.getmtime(p) > t for p in deps(unit))


def needs_build():
    return (not os.path.exists(LIB) or any(_stale(u) for u in UNITS) or
            any(os.path.exists(_obj(u)) and os.path.getmtime(_obj(u)) > os.path.getmtime(LIB) for u in UNITS))


def _compile(unit, verbose):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-save-temps=obj", "-c", os.path.join(CSRC, unit), "-o", _obj(unit)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=_bdir())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    todo = [u for u in UNITS if force or _stale(u) or not os.path.exists(_obj(u))]
    with ThreadPoolExecutor(max_workers=max(1, len(todo))) as ex:
        list(ex.map(lambda u: _compile(u, verbose), todo))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = os.path.join(_bdir(), "libcirclhip.so")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + [_obj(u) for u in UNITS] + ["-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=_bdir())
    # a NEW file renamed over the old one: a process that has the old library mapped keeps its (unlinked) file -- writing into
    # the mapped file in place would change the code under it
    tmp = LIB + ".new.%d" % os.getpid()
    shutil.copy2(out, tmp)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
