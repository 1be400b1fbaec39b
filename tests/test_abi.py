"""CPU-side checks of the drop-in boundary: libcirclhip.so builds, loads and exports every symbol
include/circl_hip.h declares; sizes match the reference's scheme constants; without a GPU every
compute entry point fails loudly (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from circl_amd import _native as nat
from circl_amd import build as cbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    cbuild.build()
    return nat.lib()


def test_header_symbols_all_exported(L):
    hdr = open(os.path.join(ROOT, "include", "circl_hip.h")).read()
    declared = set(re.findall(r"\b(circl_hip_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(nat.SYMBOLS), declared ^ set(nat.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s


def test_sizes_match_reference_constants(L):
    # kem/mlkem/mlkem{512,768,1024}/kyber.go:18-36 ; sign/mldsa/mldsa{44,65,87}/internal/dilithium.go:31-38
    for p, (ek, dk, ct) in {512: (800, 1632, 768), 768: (1184, 2400, 1088), 1024: (1568, 3168, 1568)}.items():
        assert (L.circl_hip_mlkem_ek_size(p), L.circl_hip_mlkem_dk_size(p), L.circl_hip_mlkem_ct_size(p)) == (ek, dk, ct)
    for p, (pk, sig) in {44: (1312, 2420), 65: (1952, 3309), 87: (2592, 4627)}.items():
        assert (L.circl_hip_mldsa_pk_size(p), L.circl_hip_mldsa_sig_size(p)) == (pk, sig)
    assert L.circl_hip_mlkem_ek_size(999) == 0


def test_product_does_not_reference_oracle():
    # the oracle is test infrastructure: nothing in the shipped package may import, link or call it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "circl_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liborc" not in src and "orc_" not in src and not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_no_gpu_means_loud_failure(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert L.circl_hip_init() == nat.ENODEV
    from circl_amd import hostapi
    with pytest.raises(nat.CirclHipError):
        hostapi.mlkem_encaps(768, np.zeros((1, 1184), np.uint8), np.zeros((1, 32), np.uint8))
    with pytest.raises(nat.CirclHipError):
        hostapi.keccak_f1600(np.zeros((1, 25), np.uint64))
    # key tables that live across calls: no device, no table (and nothing to free); a NULL table is refused, not dereferenced
    for kind, param, row in (("mlkem-public", 768, 1184), ("mlkem-private", 768, 2400), ("mldsa-public", 65, 1952), ("mldsa-private", 65, 4032)):
        with pytest.raises(nat.CirclHipError) as e:
            hostapi.KeyTable(kind, param, np.zeros((1, row), np.uint8))
        assert e.value.code == nat.ENODEV
    out = np.zeros(4000, np.uint8)
    import ctypes as C
    assert L.circl_hip_mlkem_encaps_table(None, None, out.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                          out.ctypes.data_as(C.c_void_p), 1) == nat.EPARAM
    L.circl_hip_keytable_free(None)


def test_build_dependencies_follow_the_include_graph(tmp_path):
    # circl_amd/build.py must rebuild after an edit to ANY header a unit includes (a stale .so would travel to the GPU box):
    # the dependency set is the #include closure, so keytable.h / lane_ops.h are in it, and a newer header makes the unit stale
    import os
    import time
    from circl_amd import build
    names = {u: {os.path.basename(d) for d in build.deps(u)} for u in build.UNITS}
    assert {"keytable.h", "host_common.h", "circl_hip.h"} <= names["host_runtime.hip"]
    assert "keytable.h" in names["api_mlkem.hip"] and "keytable.h" in names["api_mldsa.hip"]
    assert {"lane_ops.h", "prim_kernels.h", "sampler_prims.h"} <= names["api_prims.hip"]
    assert {"mldsa_sign_batched.h", "dilithium_dev.h", "keccak_dev.h"} <= names["api_mldsa.hip"]
    every = set().union(*names.values())
    assert {f for f in os.listdir(build.CSRC) if f.endswith(".h")} <= every    # no header outside the graph
    kt = os.path.join(build.CSRC, "keytable.h")
    obj = build._obj("host_runtime.hip")
    if os.path.exists(obj):
        st = os.stat(kt)
        try:
            os.utime(kt, (time.time() + 5, time.time() + 5))
            assert build._stale("host_runtime.hip") and build.needs_build()
            assert not build._stale("api_x25519.hip") or not os.path.exists(build._obj("api_x25519.hip"))
        finally:
            os.utime(kt, (st.st_atime, st.st_mtime))


def test_every_environment_knob_is_documented():
    """INTEGRATION.md lists exactly the environment knobs the library reads (they are part of the de-facto ABI)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = set()
    for p in glob.glob(os.path.join(root, "circl_amd", "csrc", "*")):
        code |= set(re.findall(r'"(CIRCL_HIP_[A-Z0-9_]+)"', open(p).read()))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    for k in sorted(code):
        tail = k[len("CIRCL_HIP"):]  # the list abbreviates neighbours: `CIRCL_HIP_KEM_CHAIN` / `_KEM_CHAIN_ENCAPS`
        assert k in doc or ("`%s`" % tail) in doc or ("`_%s`" % tail.split("_", 2)[-1]) in doc, k
    m = re.search(r"Environment knobs\*\* — exactly these (\d+)", doc)
    assert m and int(m.group(1)) == len(code), (m and m.group(1), len(code))


def test_no_wait_lds_ordering_point_is_a_compiler_fence_and_emits_nothing(tmp_path):
    """wave_lds_order() (keccak_dev.h) is what every "no-wait" exchange of a wave-private LDS buffer hangs on: the hardware needs
    nothing (a wavefront's LDS instructions execute in order), the COMPILER must keep each exchange's stores in front of its loads.
    Compiled for gfx950 here (hipcc cross-compiles): the disassembly of four store -> order -> load -> order rounds must show the
    LDS writes and reads strictly alternating, and no s_barrier (the point of the no-wait form)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "order_probe.hip"
    src.write_text(r'''
#include "keccak_dev.h"
__global__ void order_probe(uint32_t *out, const uint32_t *in) {
    __shared__ uint32_t xch[256];
    const int lane = threadIdx.x;
    uint32_t v = in[lane];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        xch[lane] = v;                       // every lane stores its own word ...
        circl::wave_lds_order();
        v = xch[(lane * 5 + r) & 63] + r;    // ... and loads ANOTHER lane's: nothing but the ordering point says the store comes first
        circl::wave_lds_order();
    }
    out[lane] = v;
}
''')
    asm = tmp_path / "order_probe.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-I", os.path.join(root, "circl_amd", "csrc"),
                           str(src), "-o", str(asm)], stderr=subprocess.DEVNULL)
    body = asm.read_text().split("order_probe", 1)[1].split(".amdhsa_kernel", 1)[0]
    lds = re.findall(r"^\s*(ds_write_b32|ds_read_b32|ds_store_b32|ds_load_b32)\b", body, re.M)
    kinds = ["w" if ("write" in x or "store" in x) else "r" for x in lds]
    assert kinds == ["w", "r"] * 4, kinds
    assert "s_barrier" not in body


def test_route_check_knows_the_librarys_defaults():
    """tools/route_check.py judges the DEFAULT thresholds: the defaults it assumes must be the ones compiled into the library."""
    import importlib.util
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("route_check", os.path.join(root, "tools", "route_check.py"))
    rc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rc)
    src = "".join(open(p).read() for p in (os.path.join(root, "circl_amd", "csrc", f) for f in ("api_mlkem.hip", "api_mldsa.hip", "host_runtime.hip")))
    for case, (_what, knob, dflt, _a, _b) in rc.CASES.items():
        m = re.search(r'env_int\("%s", (-?\d+),' % knob, src)
        assert m, knob
        compiled = int(m.group(1))
        if knob == "CIRCL_HIP_DSA_CHAIN_ITEM":  # (-1 = by parameter set: 9 up to K = 6, 8 beyond -- api_mldsa.hip dsa_chain_batch)
            assert compiled == -1 and re.search(r"k <= 6 \? 9 : 8", src) and dflt == (8 if case.endswith("87") else 9)
        else:
            assert compiled == dflt, (knob, compiled, dflt)
