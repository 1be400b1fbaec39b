"""Builds and runs tests/host_mirror_test.cpp: the C++ host layer (include/circl/{kem,sign,serving}.hpp) that
mirrors the reference's kem.Scheme / sign.Scheme above the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    out = os.path.join(ROOT, "build", "host_mirror_test")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_mirror_test.cpp"),
                           "-L", os.path.join(ROOT, "circl_amd"), "-lcirclhip", "-Wl,-rpath," + os.path.join(ROOT, "circl_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


def _build_xwing():
    out = os.path.join(ROOT, "build", "xwing_test")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "xwing_test.cpp"),
                           "-L", os.path.join(ROOT, "circl_amd"), "-lcirclhip", "-Wl,-rpath," + os.path.join(ROOT, "circl_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


def test_xwing_layer_compiles():
    from circl_amd import build as cbuild
    cbuild.build()
    _build_xwing()


@pytest.mark.gpu
def test_xwing_reference_test_vectors():
    # kem/xwing/xwing_test.go:38-85: transcript digest of the X-Wing draft's test vectors
    exe = _build_xwing()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_host_mirror_compiles():
    from circl_amd import build as cbuild
    cbuild.build()
    _build()


@pytest.mark.gpu
def test_host_mirror_runs_like_schemes_test():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def _build_hybrid():
    out = os.path.join(ROOT, "build", "hybrid_test")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "hybrid_test.cpp"),
                           "-L", os.path.join(ROOT, "circl_amd"), "-lcirclhip", "-Wl,-rpath," + os.path.join(ROOT, "circl_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


def test_x25519_checker_rfc7748():
    # RFC 7748 section 5.2 test vector 1 and the section 6.1 Diffie-Hellman example pin the pure-Python checker
    import x25519 as X
    k = bytes.fromhex("a546e36bf0527c9d3b16154b82465edd62144c0ac1fc5a18506a2244ba449ac4")
    u = bytes.fromhex("e6db6867583030db3594c1a424b15f7c726624ec26b3353b10a903a6d0ab1c4c")
    assert X.x25519(k, u).hex() == "c3da55379de9c6908e94ea4df28d084f32eccf03491c71f754b4075577a28552"
    a = bytes.fromhex("77076d0a7318a57d3c16c17251b26645df4c2f87ebc0992ab177fba51db92c2a")
    b = bytes.fromhex("5dab087e624a8a4b79e17f8b83800ee66f3bb1292618b6fd1c2f8b27ff88e0eb")
    assert X.public(a).hex() == "8520f0098930a754748b7ddcb43ef75a0dbf3a0d26381af4eba4a98eaa9b4e6a"
    assert X.x25519(a, X.public(b)).hex() == "4a5d9d5ba4ce2de1728e3bf480350f25e07e21c947d19e3376f09b3c1e161742"


def test_hybrid_layer_compiles():
    from circl_amd import build as cbuild
    cbuild.build()
    _build_hybrid()


@pytest.mark.gpu
def test_hybrid_x25519mlkem768_like_xkem_test():
    # kem/hybrid/xkem_test.go:35-69 (low-order points) + kem/schemes/schemes_test.go:60-133 (sizes, round trip)
    exe = _build_hybrid()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


@pytest.mark.gpu
def test_hybrid_x25519mlkem768_against_oracle():
    """kem/hybrid/hybrid.go:236-300 restated with the oracle's SHAKE256 / ML-KEM-768 and the pure-Python X25519:
    SHAKE256(seed) -> mlkem seed[64] || x seed[32]; x sk = SHAKE256(x seed)[:32]; pk = ek || X25519(sk, 9); ..."""
    import numpy as np
    from oracle import orc
    import x25519 as X
    exe = _build_hybrid()
    r = subprocess.run([exe, "dump", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rec = [dict()]
    for line in r.stdout.split("\n"):
        if not line.strip():
            continue
        tag, val = line.split()
        if tag in rec[-1]:
            rec.append(dict())
        rec[-1][tag] = bytes.fromhex(val)
    assert len(rec) == 4

    def shake256(data, outlen):
        return orc.sponge(data, outlen, 136, 0x1f)

    for t in rec:
        ex = shake256(t["seed"], 96)
        ek, dk = orc.mlkem_keygen(768, np.frombuffer(ex[:64], dtype=np.uint8)[None, :])
        skx = shake256(ex[64:96], 32)
        assert t["pk"] == bytes(ek[0]) + X.public(skx)
        assert t["sk"] == bytes(dk[0]) + skx
        ee = shake256(t["eseed"], 64)
        ct, ss, st = orc.mlkem_encaps(768, ek, np.frombuffer(ee[:32], dtype=np.uint8)[None, :])
        esk = shake256(ee[32:64], 32)
        assert int(st[0]) == 0
        assert t["ct"] == bytes(ct[0]) + X.public(esk)
        assert t["ss"] == bytes(ss[0]) + X.x25519(esk, t["pk"][1184:])


def _build_cgo_shape():
    out = os.path.join(ROOT, "build", "cgo_shape_test")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["gcc", "-O1", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cgo_shape_test.c"),
                           "-L", os.path.join(ROOT, "circl_amd"), "-lcirclhip", "-Wl,-rpath," + os.path.join(ROOT, "circl_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


def test_cgo_shape_harness_compiles():
    from circl_amd import build as cbuild
    cbuild.build()
    _build_cgo_shape()


@pytest.mark.gpu
def test_cgo_shaped_calls():
    # the C ABI called the way cgo's stubs for go/*/hipbatch call it: byte-aligned sub-slices, NULL for empty slices,
    # status == NULL, appended blobs, key tables, three OS threads at once
    exe = _build_cgo_shape()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
