"""Builds and runs tests/host_mirror_test.cpp: the C++ host layer (include/circl/{kem,sign}.hpp) that
mirrors the reference's kem.Scheme / sign.Scheme above the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    out = os.path.join(ROOT, "build", "host_mirror_test")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_mirror_test.cpp"),
                           "-L", os.path.join(ROOT, "circl_amd"), "-lcirclhip", "-Wl,-rpath," + os.path.join(ROOT, "circl_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


def _build_xwing():
    out = os.path.join(ROOT, "build", "xwing_test")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "xwing_test.cpp"),
                           "-L", os.path.join(ROOT, "circl_amd"), "-lcirclhip", "-lcrypto", "-Wl,-rpath," + os.path.join(ROOT, "circl_amd"),
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
