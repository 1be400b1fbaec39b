"""The host runtime under ThreadSanitizer and AddressSanitizer (`make tsan`, `make asan`): tests/race_driver.cpp drives
the library from several threads over several (logical) devices, with the default pool sizes, with one staging slot per
device, and with no mover threads.  Zero reports is the bar -- the counterpart of the reference's `go test -race`
(cloudflare/circl Makefile:44-47)."""
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "_san")


def sources_hash():
    """What the instrumented binaries are built from: a prebuilt pair in tests/_san (git-ignored, but it travels to the GPU box) is only
    trusted while its stamp carries this hash -- a stale instrumented library would test last week's runtime."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "circl_amd", "csrc", "*"))) + [os.path.join(ROOT, "include", "circl_hip.h"),
                                                                                 os.path.join(ROOT, "tests", "race_driver.cpp"),
                                                                                 os.path.join(ROOT, "tests", "san_shims.c"), os.path.join(ROOT, "Makefile")]
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()


def build(kind):
    exe = os.path.join(SAN, "race_driver_" + kind)
    stamp = os.path.join(SAN, kind + ".stamp")
    want = sources_hash()
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if not os.path.exists(exe) or have != want:  # normally prebuilt (and stamped) in the tree that travels to the GPU box
        subprocess.check_call(["make", "-B", "-j8", kind], cwd=ROOT)
        with open(stamp, "w") as f:
            f.write(want)
    return exe


def run(kind, env_extra, args=()):
    exe = build(kind)
    env = dict(os.environ, CIRCL_HIP_LOGICAL_DEVICES="4")
    if kind == "tsan":
        env["TSAN_OPTIONS"] = "suppressions=%s halt_on_error=0 second_deadlock_stack=1 history_size=4 exitcode=66" % os.path.join(ROOT, "tests", "tsan.supp")
    else:
        # protect_shadow_gap=0: the ROCm runtime maps memory where ASan keeps its shadow gap; leaks: the library's pools are
        # process-lifetime by design (worker threads may outlive static destruction)
        env["ASAN_OPTIONS"] = "protect_shadow_gap=0:detect_leaks=0:abort_on_error=0:exitcode=67"
    env.update(env_extra)
    cmd = [exe]
    if kind == "tsan" and shutil.which("setarch"):
        # gcc 11's libtsan predates kernels with 32 bits of mmap randomisation ("FATAL: ThreadSanitizer: unexpected memory
        # mapping"): run the driver with address-space randomisation off
        probe = subprocess.run(["setarch", os.uname().machine, "-R", "true"], capture_output=True)
        if probe.returncode == 0:
            cmd = ["setarch", os.uname().machine, "-R", exe]
    r = subprocess.run(cmd + [str(a) for a in args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    log = r.stdout[-3000:] + "\n" + r.stderr[-12000:]
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "san_%s_%s.log" % (kind, "_".join("%s%s" % (k[-5:], v) for k, v in sorted(env_extra.items())) or "default")), "w") as f:
        f.write(r.stdout + "\n" + r.stderr)
    reports = re.findall(r"WARNING: ThreadSanitizer: [^\n]*|ERROR: AddressSanitizer: [^\n]*", r.stderr)
    assert not reports, log
    assert r.returncode == 0 and "race_driver ok" in r.stdout, log


@pytest.mark.parametrize("env_extra", [{}, {"CIRCL_HIP_HOST_SLOTS": "1"}, {"CIRCL_HIP_HOST_THREADS": "0", "CIRCL_HIP_HOST_SLOTS": "2"}],
                         ids=["default", "single-slot", "no-workers"])
def test_host_runtime_under_thread_sanitizer(env_extra):
    run("tsan", env_extra, (3, 2, 6000, 200))


def test_host_runtime_under_address_sanitizer():
    run("asan", {}, (3, 2, 12000, 300))
