"""Not a test: throughput of the batched XOF / KangarooTwelve service through the host-buffer ABI.

    python tools/xof_bench.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import hostapi  # noqa: E402

rng = np.random.default_rng(1)


def t(fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


n, ln = 1 << 16, 8192
msgs = rng.integers(0, 256, (n, ln), dtype=np.uint8)
dt = t(lambda: hostapi.shake(168, 0x1f, msgs, 32))
print(f"SHAKE128 of {n} x {ln} B (equal lengths): {dt * 1e3:.1f} ms -> {n * ln / dt / 1e9:.1f} GB/s hashed (host ABI, PCIe-inclusive)")
rag = [rng.integers(0, 256, int(rng.integers(1, 4000)), dtype=np.uint8).tobytes() for _ in range(1 << 14)]
tot = sum(len(x) for x in rag)
dt = t(lambda: hostapi.xof(168, 0x1f, rag, 32))
print(f"SHAKE128 of {len(rag)} ragged messages ({tot / 1e6:.1f} MB, odd offsets): {dt * 1e3:.1f} ms -> {tot / dt / 1e9:.2f} GB/s (incl. Python blob building)")
import ctypes as C  # noqa: E402
from circl_amd import _native as nat  # noqa: E402
L = nat.lib()


def k12_raw(msgs, label):
    """the C entry point on a prebuilt blob (what a Go caller's []byte is): no Python in the timed region"""
    mb, mo = hostapi._blob(msgs)
    out = np.empty((len(msgs), 32), np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    dt = t(lambda: L.circl_hip_k12(p(mb), p(mo), None, None, p(out), 32, len(msgs), 0))
    tot = sum(len(m) for m in msgs)
    print(f"KangarooTwelve of {label}: {dt * 1e3:.1f} ms -> {tot / dt / 1e9:.2f} GB/s (circl_hip_k12 on a prebuilt blob, pageable, PCIe-inclusive)")
    return out


big = [rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes() for _ in range(64)]
o1 = k12_raw(big, "64 x 1 MiB")
assert (o1 == hostapi.k12(big, 32)).all()
k12_raw([rng.integers(0, 256, 24 << 20, dtype=np.uint8).tobytes()], "1 x 24 MiB")
k12_raw([rng.integers(0, 256, 256 << 20, dtype=np.uint8).tobytes()], "1 x 256 MiB")
k12_raw([rng.integers(0, 256, 4096, dtype=np.uint8).tobytes() for _ in range(1 << 15)], "32768 x 4 KiB (short messages)")
