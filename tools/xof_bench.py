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
big = [rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes() for _ in range(64)]
dt = t(lambda: hostapi.k12(big, 32))
print(f"KangarooTwelve of 64 x 1 MiB: {dt * 1e3:.1f} ms -> {64 * (1 << 20) / dt / 1e9:.2f} GB/s (incl. host-side tree layout)")
one = [rng.integers(0, 256, 24 << 20, dtype=np.uint8).tobytes()]
dt = t(lambda: hostapi.k12(one, 32))
print(f"KangarooTwelve of 1 x 24 MiB: {dt * 1e3:.1f} ms -> {(24 << 20) / dt / 1e9:.2f} GB/s")
