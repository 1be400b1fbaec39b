"""Not a test: what the HOST side of an 8-GPU node costs, measured on whatever box this runs on.

    CIRCL_HIP_LOGICAL_DEVICES=8 python tools/logical8.py [log2 n = 23]

circl_hip_mlkem_encaps(device = -1) on n pageable items: with 8 logical devices the call runs the 8-shard host code of an 8-GPU node
(8 shard threads, 8 staging pools, 8 mover pools) -- on a one-GPU box all of it drives the one GPU, so the RATE is that GPU's PCIe
rate, but the host CPU time per item, the mover threads and the page-locked pools are those of the node.  Prints CPU seconds per
10^6 items (user + system, all threads), the CPUs kept busy, the pools' high-water mark, and what 8 x the per-GPU host rate would need."""
import ctypes as C
import os
import resource
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import _native as nat  # noqa: E402

L = nat.lib()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 23
n = 1 << lg
nd = L.circl_hip_device_count()
EK, CT = 1184, 1088
rng = np.random.default_rng(3)
pool = 1 << 12
seeds = rng.integers(0, 256, (pool, 64), dtype=np.uint8)
ekp, dkp = np.empty((pool, EK), np.uint8), np.empty((pool, 2400), np.uint8)
assert L.circl_hip_mlkem_keygen(768, seeds.ctypes.data, ekp.ctypes.data, dkp.ctypes.data, pool, 0) == 0
mis = len(sys.argv) > 2 and sys.argv[2] == "misaligned"  # a Go sub-slice is 1-byte aligned: arrays 1, 3, 5, 7, 9 bytes into their allocations


def arr(rows, cols, off, fill=None):
    raw = np.zeros(rows * cols + 64, np.uint8)         # pageable, touched
    a = raw[off:off + rows * cols].reshape(rows, cols) if mis else raw[:rows * cols].reshape(rows, cols)
    if fill is not None:
        a[:] = fill
    return a


ek = arr(n, EK, 1, np.tile(ekp, (n // pool, 1)))
m = arr(n, 32, 3, rng.integers(0, 256, (n, 32), dtype=np.uint8))
ct, ss, st = arr(n, CT, 5), arr(n, 32, 7), arr(n, 1, 9).reshape(-1)


def cpu():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


def run(dev):
    c0, t0 = cpu(), time.perf_counter()
    rc = L.circl_hip_mlkem_encaps(768, ek.ctypes.data, m.ctypes.data, ct.ctypes.data, ss.ctypes.data, st.ctypes.data, n, dev)
    assert rc == 0, rc
    return time.perf_counter() - t0, cpu() - c0


print(f"{nd} logical device(s) on {L.circl_hip_physical_device(nd - 1) + 1} HIP device(s); 2^{lg} ML-KEM-768 encapsulations from pageable memory, "
      f"{len(os.sched_getaffinity(0))} CPUs in the affinity mask, CIRCL_HIP_HOST_THREADS={os.environ.get('CIRCL_HIP_HOST_THREADS', 'default')}, "
      f"arrays {'byte-misaligned (1, 3, 5, 7, 9)' if mis else 'aligned'}")
for dev, name in ((-1, "device = -1 (all logical devices)"), (0, "device = 0")):
    run(dev)                                           # warm: pools, streams
    best = None
    for _ in range(3):
        w, c = run(dev)
        if best is None or w < best[0]:
            best = (w, c)
    w, c = best
    slots, pin, dv = C.c_int(), C.c_uint64(), C.c_uint64()
    L.circl_hip_host_pool_stats(C.byref(slots), C.byref(pin), C.byref(dv))
    per_m = c / n * 1e6
    print(f"{name}: {n / w:.3e} encaps/s, wall {w:.3f} s, host CPU {c:.2f} s = {per_m:.3f} CPU-s per 10^6 items, {c / w:.1f} CPUs busy; "
          f"staging pools: {slots.value} slots, {pin.value / 2**30:.2f} GiB page-locked, {dv.value / 2**30:.2f} GiB device")
    if dev == -1:
        for rate in (3.5e7, 4.0e7):
            print(f"    8 GPUs at {rate:.1e}/s each through host buffers = {8 * rate:.1e}/s need {8 * rate * per_m / 1e6:.1f} CPUs for the byte movers + shard threads")
ref = np.zeros((4096, CT), np.uint8)
assert L.circl_hip_mlkem_encaps(768, ek.ctypes.data, m.ctypes.data, ref.ctypes.data, ss.ctypes.data, st.ctypes.data, 4096, 0) == 0
assert (ref == ct[:4096]).all()
print("bytes equal device 0's")
