"""Not a test: ML-KEM-768 encapsulation through the host-buffer C ABI (page-locked and pageable caller memory), the
tuning knobs of the staging pipeline taken from the environment (CIRCL_HIP_HOST_CHUNK / _DEPTH / _THREADS / _NT).

    python tools/host_path.py [log2 n]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import _native as nat  # noqa: E402
from circl_amd import hostapi  # noqa: E402

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
L = nat.lib()
rng = np.random.default_rng(3)
pool = 1 << 12
ekp, _ = hostapi.mlkem_keygen(768, rng.integers(0, 256, (pool, 64), dtype=np.uint8))
ek = np.tile(ekp, (n // pool, 1))
m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
EK, CT = 1184, 1088
ct, ss, st = np.zeros((n, CT), np.uint8), np.zeros((n, 32), np.uint8), np.zeros(n, np.uint8)
p = lambda a: a.ctypes.data_as(C.c_void_p)


def run(args, label):
    L.circl_hip_mlkem_encaps(768, *args, n, 0)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        rc = L.circl_hip_mlkem_encaps(768, *args, n, 0)
        ts.append(time.perf_counter() - t)
        assert rc == 0
    best = min(ts)
    print(f"{label:10s} n={n}: best {best * 1e3:7.2f} ms  median {sorted(ts)[2] * 1e3:7.2f} ms -> {n / best:.3e}/s  "
          f"H2D {n * (EK + 32) / best / 1e9:5.1f} GB/s  D2H {n * (CT + 33) / best / 1e9:5.1f} GB/s", flush=True)


run((p(ek), p(m), p(ct), p(ss), p(st)), "pageable")
ref = ct.copy()


def pinned(nbytes):
    q = L.circl_hip_alloc_host(nbytes)
    return q, np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(q))


p_ek, a_ek = pinned(n * EK); p_m, a_m = pinned(n * 32); p_ct, a_ct = pinned(n * CT); p_ss, a_ss = pinned(n * 32); p_st, a_st = pinned(n)
a_ek[:] = ek.reshape(-1); a_m[:] = m.reshape(-1)
run((p_ek, p_m, p_ct, p_ss, p_st), "pinned")
print("same output:", bool((a_ct.reshape(n, CT) == ref).all()), " env:", {k: v for k, v in os.environ.items() if k.startswith("CIRCL_HIP_HOST")})
