"""Not a test: wall-clock latency of small calls through the host-buffer ABI (pageable numpy arrays).   python tools/host_small.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import hostapi  # noqa: E402
from oracle import orc  # noqa: E402

rng = np.random.default_rng(1)
out = []
for n in (1, 16, 64, 256, 1024):
    ek, dk = orc.mlkem_keygen(768, rng.integers(0, 256, (n, 64), dtype=np.uint8))
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ct, ss, _ = hostapi.mlkem_encaps(768, ek, m)
    ts, td = [], []
    for _ in range(50):
        t = time.perf_counter(); hostapi.mlkem_encaps(768, ek, m); ts.append(time.perf_counter() - t)
        t = time.perf_counter(); hostapi.mlkem_decaps(768, dk, ct); td.append(time.perf_counter() - t)
    ts.sort(); td.sort()
    pub, prv = hostapi.KeyTable("mlkem-public", 768, ek[:1]), hostapi.KeyTable("mlkem-private", 768, dk[:1])  # ONE resident key
    ct1, _, _ = pub.encaps(m)
    tt, tu = [], []
    for _ in range(50):
        t = time.perf_counter(); pub.encaps(m); tt.append(time.perf_counter() - t)
        t = time.perf_counter(); prv.decaps(ct1); tu.append(time.perf_counter() - t)
    tt.sort(); tu.sort()
    pub.close(); prv.close()
    out.append(f"n={n}: encaps {ts[25] * 1e6:.0f} decaps {td[25] * 1e6:.0f}, resident key {tt[25] * 1e6:.0f} / {tu[25] * 1e6:.0f}")
print("ML-KEM-768 through host buffers, median us:", " | ".join(out), {k: v for k, v in os.environ.items() if k.startswith("CIRCL_HIP_HOST")})
