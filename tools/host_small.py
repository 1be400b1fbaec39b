"""Not a test: wall-clock latency of small calls through the host-buffer ABI (pageable numpy arrays).   python tools/host_small.py [reps]

Per size: 50 calls per sample, `reps` samples (default 5) after a warm-up burst that brings the chip to its working clock; printed as
median of the samples' medians [min-max of them].  profiles/r05_host_small.txt holds its output from several boxes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import hostapi  # noqa: E402
from oracle import orc  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(1)
# warm-up: half a second of real work (a cold chip starts at a low clock: profiles/r04_kem_round_sweep.txt)
ekw, _ = orc.mlkem_keygen(768, rng.integers(0, 256, (1 << 14, 64), dtype=np.uint8))
mw = rng.integers(0, 256, (1 << 14, 32), dtype=np.uint8)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    hostapi.mlkem_encaps(768, ekw, mw)


def sample(fn):
    ts = []
    for _ in range(50):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    ts.sort()
    return ts[25] * 1e6


def stat(fn):
    fn()
    v = sorted(sample(fn) for _ in range(reps))
    return "%.0f [%.0f-%.0f]" % (v[len(v) // 2], v[0], v[-1])


print("ML-KEM-768 through host buffers, us per call: median of %d samples of 50 calls [min-max]" % reps,
      {k: v for k, v in os.environ.items() if k.startswith("CIRCL_HIP_")})
for n in (1, 16, 64, 256, 1024, 4096):
    ek, dk = orc.mlkem_keygen(768, rng.integers(0, 256, (n, 64), dtype=np.uint8))
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ct, ss, _ = hostapi.mlkem_encaps(768, ek, m)
    e, d = stat(lambda: hostapi.mlkem_encaps(768, ek, m)), stat(lambda: hostapi.mlkem_decaps(768, dk, ct))
    pub, prv = hostapi.KeyTable("mlkem-public", 768, ek[:1]), hostapi.KeyTable("mlkem-private", 768, dk[:1])  # ONE resident key
    ct1, _, _ = pub.encaps(m)
    te, td = stat(lambda: pub.encaps(m)), stat(lambda: prv.decaps(ct1))
    pub.close(); prv.close()
    print(f"n={n:5d}: encaps {e:>16s}  decaps {d:>16s} | resident key: encaps {te:>16s}  decaps {td:>16s}", flush=True)
