"""Not a test: cost per call of the hybrid KEMs and X25519 against the batch size (device-resident, 20 stream-ordered calls per sample).
   python tools/hybrid_latency.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import device as cdev  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(1)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    b = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        b = min(b, (time.perf_counter() - t) / reps)
    return b * 1e6


for logn in (0, 6, 10, 12, 14):
    n = 1 << logn
    rnd = lambda cols: torch.randint(0, 256, (n, cols), dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
    k, u = rnd(32), rnd(32)
    out, ok = cdev.x25519(k, u)
    line = [f"n=2^{logn:<2d} X25519 shared {timed(lambda: cdev.x25519(k, u, out, ok)):7.1f} us, keygen {timed(lambda: cdev.x25519(k, None, out, ok)):7.1f} us"]
    for scheme, name in ((cdev.XWING, "X-Wing"), (cdev.X25519MLKEM768, "X25519MLKEM768")):
        h = cdev.HybridDevice(scheme, n)
        seeds, es = rnd(h.S["seed"]), rnd(h.S["eseed"])
        pk, sk = h.keygen(seeds)
        ct, ss, st = h.encaps(pk, es)
        ss2, st2 = h.decaps(sk, ct)
        torch.cuda.synchronize()
        assert bool((ss == ss2).all()) and not bool(st.any()) and not bool(st2.any())
        line.append(f"{name}: keygen {timed(lambda: h.keygen(seeds)):7.1f} encaps {timed(lambda: h.encaps(pk, es)):7.1f} decaps {timed(lambda: h.decaps(sk, ct)):7.1f} us")
    print(" | ".join(line))
