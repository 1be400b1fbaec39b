"""Not a test: ML-DSA signing latency of very small batches (which route: CIRCL_HIP_SIGN_BATCHED_MIN).   python tools/dsa_sign_small.py [param]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import device as cdev  # noqa: E402

param = int(sys.argv[1]) if len(sys.argv) > 1 else 65
out = []
for n in (1, 2, 4, 8, 16, 32):
    eng = cdev.MLDSADevice(param, n, "cuda", sign=True)
    g = torch.Generator(device="cuda").manual_seed(1)
    pk, sk = eng.keygen(torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g))
    msg = torch.randint(0, 256, (n * 32 + 16,), dtype=torch.uint8, device="cuda", generator=g)
    sig = eng.sign(sk, msg)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t = time.perf_counter()
        eng.sign(sk, msg, sig)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    ts.sort()
    assert bool(eng.verify(pk, sig, msg).all())
    out.append(f"n={n}: {ts[len(ts) // 2] * 1e6:.0f} (min {ts[0] * 1e6:.0f})")
print(f"ML-DSA-{param} sign, median us:", " | ".join(out), {k: v for k, v in os.environ.items() if k.startswith("CIRCL_HIP_SIGN")})
