"""Not a test: A/B of signing variants on ONE box in ONE process (box-to-box spread is 3-4 %, more than most effects).
   python tools/sign_ab.py <env name> <v0,v1,...> [param] [log2 n] [rounds]   -- the env knob must be read per call by the library"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import device as cdev  # noqa: E402

name = sys.argv[1]
variants = sys.argv[2].split(",")
param = int(sys.argv[3]) if len(sys.argv) > 3 else 65
n = 1 << (int(sys.argv[4]) if len(sys.argv) > 4 else 18)
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 4
g = torch.Generator(device="cuda").manual_seed(1)
eng = cdev.MLDSADevice(param, n, "cuda", sign=True)
pk, sk = eng.keygen(torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g))
msg = torch.randint(0, 256, (n * 32 + 16,), dtype=torch.uint8, device="cuda", generator=g)
ref = None
best = {v: 1e9 for v in variants}
for r in range(rounds + 1):
    for v in variants:
        os.environ[name] = v
        sig = eng.sign(sk, msg).clone() if r == 0 else None
        if r == 0:
            torch.cuda.synchronize()
            ref = sig if ref is None else ref
            assert bool((sig == ref).all()), "variant %s gives different signatures" % v
            assert bool(eng.verify(pk, sig, msg).all())
            continue
        t = time.perf_counter()
        eng.sign(sk, msg)
        torch.cuda.synchronize()
        best[v] = min(best[v], time.perf_counter() - t)
for v in variants:
    print(f"ML-DSA-{param} n={n} {name}={v}: best of {rounds} {best[v] * 1e3:.2f} ms -> {n / best[v]:.3e}/s (signatures identical across variants)")
