"""Experiment, not a test: one signing call over n items against P concurrent calls over n / P items on P streams
(what an in-library split of the batch would do).   python tools/sign_split.py [param] [log2 n]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import device as cdev  # noqa: E402

param = int(sys.argv[1]) if len(sys.argv) > 1 else 65
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 18
n = 1 << logn
g = torch.Generator(device="cuda").manual_seed(1)
eng = cdev.MLDSADevice(param, n, "cuda", sign=True)
pk, sk = eng.keygen(torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g))
msg = torch.randint(0, 256, (n * 32 + 16,), dtype=torch.uint8, device="cuda", generator=g)
sig = eng.sign(sk, msg)
torch.cuda.synchronize()
for _ in range(2):
    t = time.perf_counter()
    eng.sign(sk, msg, sig)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
print(f"ML-DSA-{param} sign n=2^{logn}, one call: {dt * 1e3:.2f} ms -> {n / dt:.3e}/s")
ref = sig.clone()
for P in (2, 4):
    h = n // P
    engs = [cdev.MLDSADevice(param, h, "cuda", sign=True) for _ in range(P)]
    streams = [torch.cuda.Stream() for _ in range(P)]
    sks = [sk[p * h:(p + 1) * h].contiguous() for p in range(P)]
    msgs = [msg[p * h * 32:(p + 1) * h * 32 + 16].clone() for p in range(P)]
    sigs = [None] * P
    for rep in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for p in range(P):
            with torch.cuda.stream(streams[p]):
                sigs[p] = engs[p].sign(sks[p], msgs[p], sigs[p])
        te = time.perf_counter() - t
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    same = all(bool((sigs[p] == ref[p * h:(p + 1) * h]).all()) for p in range(P))
    print(f"  {P} calls of 2^{logn}/{P} on {P} streams: enqueue {te * 1e3:.2f} ms, complete {dt * 1e3:.2f} ms -> {n / dt:.3e}/s  same signatures: {same}")
    del engs
