"""Not a test: ML-DSA verify / sign latency against the batch size (device-resident, 32-byte messages).   python tools/dsa_latency.py [param]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import device as cdev  # noqa: E402

param = int(sys.argv[1]) if len(sys.argv) > 1 else 65
for logn in ([int(x) for x in os.environ["CIRCL_LATENCY_LOGNS"].split(",")] if os.environ.get("CIRCL_LATENCY_LOGNS") else (0, 4, 6, 8, 10, 12, 14)):
    n = 1 << logn
    eng = cdev.MLDSADevice(param, n, "cuda", sign=True)
    g = torch.Generator(device="cuda").manual_seed(1)
    pk, sk = eng.keygen(torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g))
    msg = torch.randint(0, 256, (n * 32 + 16,), dtype=torch.uint8, device="cuda", generator=g)
    sig = eng.sign(sk, msg)
    torch.cuda.synchronize()
    out = []
    for fn in (lambda: eng.verify(pk, sig, msg), lambda: eng.verify_shared(pk[:1], sig, msg), lambda: eng.sign(sk, msg, sig), lambda: eng.keygen(msg[:n * 32].view(n, 32))):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        out.append(best * 1e6)
    print(f"ML-DSA-{param} n=2^{logn:<2d}: verify {out[0]:8.1f} us | verify, one key {out[1]:8.1f} us | sign {out[2]:8.1f} us | keygen {out[3]:8.1f} us")
