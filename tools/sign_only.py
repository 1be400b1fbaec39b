"""Not a test: one ML-DSA batch signing call (distinct GPU-made keys), for counter passes.   python tools/sign_only.py [param] [log2 n]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import device as cdev  # noqa: E402

os.environ.setdefault("CIRCL_HIP_SIGN_NOSPLIT", "1")  # one stream: the counters of a round kernel belong to one launch
param = int(sys.argv[1]) if len(sys.argv) > 1 else 65
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 17)
g = torch.Generator(device="cuda").manual_seed(1)
eng = cdev.MLDSADevice(param, n, "cuda", sign=True)
pk, sk = eng.keygen(torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g))
msg = torch.randint(0, 256, (n * 32 + 16,), dtype=torch.uint8, device="cuda", generator=g)
torch.cuda.synchronize()
t = time.perf_counter()
sig = eng.sign(sk, msg)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"ML-DSA-{param} sign n={n}: {dt * 1e3:.3f} ms -> {n / dt:.3e}/s (first call)")
