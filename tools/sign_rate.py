"""Not a test: rate of one ML-DSA batch signing configuration (distinct GPU-made keys).   python tools/sign_rate.py [param] [log2 n] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import device as cdev  # noqa: E402

param = int(sys.argv[1]) if len(sys.argv) > 1 else 65
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 18)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
g = torch.Generator(device="cuda").manual_seed(1)
eng = cdev.MLDSADevice(param, n, "cuda", sign=True)
pk, sk = eng.keygen(torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g))
msg = torch.randint(0, 256, (n * 32 + 16,), dtype=torch.uint8, device="cuda", generator=g)
sig = eng.sign(sk, msg)
torch.cuda.synchronize()
best = 1e9
for _ in range(reps):
    t = time.perf_counter()
    eng.sign(sk, msg, sig)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t)
ok = bool(eng.verify(pk, sig, msg).all())
env = {k: v for k, v in os.environ.items() if k.startswith("CIRCL_HIP_SIGN")}
print(f"ML-DSA-{param} sign n={n}: best of {reps} {best * 1e3:.2f} ms -> {n / best:.3e}/s  all verify: {ok}  {env}")
