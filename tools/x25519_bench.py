"""X25519 batch rate on resident inputs (circl_hip_x25519_dev) and through host buffers.  Not a test."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import _native as nat, hostapi  # noqa: E402

L = nat.lib()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
g = torch.Generator(device="cuda").manual_seed(1)
k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
u = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
out = torch.empty_like(k)
ok = torch.empty(n, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for name, pt in (("Shared (variable point)", u), ("KeyGen (base point)", None)):
    for rep in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        nat.check(L.circl_hip_x25519_dev(k.data_ptr(), pt.data_ptr() if pt is not None else None, out.data_ptr(), ok.data_ptr(), n, st), "x25519_dev")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(f"X25519 {name}: n = 2^{logn}: {dt * 1e3:.2f} ms -> {n / dt:.3e} /s (device-resident)")
kh, uh = k.cpu().numpy(), u.cpu().numpy()
for rep in range(2):
    t = time.perf_counter()
    hostapi.x25519(kh, uh)
    dt = time.perf_counter() - t
print(f"X25519 Shared through host buffers: {dt * 1e3:.2f} ms -> {n / dt:.3e} /s")
