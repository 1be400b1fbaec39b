"""Not a test: ML-DSA-65 verify / sign rates against the message length: 32 B, 1 KB, 64 KB, and a batch of
32-byte messages with ONE 64 KB message in it.   python tools/msglen_bench.py [log2 n]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import _native as nat  # noqa: E402
from circl_amd import device as cdev  # noqa: E402

param = 65
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 14
L = nat.lib()


def run(n, lens, label):
    eng = cdev.MLDSADevice(param, n, "cuda", msg_len=32, sign=True)
    g = torch.Generator(device="cuda").manual_seed(3)
    pk, sk = eng.keygen(torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g))
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum(lens)
    eng.off = torch.from_numpy(off).cuda()
    msg = torch.randint(0, 256, (int(off[-1]) + 16,), dtype=torch.uint8, device="cuda", generator=g)
    eng._msg = lambda m: m.data_ptr()
    sig = eng.sign(sk, msg)
    torch.cuda.synchronize()
    res = []
    for fn in (lambda: eng.sign(sk, msg, sig), lambda: eng.verify(pk, sig, msg)):
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        res.append(best)
    ok = bool(eng.ok.all())
    print(f"ML-DSA-{param} n={n} {label:34s}: sign {res[0] * 1e3:8.2f} ms ({n / res[0]:.3e}/s)  verify {res[1] * 1e3:8.2f} ms ({n / res[1]:.3e}/s)  "
          f"{int(off[-1]) / 1e6:8.1f} MB of messages  all verify: {ok}")


n = 1 << logn
run(n, np.full(n, 32), "32-byte messages")
run(n, np.full(n, 1024), "1 KB messages")
run(n, np.full(n, 8192), "8 KB messages")
run(max(n // 16, 64), np.full(max(n // 16, 64), 65536), "64 KB messages")
mixed = np.full(n, 32)
mixed[n // 2] = 65536
run(n, mixed, "32 B, one 64 KB message")
mixed = np.full(n, 32)
mixed[::64] = 8192
run(n, mixed, "32 B, every 64th 8 KB")
