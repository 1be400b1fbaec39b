import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from circl_amd import device as cdev
n = 1 << 18
for param in (768, 1024):
    eng = cdev.MLKEMDevice(param, n)
    g = torch.Generator(device="cuda").manual_seed(1)
    seeds = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
    m = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    ek, dk = eng.keygen(seeds)
    ct = torch.empty((n, eng.CT), dtype=torch.uint8, device="cuda"); ss = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    eng.encaps(ek, m, ct, ss)
    ss2 = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    eng.decaps(dk, ct, ss2); torch.cuda.synchronize()
    cdev.profile_enable(True)
    for _ in range(3): eng.decaps(dk, ct, ss2)
    torch.cuda.synchronize()
    cdev.profile_enable(False)
    print(param, {k: "%.3f ms x%d" % (cdev.profile_read(k)[0] / max(cdev.profile_read(k)[1], 1), cdev.profile_read(k)[1]) for k in ("mlkem_hash", "mlkem_decrypt", "mlkem_encrypt")})
