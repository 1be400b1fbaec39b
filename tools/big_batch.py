import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, time
from circl_amd import device as cdev
for logn in (22, 23):
    n = 1 << logn
    g = torch.Generator(device="cuda").manual_seed(logn)
    seeds = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
    m = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    eng = cdev.MLKEMDevice(768, n)
    ek, dk = eng.keygen(seeds)
    ct = torch.empty((n, 1088), dtype=torch.uint8, device="cuda"); ss = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize(); t = time.perf_counter()
    eng.encaps(ek, m, ct, ss); torch.cuda.synchronize(); te = time.perf_counter() - t
    ss2 = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    t = time.perf_counter(); eng.decaps(dk, ct, ss2); torch.cuda.synchronize(); td = time.perf_counter() - t
    print(f"n=2^{logn}: encaps {n/te:.3e}/s decaps {n/td:.3e}/s roundtrip {bool((ss==ss2).all())} status {int(eng.status.sum())} mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    del eng, ek, dk, ct, ss, ss2, seeds, m
