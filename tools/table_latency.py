"""Not a test: cost per call through a resident key table of ONE key (circl_hip_*_keytable_new) against the one-key entry points that
parse the key on every call (device-resident, 20 stream-ordered calls per sample).   python tools/table_latency.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import device as cdev, hostapi  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(1)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    b = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        b = min(b, (time.perf_counter() - t) / reps)
    return b * 1e6


kg = cdev.MLKEMDevice(768, 1)
ek1, dk1 = kg.keygen(torch.randint(0, 256, (1, 64), dtype=torch.uint8, device="cuda", generator=g))
pub = hostapi.KeyTable("mlkem-public", 768, ek1.cpu().numpy())
prv = hostapi.KeyTable("mlkem-private", 768, dk1.cpu().numpy())
dg = cdev.MLDSADevice(65, 1, "cuda", sign=True)
pk1, sk1 = dg.keygen(torch.randint(0, 256, (1, 32), dtype=torch.uint8, device="cuda", generator=g))
vt = hostapi.KeyTable("mldsa-public", 65, pk1.cpu().numpy())
st_ = hostapi.KeyTable("mldsa-private", 65, sk1.cpu().numpy())
for logn in [int(x) for x in os.environ.get("CIRCL_LATENCY_LOGNS", "0,6,10,12,14,16,18").split(",")]:
    n = 1 << logn
    eng = cdev.MLKEMDevice(768, n)
    m = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    ct, ss, st = eng.encaps_shared(ek1, m)
    ct = ct.clone()
    ss2, st2 = torch.empty_like(ss), torch.empty_like(st)
    ct_t, ss_t, _ = eng.encaps_table(pub, m, ct=torch.empty_like(ct), ss=torch.empty_like(ss))
    torch.cuda.synchronize()
    assert bool((ct_t == ct).all()) and bool((ss_t == ss).all())
    line = (f"n=2^{logn:<2d} ML-KEM-768 encaps: one key {timed(lambda: eng.encaps_shared(ek1, m)):7.1f} -> table {timed(lambda: eng.encaps_table(pub, m)):7.1f} us | "
            f"decaps: {timed(lambda: eng.decaps_shared(dk1, ct, ss2, st2)):7.1f} -> {timed(lambda: eng.decaps_table(prv, ct, ss=ss2, status=st2)):7.1f} us")
    if logn <= 14:
        d = cdev.MLDSADevice(65, n, "cuda", sign=True)
        msg = torch.randint(0, 256, (n * 32 + 16,), dtype=torch.uint8, device="cuda", generator=g)
        sig = d.sign(sk1, msg, shared=True)
        ok = d.verify_table(vt, sig, msg)
        torch.cuda.synchronize()
        assert bool(ok.all())
        line += f" | ML-DSA-65 verify: {timed(lambda: d.verify_shared(pk1, sig, msg)):7.1f} -> {timed(lambda: d.verify_table(vt, sig, msg)):7.1f} us"
        sig2 = d.sign_table(st_, msg)
        torch.cuda.synchronize()
        assert bool((sig2 == sig).all())
        line += f" | sign: {timed(lambda: d.sign(sk1, msg, sig, shared=True), 10):7.1f} -> {timed(lambda: d.sign_table(st_, msg, sig), 10):7.1f} us"
    print(line)
