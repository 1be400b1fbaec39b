"""X-Wing / X25519MLKEM768 / X25519 batch rates on resident inputs and through host buffers.  Not a test.
   python tools/hybrid_bench.py [log2 n]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C  # noqa: E402

from circl_amd import _native as nat, device as cdev  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << logn
g = torch.Generator(device="cuda").manual_seed(1)


def rnd(cols):
    return torch.randint(0, 256, (n, cols), dtype=torch.uint8, device="cuda", generator=g)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


HOST_ONLY = len(sys.argv) > 2 and sys.argv[2] == "host"
k, u = rnd(32), rnd(32)
out, ok = cdev.x25519(k, u)
for name, pt in (() if HOST_ONLY else (("Shared", u), ("KeyGen", None))):
    dt = timed(lambda: cdev.x25519(k, pt, out, ok))
    print(f"X25519 {name}: 2^{logn} in {dt * 1e3:.2f} ms -> {n / dt:.3e} /s (device-resident)")
for scheme, name in ((cdev.XWING, "X-Wing"),) + (() if HOST_ONLY else ((cdev.X25519MLKEM768, "X25519MLKEM768"),)):
    h = cdev.HybridDevice(scheme, n)
    seeds, es = rnd(h.S["seed"]), rnd(h.S["eseed"])
    pk, sk = h.keygen(seeds)
    ct, ss, st = h.encaps(pk, es)
    ss2, st2 = h.decaps(sk, ct)
    torch.cuda.synchronize()
    assert bool((ss == ss2).all()) and not bool(st.any()) and not bool(st2.any())
    for op, fn in (() if HOST_ONLY else (("DeriveKeyPair", lambda: h.keygen(seeds)), ("Encapsulate", lambda: h.encaps(pk, es)), ("Decapsulate", lambda: h.decaps(sk, ct)))):
        dt = timed(fn)
        print(f"{name} {op}: 2^{logn} in {dt * 1e3:.2f} ms -> {n / dt:.3e} /s (device-resident)")
    # host-buffer ABI: pageable numpy arrays, outputs allocated and touched once (a Go caller reuses its slices)
    L = nat.lib()
    vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    pkh, esh, skh = pk.cpu().numpy(), es.cpu().numpy(), sk.cpu().numpy()
    cth, ssh, sth, ss2h = np.zeros((n, h.S["ct"]), np.uint8), np.zeros((n, h.S["ss"]), np.uint8), np.zeros(n, np.uint8), np.zeros((n, h.S["ss"]), np.uint8)
    for rep in range(3):
        t = time.perf_counter()
        nat.check(L.circl_hip_hybrid_encaps(scheme, vp(pkh), vp(esh), vp(cth), vp(ssh), vp(sth), n, 0), "encaps")
        dte = time.perf_counter() - t
        t = time.perf_counter()
        nat.check(L.circl_hip_hybrid_decaps(scheme, vp(skh), vp(cth), vp(ss2h), vp(sth), n, 0), "decaps")
        dtd = time.perf_counter() - t
    assert (ssh == ss.cpu().numpy()).all() and (ss2h == ssh).all()
    print(f"{name} through host buffers (pageable): Encapsulate {dte * 1e3:.2f} ms -> {n / dte:.3e} /s, Decapsulate {dtd * 1e3:.2f} ms -> {n / dtd:.3e} /s")
    del h
