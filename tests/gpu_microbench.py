"""Not a test: device-resident timings of every batch operation (run on the GPU box via gpurun).

    python tests/gpu_microbench.py [logn]
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import _native as nat  # noqa: E402
from circl_amd import device as cdev  # noqa: E402
from oracle import orc  # noqa: E402


def timeit(fn, iters=5, warm_ms=40.0):
    """Mean device time of `iters` stream-ordered calls.  The calls before the timed ones keep the GPU busy for ~warm_ms: a chip that
    idled while the host prepared inputs starts at a low clock, and a 2 ms batch timed cold reads up to 18 % slow
    (2^18 encapsulations: 2.24 ms cold, 1.90 ms warm; tools/kem_round_sweep.py)."""
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < warm_ms:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def kem(param, n, pool=1 << 12):
    rng = np.random.default_rng(param)
    eng = cdev.MLKEMDevice(param, n)
    seeds = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
    ek, dk = eng.keygen(seeds)
    ms = timeit(lambda: eng.keygen(seeds, ek, dk))
    print(f"ML-KEM-{param} keygen  n={n}: {ms:8.3f} ms -> {n / ms * 1e3:.3e}/s")
    m = torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).cuda()
    ct = torch.empty((n, eng.CT), dtype=torch.uint8, device="cuda")
    ss = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: eng.encaps(ek, m, ct, ss))
    print(f"ML-KEM-{param} encaps  n={n}: {ms:8.3f} ms -> {n / ms * 1e3:.3e}/s")
    L = nat.lib()
    wsb = L.circl_hip_mlkem_workspace_size(param, n)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    st1 = torch.empty(n, dtype=torch.uint8, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ct_s, ss_s = torch.empty_like(ct), torch.empty_like(ss)

    def shared():
        rc = L.circl_hip_mlkem_encaps_shared_dev(param, ek[:1].data_ptr(), m.data_ptr(), ct_s.data_ptr(), ss_s.data_ptr(), st1.data_ptr(), n,
                                                 ws.data_ptr(), wsb, stream)
        assert rc == 0, rc
    ms = timeit(shared)
    print(f"ML-KEM-{param} encaps, shared key  n={n}: {ms:8.3f} ms -> {n / ms * 1e3:.3e}/s")
    ss2 = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: eng.decaps(dk, ct, ss2))
    torch.cuda.synchronize()
    print(f"ML-KEM-{param} decaps  n={n}: {ms:8.3f} ms -> {n / ms * 1e3:.3e}/s   roundtrip ok: {bool((ss == ss2).all())}, status {int(eng.status.sum())}")
    ss3 = torch.empty_like(ss2)

    def dshared():
        rc = L.circl_hip_mlkem_decaps_shared_dev(param, dk[:1].data_ptr(), ct_s.data_ptr(), ss3.data_ptr(), st1.data_ptr(), n, ws.data_ptr(), wsb, stream)
        assert rc == 0, rc
    ms = timeit(dshared)
    torch.cuda.synchronize()
    print(f"ML-KEM-{param} decaps, shared key  n={n}: {ms:8.3f} ms -> {n / ms * 1e3:.3e}/s   roundtrip ok: {bool((ss3 == ss_s).all())}")
    idx = rng.integers(0, n, 128)
    ek0, dk0 = orc.mlkem_keygen(param, seeds[idx].cpu().numpy())
    print("   keygen parity:", bool((ek0 == ek[idx].cpu().numpy()).all() and (dk0 == dk[idx].cpu().numpy()).all()))


def dsa(param, n, pool=1 << 11):
    L = nat.lib()
    PK, SK, SIG = orc.DSA_SIZES[param]
    rng = np.random.default_rng(param)
    pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (pool, 32), dtype=np.uint8))
    msgs = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(pool)]
    sig = orc.mldsa_sign(param, sk, msgs)
    sig0 = sig.copy()
    bad = rng.choice(pool, pool // 64, replace=False)
    sig[bad, 100] ^= 4
    reps = n // pool
    d_pk = torch.from_numpy(np.tile(pk, (reps, 1))).cuda()
    d_sig = torch.from_numpy(np.tile(sig, (reps, 1))).cuda()
    d_msg = torch.from_numpy(np.frombuffer(b"".join(msgs) * reps, np.uint8).copy()).cuda()
    d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64).cuda()
    ok = torch.empty(n, dtype=torch.uint8, device="cuda")
    wsb = L.circl_hip_mldsa_workspace_size(param, n)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        rc = L.circl_hip_mldsa_verify_dev(param, d_pk.data_ptr(), d_sig.data_ptr(), d_msg.data_ptr(), d_off.data_ptr(), None, None,
                                          ok.data_ptr(), n, ws.data_ptr(), wsb, st)
        assert rc == 0, rc
    ms = timeit(run, 3)
    want = np.ones(pool, np.uint8)
    want[bad] = 0
    good = bool((ok.cpu().numpy() == np.tile(want, reps)).all())
    print(f"ML-DSA-{param} verify  n={n}: {ms:8.3f} ms -> {n / ms * 1e3:.3e}/s   results as expected: {good}")
    # shared key: the pool's first key for every item (signatures of other keys fail, the work is the same)
    def run_shared():
        rc = L.circl_hip_mldsa_verify_shared_dev(param, d_pk.data_ptr(), d_sig.data_ptr(), d_msg.data_ptr(), d_off.data_ptr(), None, None,
                                                 ok.data_ptr(), n, ws.data_ptr(), wsb, st)
        assert rc == 0, rc
    ms = timeit(run_shared)
    print(f"ML-DSA-{param} verify, shared key  n={n}: {ms:8.3f} ms -> {n / ms * 1e3:.3e}/s")
    seeds = torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).cuda()
    kpk = torch.empty((n, PK), dtype=torch.uint8, device="cuda")
    ksk = torch.empty((n, SK), dtype=torch.uint8, device="cuda")

    def kg():
        rc = L.circl_hip_mldsa_keygen_dev(param, seeds.data_ptr(), kpk.data_ptr(), ksk.data_ptr(), n, ws.data_ptr(), wsb, st)
        assert rc == 0, rc
    ms = timeit(kg, 3)
    print(f"ML-DSA-{param} keygen  n={n}: {ms:8.3f} ms -> {n / ms * 1e3:.3e}/s")
    ns = n
    d_sk = torch.from_numpy(np.tile(sk, (max(ns // pool, 1), 1))[:ns].copy()).cuda()
    d_rnd = torch.zeros((ns, 32), dtype=torch.uint8, device="cuda")
    ssig = torch.empty((ns, SIG), dtype=torch.uint8, device="cuda")
    swsb = L.circl_hip_mldsa_sign_workspace_size(param, ns)
    sws = torch.empty(swsb, dtype=torch.uint8, device="cuda")

    def sg():
        rc = L.circl_hip_mldsa_sign_dev(param, d_sk.data_ptr(), d_msg.data_ptr(), d_off.data_ptr(), None, None, d_rnd.data_ptr(), 0,
                                        ssig.data_ptr(), ns, sws.data_ptr(), swsb, st)
        assert rc == 0, rc
    ms = timeit(sg, 2)
    same = bool((ssig[:pool].cpu().numpy() == sig0[:min(pool, ns)]).all()) if ns >= pool else None
    print(f"ML-DSA-{param} sign    n={ns}: {ms:8.3f} ms -> {ns / ms * 1e3:.3e}/s   equals oracle signatures: {same}")

    def sg_shared():
        rc = L.circl_hip_mldsa_sign_shared_dev(param, d_sk.data_ptr(), d_msg.data_ptr(), d_off.data_ptr(), None, None, d_rnd.data_ptr(), 0,
                                               ssig.data_ptr(), ns, sws.data_ptr(), swsb, st)
        assert rc == 0, rc
    ms = timeit(sg_shared, 2)
    print(f"ML-DSA-{param} sign, shared key  n={ns}: {ms:8.3f} ms -> {ns / ms * 1e3:.3e}/s")
    cdev.profile_enable(True)
    run(); torch.cuda.synchronize()
    cdev.profile_enable(False)
    print("   kernel ms: prep+final %.3f  verify %.3f" % (cdev.profile_read("mldsa_hash")[0], cdev.profile_read("mldsa_verify")[0]))


def latency_curve(param=768):
    """Device-resident ML-KEM encapsulation latency and rate against the batch size (stream-ordered, one call per sample);
    plus the wall-clock latency of a batch of one through the host-buffer ABI."""
    from circl_amd import hostapi
    rng = np.random.default_rng(1)
    logns = [int(x) for x in os.environ["CIRCL_LATENCY_LOGNS"].split(",")] if os.environ.get("CIRCL_LATENCY_LOGNS") else (0, 6, 10, 11, 12, 13, 14, 15, 16, 18, 20)
    for logn in logns:
        n = 1 << logn
        eng = cdev.MLKEMDevice(param, n)
        seeds = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
        ek, dk = eng.keygen(seeds)
        m = torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).cuda()
        ct = torch.empty((n, eng.CT), dtype=torch.uint8, device="cuda")
        ss = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
        ms = timeit(lambda: eng.encaps(ek, m, ct, ss), 20 if n < (1 << 18) else 5)
        print(f"ML-KEM-{param} encaps  n=2^{logn:<2d}: {ms * 1e3:10.1f} us per call -> {n / ms * 1e3:.3e}/s")
        if os.environ.get("CIRCL_LATENCY_ALL"):  # the other operations a TLS front-end batches: decapsulation, per-item and under ONE key
            ss2 = torch.empty_like(ss)
            md = timeit(lambda: eng.decaps(dk, ct, ss2), 20 if n < (1 << 18) else 5)
            ms1 = timeit(lambda: eng.encaps_shared(ek[:1], m, ct, ss), 20 if n < (1 << 18) else 5)
            mds = timeit(lambda: eng.decaps_shared(dk[:1], ct, ss2), 20 if n < (1 << 18) else 5)
            mk = timeit(lambda: eng.keygen(seeds), 20 if n < (1 << 18) else 5)
            print(f"ML-KEM-{param}    decaps {md * 1e3:8.1f} us ({n / md * 1e3:.3e}/s) | encaps, one key {ms1 * 1e3:8.1f} us ({n / ms1 * 1e3:.3e}/s) | "
                  f"decaps, one key {mds * 1e3:8.1f} us ({n / mds * 1e3:.3e}/s) | keygen {mk * 1e3:8.1f} us ({n / mk * 1e3:.3e}/s)")
    ek1, _ = orc.mlkem_keygen(param, rng.integers(0, 256, (1, 64), dtype=np.uint8))
    m1 = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    hostapi.mlkem_encaps(param, ek1, m1)
    t = time.perf_counter()
    for _ in range(50):
        hostapi.mlkem_encaps(param, ek1, m1)
    print(f"ML-KEM-{param} encaps  n=1 through the host-buffer ABI (H2D + 2 kernels + D2H + sync): {(time.perf_counter() - t) / 50 * 1e6:.0f} us")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "latency":
    latency_curve(int(sys.argv[3]) if len(sys.argv) > 3 else 768)
    sys.exit(0)

if __name__ == "__main__" and not (len(sys.argv) > 2 and sys.argv[2] == "host"):
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    for p in (768, 512, 1024):
        kem(p, 1 << logn)
    for p in (65, 44, 87):
        dsa(p, 1 << (logn - 2))


def host_path(param=768, n=1 << 20):
    """End-to-end through the host-buffer C ABI (pageable numpy memory): includes H2D + D2H over PCIe."""
    from circl_amd import hostapi
    rng = np.random.default_rng(3)
    pool = 1 << 12
    ekp, _ = orc.mlkem_keygen(param, rng.integers(0, 256, (pool, 64), dtype=np.uint8))
    ek = np.tile(ekp, (n // pool, 1))
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    hostapi.mlkem_encaps(param, ek[:4096], m[:4096])
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        ct, ss, st = hostapi.mlkem_encaps(param, ek, m)
        best = min(best, time.perf_counter() - t)
    print(f"ML-KEM-{param} encaps through host-buffer ABI (pageable, PCIe-inclusive) n={n}: {best * 1e3:.1f} ms -> {n / best:.3e}/s")
    # the same call on page-locked buffers from circl_hip_alloc_host
    L = nat.lib()
    EK, CT = 1184, 1088

    def pinned(nbytes):
        p = L.circl_hip_alloc_host(nbytes)
        assert p
        return p, np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p))
    p_ek, a_ek = pinned(n * EK); p_m, a_m = pinned(n * 32); p_ct, a_ct = pinned(n * CT); p_ss, a_ss = pinned(n * 32); p_st, a_st = pinned(n)
    a_ek[:] = ek.reshape(-1); a_m[:] = m.reshape(-1)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        rc = L.circl_hip_mlkem_encaps(param, p_ek, p_m, p_ct, p_ss, p_st, n, 0)
        best = min(best, time.perf_counter() - t)
        assert rc == 0
    ok = bool((a_ct.reshape(n, CT)[:1000] == ct[:1000]).all())
    print(f"ML-KEM-{param} encaps through host-buffer ABI (pinned, PCIe-inclusive)   n={n}: {best * 1e3:.1f} ms -> {n / best:.3e}/s  same output: {ok}")
    for p_ in (p_ek, p_m, p_ct, p_ss, p_st):
        L.circl_hip_free_host(p_)


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "host":
    host_path()
