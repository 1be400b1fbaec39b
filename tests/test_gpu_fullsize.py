"""BASELINE.json's full batch sizes on the GPU, checked through size-independent properties
(keygen -> encaps -> decaps round trip on all items, verification results of a tiled signed pool)
plus bit-exact oracle parity on a uniform sample.  Device-resident (torch owns the HBM buffers)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mlkem768_roundtrip_2p20_device_resident():
    # configs[1] / configs[2] shape: ML-KEM-768 Encaps + Decaps, batch 2^20 per GPU
    import torch
    from circl_amd import device as cdev
    from oracle import orc
    n = 1 << 20
    g = torch.Generator(device="cuda").manual_seed(20)
    seeds = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
    m = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    eng = cdev.MLKEMDevice(768, n)
    ek, dk = eng.keygen(seeds)
    ct = torch.empty((n, 1088), dtype=torch.uint8, device="cuda")
    ss = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    st = torch.empty(n, dtype=torch.uint8, device="cuda")
    eng.encaps(ek, m, ct, ss, st)
    ss2 = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    st2 = torch.empty(n, dtype=torch.uint8, device="cuda")
    eng.decaps(dk, ct, ss2, st2)
    torch.cuda.synchronize()
    assert int(st.sum()) == 0 and int(st2.sum()) == 0
    assert bool((ss == ss2).all())                      # all 2^20 items round-trip
    assert int((ss == 0).all(dim=1).sum()) == 0          # no untouched rows
    # corrupted ciphertexts decapsulate to something else (implicit rejection), for every item
    ct[:, 17] ^= 0x20
    ss3 = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    eng.decaps(dk, ct, ss3, st2)
    torch.cuda.synchronize()
    assert int((ss3 == ss).all(dim=1).sum()) == 0
    ct[:, 17] ^= 0x20
    # bit-exact oracle parity on a uniform sample of 2^16 items, all three operations
    idx = torch.from_numpy(np.random.default_rng(1).choice(n, 1 << 16, replace=False)).cuda()
    ek0, dk0 = orc.mlkem_keygen(768, seeds[idx].cpu().numpy())
    assert (ek0 == ek[idx].cpu().numpy()).all() and (dk0 == dk[idx].cpu().numpy()).all()
    ct0, ss0, _ = orc.mlkem_encaps(768, ek0, m[idx].cpu().numpy())
    assert (ct0 == ct[idx].cpu().numpy()).all() and (ss0 == ss[idx].cpu().numpy()).all()
    ss30, _ = orc.mlkem_decaps(768, dk0, np.ascontiguousarray(ct0 ^ np.eye(1, 1088, 17, dtype=np.uint8) * 0x20))
    assert (ss30 == ss3[idx].cpu().numpy()).all()
    # ... and the WHOLE batch once (north_star: "every ciphertext, shared secret ... is bit-exact"): all 2^20 distinct-key
    # encapsulations against the oracle (~7 s on 16 host threads), which bench.py repeats for its headline batch
    ct_all, ss_all, st_all = orc.mlkem_encaps(768, ek.cpu().numpy(), m.cpu().numpy())
    assert not st_all.any() and (ct_all == ct.cpu().numpy()).all() and (ss_all == ss.cpu().numpy()).all()


def test_mlkem1024_roundtrip_2p18():
    import torch
    from circl_amd import device as cdev
    n = 1 << 18
    g = torch.Generator(device="cuda").manual_seed(21)
    seeds = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
    m = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    eng = cdev.MLKEMDevice(1024, n)
    ek, dk = eng.keygen(seeds)
    ct = torch.empty((n, 1568), dtype=torch.uint8, device="cuda")
    ss = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    eng.encaps(ek, m, ct, ss)
    ss2 = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    eng.decaps(dk, ct, ss2)
    torch.cuda.synchronize()
    assert bool((ss == ss2).all()) and int(eng.status.sum()) == 0


@pytest.mark.parametrize("param", [65, 87])
def test_mldsa_verify_2p18_tiled_pool(param):
    # configs[3] shape: ML-DSA-65 verify, batch 2^18, >= 1 % corrupted signatures
    import torch
    from circl_amd import _native as nat
    from oracle import orc
    L = nat.lib()
    n, pool = 1 << 18, 1 << 9
    rng = np.random.default_rng(param)
    pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (pool, 32), dtype=np.uint8))
    msgs = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(pool)]
    sig = orc.mldsa_sign(param, sk, msgs)
    want = np.ones(pool, np.uint8)
    bad = rng.choice(pool, pool // 32, replace=False)
    ct = {65: 48, 87: 64}[param]
    for k, i in enumerate(bad):
        if k % 3 == 0:
            sig[i, int(rng.integers(0, ct))] ^= 1            # c~
        elif k % 3 == 1:
            sig[i, ct + int(rng.integers(0, 2000))] ^= 8     # z
        else:
            sig[i, -1] = 0xFF                                # non-canonical hint
    want[bad] = 0
    assert (orc.mldsa_verify(param, pk, sig, msgs) == want).all()
    reps = n // pool
    d_pk = torch.from_numpy(np.tile(pk, (reps, 1))).cuda()
    d_sig = torch.from_numpy(np.tile(sig, (reps, 1))).cuda()
    d_msg = torch.from_numpy(np.frombuffer(b"".join(msgs) * reps + b"\0" * 16, np.uint8).copy()).cuda()
    d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64).cuda()
    ok = torch.empty(n, dtype=torch.uint8, device="cuda")
    wsb = L.circl_hip_mldsa_workspace_size(param, n)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    rc = L.circl_hip_mldsa_verify_dev(param, d_pk.data_ptr(), d_sig.data_ptr(), d_msg.data_ptr(), d_off.data_ptr(), None, None,
                                      ok.data_ptr(), n, ws.data_ptr(), wsb, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert (ok.cpu().numpy() == np.tile(want, reps)).all()


def test_two_streams_concurrently():
    # entry points are re-entrant: two device-resident batches on two streams with separate workspaces
    import torch
    from circl_amd import device as cdev
    from oracle import orc
    n = 5000
    rng = np.random.default_rng(8)
    res = []
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    inputs = []
    for k in range(2):
        ek, _ = orc.mlkem_keygen(768, rng.integers(0, 256, (n, 64), dtype=np.uint8))
        m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        inputs.append((ek, m, torch.from_numpy(ek).cuda(), torch.from_numpy(m).cuda(), cdev.MLKEMDevice(768, n)))
    torch.cuda.synchronize()
    for rep in range(3):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                inputs[k][4].encaps(inputs[k][2], inputs[k][3])
    torch.cuda.synchronize()
    for k in range(2):
        ct0, ss0, _ = orc.mlkem_encaps(768, inputs[k][0], inputs[k][1])
        assert (inputs[k][4].ct.cpu().numpy() == ct0).all() and (inputs[k][4].ss.cpu().numpy() == ss0).all()


def test_mixed_mlkem1024_and_mldsa87_concurrently():
    # configs[4] shape per GPU (2^20 mixed items over 8 GPUs = 2^17 per device): 2^16 ML-KEM-1024 encapsulations and
    # 2^16 ML-DSA-87 verifications submitted on two streams at once, each half checked as in the tests above
    import torch
    from circl_amd import _native as nat
    from circl_amd import device as cdev
    from oracle import orc
    L = nat.lib()
    n, pool = 1 << 16, 1 << 8
    rng = np.random.default_rng(55)
    pk, sk = orc.mldsa_keygen(87, rng.integers(0, 256, (pool, 32), dtype=np.uint8))
    msgs = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(pool)]
    sig = orc.mldsa_sign(87, sk, msgs)
    want = np.ones(pool, np.uint8)
    bad = rng.choice(pool, pool // 16, replace=False)
    for i in bad:
        sig[i, 64 + int(rng.integers(0, 2000))] ^= 8
    want[bad] = 0
    reps = n // pool
    d_pk = torch.from_numpy(np.tile(pk, (reps, 1))).cuda()
    d_sig = torch.from_numpy(np.tile(sig, (reps, 1))).cuda()
    d_msg = torch.from_numpy(np.frombuffer(b"".join(msgs) * reps + b"\0" * 16, np.uint8).copy()).cuda()
    d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64).cuda()
    ok = torch.empty(n, dtype=torch.uint8, device="cuda")
    wsb = L.circl_hip_mldsa_workspace_size(87, n)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(56)
    seeds = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
    m = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    eng = cdev.MLKEMDevice(1024, n)
    ek, dk = eng.keygen(seeds)
    ct = torch.empty((n, 1568), dtype=torch.uint8, device="cuda")
    ss = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    s_kem, s_dsa = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(2):
        with torch.cuda.stream(s_kem):
            eng.encaps(ek, m, ct, ss)
        rc = L.circl_hip_mldsa_verify_dev(87, d_pk.data_ptr(), d_sig.data_ptr(), d_msg.data_ptr(), d_off.data_ptr(), None, None,
                                          ok.data_ptr(), n, ws.data_ptr(), wsb, C.c_void_p(s_dsa.cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    assert (ok.cpu().numpy() == np.tile(want, reps)).all()
    ss2 = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    eng.decaps(dk, ct, ss2)
    torch.cuda.synchronize()
    assert bool((ss == ss2).all()) and int(eng.status.sum()) == 0
    idx = np.random.default_rng(2).choice(n, 1 << 10, replace=False)
    ti = torch.from_numpy(idx).cuda()
    ct0, ss0, _ = orc.mlkem_encaps(1024, ek[ti].cpu().numpy(), m[ti].cpu().numpy())
    assert (ct0 == ct[ti].cpu().numpy()).all() and (ss0 == ss[ti].cpu().numpy()).all()


@pytest.mark.parametrize("param", [65])
def test_mldsa_verify_2p18_distinct_keys(param):
    # configs[3] as SURVEY 8(d) words it: DISTINCT pk_i from seeded keygen, signatures from the deterministic signer, >= 1 %
    # corrupted (bit flip in z, in c~, non-canonical hint).  Keys and signatures are made on the GPU (both parity-pinned:
    # ACVP keyGen / sigGen, KAT hashes) and re-checked here: the oracle verifies a 2^14 sample of them and signs a 2^10 one.
    import torch
    from circl_amd import device as cdev
    from oracle import orc
    n = 1 << 18
    g = torch.Generator(device="cuda").manual_seed(param)
    seeds = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    msg = torch.randint(0, 256, (n * 32 + 16,), dtype=torch.uint8, device="cuda", generator=g)
    eng = cdev.MLDSADevice(param, n, "cuda", msg_len=32, sign=True)
    pk, sk = eng.keygen(seeds)
    sig = eng.sign(sk, msg)
    torch.cuda.synchronize()
    assert int((pk[1:] == pk[:-1]).all(dim=1).sum()) == 0  # keys are distinct
    good = sig.clone()
    bad = torch.arange(0, n, 64, device="cuda")
    kinds = torch.arange(bad.numel(), device="cuda") % 3
    ct = {44: 32, 65: 48, 87: 64}[param]
    sig[bad[kinds == 0], ct + 200] ^= 0x10
    sig[bad[kinds == 1], 5] ^= 0x80
    sig[bad[kinds == 2], eng.SIG - 1] = 0xFF
    ok = eng.verify(pk, sig, msg)
    torch.cuda.synchronize()
    want = torch.ones(n, dtype=torch.uint8, device="cuda")
    want[bad] = 0
    assert bool((ok == want).all())
    idx = np.sort(np.random.default_rng(3).choice(n, 1 << 14, replace=False))
    ti = torch.from_numpy(idx).cuda()
    msgs = [bytes(r) for r in msg[:n * 32].view(n, 32)[ti].cpu().numpy()]
    assert (orc.mldsa_verify(param, pk[ti].cpu().numpy(), sig[ti].cpu().numpy(), msgs) == ok[ti].cpu().numpy()).all()
    si = ti[:1 << 10]
    assert (orc.mldsa_sign(param, sk[si].cpu().numpy(), msgs[:1 << 10]) == good[si].cpu().numpy()).all()
    pk0, sk0 = orc.mldsa_keygen(param, seeds[si].cpu().numpy())
    assert (pk0 == pk[si].cpu().numpy()).all() and (sk0 == sk[si].cpu().numpy()).all()


def test_sign_dev_is_asynchronous():
    # the signer returns before its work is done (no hidden stream synchronisation): the enqueue takes a fraction of the
    # time to completion, and work queued behind it on the same stream sees the finished signatures
    import time
    import torch
    from circl_amd import device as cdev
    n = 1 << 16
    g = torch.Generator(device="cuda").manual_seed(9)
    eng = cdev.MLDSADevice(65, n, "cuda", msg_len=32, sign=True)
    pk, sk = eng.keygen(torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g))
    msg = torch.randint(0, 256, (n * 32 + 16,), dtype=torch.uint8, device="cuda", generator=g)
    sig = eng.sign(sk, msg)
    torch.cuda.synchronize()
    sig.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.sign(sk, msg, sig)
    ok = eng.verify(pk, sig, msg)       # same stream, no synchronisation in between
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    assert bool(ok.all())
    print(f"sign+verify 2^16: enqueued in {t_enq * 1e3:.2f} ms, complete after {t_all * 1e3:.2f} ms")
    assert t_enq < 0.8 * t_all


def test_sign_split_path_equals_single_stream_and_concurrent_callers():
    # batches of >= 2^16 items are signed as two halves on two library-owned streams (api_mldsa.hip): same signatures as the
    # one-stream path (CIRCL_HIP_SIGN_NOSPLIT=1, in a subprocess: the switch is read once), an odd batch size, and two
    # callers on two streams of their own sharing the library's pair
    import subprocess
    import sys
    import threading
    import torch
    from circl_amd import device as cdev
    n = (1 << 16) + 37
    g = torch.Generator(device="cuda").manual_seed(21)
    eng = cdev.MLDSADevice(65, n, "cuda", msg_len=32, sign=True)
    seeds = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    pk, sk = eng.keygen(seeds)
    msg = torch.randint(0, 256, (n * 32 + 16,), dtype=torch.uint8, device="cuda", generator=g)
    sig = eng.sign(sk, msg).clone()
    assert bool(eng.verify(pk, sig, msg).all())
    import hashlib
    digest = hashlib.sha256(sig.cpu().numpy().tobytes()).hexdigest()
    code = (
        "import sys, hashlib, torch; sys.path.insert(0, %r)\n"
        "from circl_amd import device as cdev\n"
        "n = (1 << 16) + 37; g = torch.Generator(device='cuda').manual_seed(21)\n"
        "eng = cdev.MLDSADevice(65, n, 'cuda', msg_len=32, sign=True)\n"
        "seeds = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device='cuda', generator=g)\n"
        "pk, sk = eng.keygen(seeds)\n"
        "msg = torch.randint(0, 256, (n * 32 + 16,), dtype=torch.uint8, device='cuda', generator=g)\n"
        "print(hashlib.sha256(eng.sign(sk, msg).cpu().numpy().tobytes()).hexdigest())\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env=dict(os.environ, CIRCL_HIP_SIGN_NOSPLIT="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().split("\n")[-1] == digest

    # two concurrent callers, each with its own engine (workspace) and stream
    engs = [cdev.MLDSADevice(65, n, "cuda", msg_len=32, sign=True) for _ in range(2)]
    outs = [None, None]

    def work(i):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2):
                outs[i] = engs[i].sign(sk, msg)
        s.synchronize()

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert bool((outs[0] == sig).all()) and bool((outs[1] == sig).all())
