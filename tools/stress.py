"""Not a test: randomized soak of the batch paths against the oracle (run on the GPU box).  Looks for rare races
(speculative signing tail, scratch-row reuse, ticket scheduling) by varying batch sizes and repeating."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import hostapi  # noqa: E402
from oracle import orc, hybrid as ohyb  # noqa: E402



def _same(a, b, what):
    assert (np.asarray(a) == np.asarray(b)).all(), what


def drive(q, calls, width):
    """The asynchronous form from one thread: `calls` = [(submit, check)], at most `width` tickets outstanding; q.poll / q.wait as KeyTable and
    CallQueue offer them.  Tickets of one queue finish in issue order."""
    fifo = []

    def reap(block):
        while fifo:
            st = q.poll([fifo[0][0]])[0]
            if st == 0:
                if not block:
                    return
                st = q.wait(fifo[0][0], 5_000_000)
            assert st == 1, ("ticket state", st)
            fifo.pop(0)[1]()
    for submit, check in calls:
        while True:
            while len(fifo) >= width:
                reap(True)
            rc, tk = submit()
            if rc == -7:  # CIRCL_HIP_EAGAIN: every device batch busy
                reap(True)
                continue
            assert rc == 0, ("submit", rc)
            fifo.append((tk, check))
            reap(False)
            break
    while fifo:
        reap(True)


rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60)
it = 0
while time.time() < t_end:
    it += 1
    p = int(rng.choice([512, 768, 1024]))
    n = int(rng.choice([1, 2, 7, 63, 64, 65, 1000, 4097, 30000]))
    seeds = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    ek, dk = hostapi.mlkem_keygen(p, seeds)
    ek0, dk0 = orc.mlkem_keygen(p, seeds)
    assert (ek == ek0).all() and (dk == dk0).all(), ("kem keygen", p, n)
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ct, ss, st = hostapi.mlkem_encaps(p, ek, m)
    ct0, ss0, _ = orc.mlkem_encaps(p, ek, m)
    assert (ct == ct0).all() and (ss == ss0).all() and not st.any(), ("encaps", p, n)
    ct[:: 5, 3] ^= 1
    ss2, st2 = hostapi.mlkem_decaps(p, dk, ct)
    ss20, _ = orc.mlkem_decaps(p, dk, ct)
    assert (ss2 == ss20).all(), ("decaps", p, n)
    cts, sss, _ = hostapi.mlkem_encaps_shared(p, ek[:1], m)
    ct1, ss1, _ = orc.mlkem_encaps(p, np.tile(ek[:1], (n, 1)), m)
    assert (cts == ct1).all() and (sss == ss1).all(), ("encaps shared", p, n)
    ssd, _ = hostapi.mlkem_decaps_shared(p, dk[:1], cts)
    assert (ssd == sss).all(), ("decaps shared", p, n)
    nk = int(rng.integers(1, min(n, 50) + 1))
    idx = rng.integers(0, nk, n).astype(np.uint32)
    ctk, ssk, stk = hostapi.mlkem_encaps_keyed(p, ek[:nk], idx, m)
    ctg, ssg, _ = orc.mlkem_encaps(p, ek[idx], m)
    assert (ctk == ctg).all() and (ssk == ssg).all() and not stk.any(), ("encaps keyed", p, n, nk)
    ssdk, stdk = hostapi.mlkem_decaps_keyed(p, dk[:nk], idx, ctk)
    assert (ssdk == ssk).all() and not stdk.any(), ("decaps keyed", p, n, nk)
    # the asynchronous form over the same answers: resident tables (random batch limit, random call sizes, a window of tickets) and a call queue
    na = min(n, 160)
    mi = int(rng.choice([8, 64, 512]))
    pub, prv = hostapi.KeyTable("mlkem-public", p, ek[:nk]), hostapi.KeyTable("mlkem-private", p, dk[:nk])
    pub.async_start(mi, 0)
    prv.async_start(mi, int(rng.choice([0, 40])))
    qe = hostapi.CallQueue("mlkem-encaps", p, mi)
    CTB = ctk.shape[1]
    calls, lo = [], 0
    while lo < na:
        c = int(min(na - lo, rng.integers(1, mi // 4 + 1)))
        a, b = lo, lo + c
        oc, os_, ost = np.full((c, CTB), 0xAA, np.uint8), np.full((c, 32), 0xAA, np.uint8), np.full(c, 0xAA, np.uint8)
        od, odst = np.full((c, 32), 0xAA, np.uint8), np.full(c, 0xAA, np.uint8)
        qc, qs, qst = np.full((c, CTB), 0xAA, np.uint8), np.full((c, 32), 0xAA, np.uint8), np.full(c, 0xAA, np.uint8)
        calls.append((lambda a=a, b=b, oc=oc, os_=os_, ost=ost: pub.submit_encaps(m[a:b], oc, os_, ost, key_idx=idx[a:b]),
                      lambda a=a, b=b, oc=oc, os_=os_, ost=ost: (_same(oc, ctk[a:b], "async encaps"), _same(os_, ssk[a:b], "async encaps ss"), _same(ost, stk[a:b], "st"))))
        calls.append((lambda a=a, b=b, od=od, odst=odst: prv.submit_decaps(ctk[a:b], od, odst, key_idx=idx[a:b]),
                      lambda a=a, b=b, od=od, odst=odst: (_same(od, ssk[a:b], "async decaps"), _same(odst, stdk[a:b], "st"))))
        lo = b
    drive(pub, calls[0::2], int(rng.choice([1, 4, 16])))
    drive(prv, calls[1::2], int(rng.choice([1, 4, 16])))
    qcalls, lo = [], 0
    while lo < na:
        c = int(min(na - lo, rng.integers(1, mi // 4 + 1)))
        a, b = lo, lo + c
        qc, qs, qst = np.full((c, CTB), 0xAA, np.uint8), np.full((c, 32), 0xAA, np.uint8), np.full(c, 0xAA, np.uint8)
        qcalls.append((lambda a=a, b=b, qc=qc, qs=qs, qst=qst: qe.submit(ek[a:b], m[a:b], qc, qs, qst),
                       lambda a=a, b=b, qc=qc, qs=qs: (_same(qc, ct0[a:b], "queue encaps"), _same(qs, ss0[a:b], "queue encaps ss"))))
        lo = b
    drive(qe, qcalls, int(rng.choice([2, 8])))
    assert pub.try_close() == 0 and prv.try_close() == 0 and qe.close() == 0, "close"

    d = int(rng.choice([44, 65, 87, 2, 3, 5]))
    n = int(rng.choice([1, 3, 15, 16, 17, 100, 513, 1025, 3000, 9000]))
    s32 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk, sk = hostapi.mldsa_keygen(d, s32)
    pk0, sk0 = orc.mldsa_keygen(d, s32)
    assert (pk == pk0).all() and (sk == sk0).all(), ("dsa keygen", d, n)
    msgs = [rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8).tobytes() for _ in range(n)]
    sig = hostapi.mldsa_sign(d, sk, msgs)
    assert (sig == orc.mldsa_sign(d, sk, msgs)).all(), ("sign", d, n)
    sig[:: 4, 40] ^= 8
    ok = hostapi.mldsa_verify(d, pk, sig, msgs)
    assert ok.tolist() == orc.mldsa_verify(d, pk, sig, msgs).tolist(), ("verify", d, n)
    sg = hostapi.mldsa_sign_shared(d, sk[:1], msgs)
    assert (sg == orc.mldsa_sign(d, np.tile(sk[:1], (n, 1)), msgs)).all(), ("sign shared", d, n)
    assert hostapi.mldsa_verify_shared(d, pk[:1], sg, msgs).all(), ("verify shared", d, n)
    nk = int(rng.integers(1, min(n, 30) + 1))
    idx = rng.integers(0, nk, n).astype(np.uint32)
    okk = hostapi.mldsa_verify_keyed(d, pk[:nk], idx, sig, msgs)
    assert okk.tolist() == orc.mldsa_verify(d, pk[idx], sig, msgs).tolist(), ("verify keyed", d, n, nk)
    if d in (44, 65, 87):  # (resident tables are ML-DSA's) the asynchronous verification over the same verdicts, ragged messages
        na = min(n, 60)
        ver = hostapi.KeyTable("mldsa-public", d, pk[:nk])
        mi = int(rng.choice([8, 32]))
        ver.async_start(mi, 0)
        vcalls, lo = [], 0
        while lo < na:
            c = int(min(na - lo, rng.integers(1, mi // 4 + 1)))
            a, b = lo, lo + c
            ov = np.full(c, 0xAA, np.uint8)
            vcalls.append((lambda a=a, b=b, ov=ov: ver.submit_verify(sig[a:b], msgs[a:b], ov, key_idx=idx[a:b]),
                           lambda a=a, b=b, ov=ov: _same(ov, okk[a:b], "async verify")))
            lo = b
        drive(ver, vcalls, int(rng.choice([1, 6])))
        assert ver.try_close() == 0
    # X25519 and the hybrid KEMs (device composition, pair kernel, fixed-base comb)
    n = int(rng.choice([1, 2, 63, 64, 65, 700, 5000]))
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    u = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    o, ok = hostapi.x25519(k, u)
    o0, ok0 = orc.x25519(k, u)
    assert (o == o0).all() and (ok == ok0).all(), ("x25519", n)
    assert (hostapi.x25519(k)[0] == orc.x25519(k)[0]).all(), ("x25519 base", n)
    sch = int(rng.choice([1, 2, 3, 4]))
    S = hostapi.HYBRID_SIZES[sch]
    n = int(rng.choice([1, 3, 65, 400]))
    seeds = rng.integers(0, 256, (n, S["seed"]), dtype=np.uint8)
    es = rng.integers(0, 256, (n, S["eseed"]), dtype=np.uint8)
    pk, sk = hostapi.hybrid_keygen(sch, seeds)
    if sch == 1:
        pk0, sk0 = ohyb.xwing_keygen(seeds)[:2]
        ct0, ss0, st0 = ohyb.xwing_encaps(pk0, es)
    else:
        pk0, sk0 = ohyb.hybrid_keygen(seeds, sch)
        ct0, ss0, st0 = ohyb.hybrid_encaps(pk0, es, sch)
    assert (pk == pk0).all() and (sk == sk0).all(), ("hybrid keygen", sch, n)
    ct, ss, st = hostapi.hybrid_encaps(sch, pk, es)
    assert (ct == ct0).all() and (ss == ss0).all() and (st == st0).all(), ("hybrid encaps", sch, n)
    ss2, st2 = hostapi.hybrid_decaps(sch, sk, ct)
    assert (ss2 == ss).all() and not st2.any(), ("hybrid decaps", sch, n)
    if it % 7 == 0:  # the two-stream split of large signing batches, against the one-key oracle on a sample
        d = int(rng.choice([44, 65, 87]))
        n = (1 << 16) + int(rng.integers(0, 300))
        s32 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        pkd, skd = hostapi.mldsa_keygen(d, s32)
        msgs = [bytes(rng.integers(0, 256, 20, dtype=np.uint8)) for _ in range(n)]
        sig = hostapi.mldsa_sign(d, skd, msgs)
        assert hostapi.mldsa_verify(d, pkd, sig, msgs).all(), ("big sign verify", d, n)
        idx = rng.choice(n, 200, replace=False)
        assert (sig[idx] == orc.mldsa_sign(d, skd[idx], [msgs[i] for i in idx])).all(), ("big sign oracle", d, n)
print("stress ok:", it, "iterations")
