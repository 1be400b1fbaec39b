"""The ASYNCHRONOUS table API (circl_hip_keytable_async_start / *_table_submit / circl_hip_poll / circl_hip_wait / _eventfd): nobody sleeps per
call -- submitters copy their rows in and get a ticket, ONE library thread per queue launches the batches and writes the results straight
into the submitters' buffers.  It serves the callers of kem.Scheme.Encapsulate / Decapsulate (kem/mlkem/mlkem768/kyber.go:347-386,
hpke/algs.go:283-285) and sign.Scheme.Verify (sign/mldsa/mldsa65/dilithium.go:305) from ONE goroutine per device instead of one blocked OS
thread per call.  Whatever batch a submitted call ends up in, its bytes must be the oracle's -- checked call after call, from two threads
that interleave submit and poll, including a bad decapsulation-key entry, an out-of-range index and a NULL required input."""
import os
import select
import threading

import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu

EPARAM, EBUSY, EAGAIN = -1, -6, -7


def _threads(T, body):
    errs = []
    gate = threading.Barrier(T)

    def wrap(t):
        try:
            gate.wait()
            body(t)
        except Exception as e:  # noqa: BLE001
            import traceback
            errs.append((t, repr(e), traceback.format_exc()[-600:]))
    th = [threading.Thread(target=wrap, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs[:2]


class Window:
    """A reactor's outstanding requests: FIFO of (ticket, check) -- tickets of one queue complete in issue order."""

    def __init__(self, table, width):
        self.table, self.width, self.fifo, self.done = table, width, [], 0

    def room(self):
        return len(self.fifo) < self.width

    def push(self, ticket, check):
        self.fifo.append((ticket, check))

    def reap(self, block=False):
        while self.fifo:
            st = self.table.poll([self.fifo[0][0]])[0]
            if st == 0:
                if not block:
                    return
                st = self.table.wait(self.fifo[0][0], 2_000_000)
                assert st != 0, "a ticket did not complete within 2 s"
            assert st == 1, st
            self.fifo.pop(0)[1]()
            self.done += 1


@pytest.mark.parametrize("param,max_items,wait_us", [(768, 256, 0), (1024, 64, 150), (512, 8, 0)])
def test_mlkem_submit_poll_from_two_threads_equals_the_oracle(param, max_items, wait_us):
    from circl_amd import hostapi
    rng = np.random.default_rng(param)
    nkeys, pool = 5, 500
    ek, dk = orc.mlkem_keygen(param, rng.integers(0, 256, (nkeys, 64), dtype=np.uint8))
    dk_bad = dk.copy()
    dk_bad[3, -40] ^= 1                                     # entry 3: H(ek) inside dk no longer matches -> kem.ErrPrivKey (status 2) per item
    m = rng.integers(0, 256, (pool, 32), dtype=np.uint8)
    idx = rng.integers(0, nkeys, pool).astype(np.uint32)
    ct0, ss0, _ = orc.mlkem_encaps(param, ek[idx], m)
    ct_in = ct0.copy()
    ct_in[::3, 11] ^= 2                                     # implicit rejection for every third item
    ssd0, std0 = orc.mlkem_decaps(param, dk_bad[idx], ct_in)
    assert (std0[idx == 3] == 2).all() and (std0[idx != 3] == 0).all()
    CT = ct0.shape[1]
    pub = hostapi.KeyTable("mlkem-public", param, ek)
    prv = hostapi.KeyTable("mlkem-private", param, dk_bad)
    assert list(prv.key_status) == [0, 0, 0, 2, 0]
    pub.async_start(max_items, wait_us)
    prv.async_start(max_items, wait_us, eventfd=True)
    assert pub.eventfd() == -1 and prv.eventfd() >= 0

    # what submit rejects at once (nothing joins a batch): an index past the table, a NULL required input, more items than a call may hold
    one = np.zeros((1, CT), np.uint8), np.zeros((1, 32), np.uint8), np.zeros(1, np.uint8)
    rc, _ = pub.submit_encaps(m[:1], *one, key_idx=[nkeys])
    assert rc == EPARAM
    L = hostapi.nat.lib()
    import ctypes as C
    t = C.c_uint64()
    assert L.circl_hip_mlkem_encaps_table_submit(pub.handle, None, None, hostapi._p(one[0]), hostapi._p(one[1]), hostapi._p(one[2]), 1, C.byref(t)) == EPARAM
    assert L.circl_hip_mlkem_decaps_table_submit(prv.handle, None, None, hostapi._p(one[1]), hostapi._p(one[2]), 1, C.byref(t)) == EPARAM
    big = max_items // 4 + 1
    rc, _ = pub.submit_encaps(m[:big], np.zeros((big, CT), np.uint8), np.zeros((big, 32), np.uint8), np.zeros(big, np.uint8), key_idx=idx[:big])
    assert rc == EPARAM
    rc, _ = prv.submit_encaps(m[:1], *one)                  # the wrong kind of table
    assert rc == EPARAM
    # ... and the blocking call on the same tables says the same about a NULL input (it used to encapsulate to an all-zero seed)
    assert L.circl_hip_mlkem_encaps_table(pub.handle, None, None, hostapi._p(one[0]), hostapi._p(one[1]), hostapi._p(one[2]), 1) == EPARAM

    eagain = [0, 0]

    def body(t):
        r = np.random.default_rng(77 + t)
        we, wd = Window(pub, 24), Window(prv, 24)
        for _ in range(120):
            n = int(r.choice([1, 1, 1, 2, max(1, max_items // 4)]))
            lo = int(r.integers(0, pool - n))
            if we.room():
                ct, ss, st = np.full((n, CT), 0xAA, np.uint8), np.full((n, 32), 0xAA, np.uint8), np.full(n, 0xAA, np.uint8)
                rc, tk = pub.submit_encaps(m[lo:lo + n], ct, ss, st, key_idx=idx[lo:lo + n])
                if rc == EAGAIN:
                    eagain[t] += 1
                else:
                    assert rc == 0 and tk != 0, rc

                    def chk(ct=ct, ss=ss, st=st, lo=lo, n=n):
                        assert (st == 0).all() and (ct == ct0[lo:lo + n]).all() and (ss == ss0[lo:lo + n]).all(), ("encaps", t, lo, n)
                    we.push(tk, chk)
            if wd.room():
                ss, st = np.full((n, 32), 0xAA, np.uint8), np.full(n, 0xAA, np.uint8)
                rc, tk = prv.submit_decaps(ct_in[lo:lo + n], ss, st, key_idx=idx[lo:lo + n])
                if rc == EAGAIN:
                    eagain[t] += 1
                else:
                    assert rc == 0, rc

                    def chk(ss=ss, st=st, lo=lo, n=n):
                        assert (st == std0[lo:lo + n]).all() and (ss == ssd0[lo:lo + n]).all(), ("decaps", t, lo, n)
                    wd.push(tk, chk)
            we.reap()
            wd.reap()
        we.reap(block=True)
        wd.reap(block=True)
        assert we.done > 20 and wd.done > 20

    _threads(2, body)
    # a BLOCKING call through a table with a queue: submit + wait inside the library, same bytes
    ct, ss, st = pub.encaps(m[:3], idx[:3])
    assert (ct == ct0[:3]).all() and (ss == ss0[:3]).all() and not st.any()
    # the eventfd counted the private table's batches
    fd = prv.eventfd()
    r, _, _ = select.select([fd], [], [], 0)
    assert r and int.from_bytes(os.read(fd, 8), "little") >= 1
    calls, items, launches = prv.coalesce_stats()
    assert calls >= 40 and items >= calls and 1 <= launches <= calls
    assert pub.poll([0])[0] == 1                            # ticket 0 (an empty call) is always done
    assert pub.poll([1 << 40])[0] == EPARAM                 # a ticket this queue never issued
    assert pub.async_stop() == 0 and prv.try_close() == 0
    rc, _ = pub.submit_encaps(m[:1], *one)                  # no queue any more
    assert rc == EPARAM
    pub.close()


@pytest.mark.parametrize("param", [44, 65, 87])
def test_mldsa_verify_submit_equals_the_oracle(param):
    from circl_amd import hostapi
    rng = np.random.default_rng(param)
    nkeys, pool = 3, 60
    pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (nkeys, 32), dtype=np.uint8))
    idx = rng.integers(0, nkeys, pool).astype(np.uint32)
    msgs = [bytes(rng.integers(0, 256, int(rng.integers(0, 90)), dtype=np.uint8)) for _ in range(pool)]
    ctxs = [bytes(rng.integers(0, 256, int(rng.integers(0, 20)), dtype=np.uint8)) for _ in range(pool)]
    sig = orc.mldsa_sign(param, sk[idx], msgs, ctxs=ctxs)
    sig[::4, 50] ^= 1
    ok0 = orc.mldsa_verify(param, pk[idx], sig, msgs, ctxs=ctxs)
    assert ok0.any() and not ok0.all()
    tab = hostapi.KeyTable("mldsa-public", param, pk)
    tab.async_start(64, 0)
    w = Window(tab, 16)
    rng2 = np.random.default_rng(5)
    for _ in range(40):
        n = int(rng2.choice([1, 1, 2, 5]))
        lo = int(rng2.integers(0, pool - n))
        while not w.room():
            w.reap(block=True)
        ok = np.full(n, 0xAA, np.uint8)
        rc, tk = tab.submit_verify(sig[lo:lo + n], msgs[lo:lo + n], ok, ctxs=ctxs[lo:lo + n], key_idx=idx[lo:lo + n])
        if rc == EAGAIN:
            w.reap(block=True)
            continue
        assert rc == 0, rc

        def chk(ok=ok, lo=lo, n=n):
            assert (ok == ok0[lo:lo + n]).all(), (lo, n, ok, ok0[lo:lo + n])
        w.push(tk, chk)
        w.reap()
    w.reap(block=True)
    assert w.done >= 30
    # a context longer than 255 bytes is the reference's `return false` (sign/mldsa/mldsa65/dilithium.go:115-118), not an error: the
    # item's verdict is 0, as through the blocking call
    good = int(np.flatnonzero(ok0)[0])
    ok = np.full(1, 0xAA, np.uint8)
    rc, tk = tab.submit_verify(sig[good:good + 1], msgs[good:good + 1], ok, ctxs=[b"x" * 256], key_idx=idx[good:good + 1])
    assert rc == 0 and tab.wait(tk, 2_000_000) == 1 and ok[0] == 0
    assert tab.try_close() == 0


def test_setters_refuse_while_calls_are_in_flight_and_nothing_is_freed_under_a_caller():
    """circl_hip_keytable_set_coalesce on a live table used to free its coalescer under the callers.  Now it returns
    CIRCL_HIP_EBUSY (and changes nothing) while calls are inside; 8 threads call while the main thread toggles, every result stays right."""
    from circl_amd import hostapi
    rng = np.random.default_rng(9)
    ek, _ = orc.mlkem_keygen(768, rng.integers(0, 256, (2, 64), dtype=np.uint8))
    m = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    idx = (np.arange(64) % 2).astype(np.uint32)
    ct0, ss0, _ = orc.mlkem_encaps(768, ek[idx], m)
    tab = hostapi.KeyTable("mlkem-public", 768, ek)
    tab.set_coalesce(64)
    stop = threading.Event()
    seen = {"busy": 0, "ok": 0}

    def body(t):
        if t == 0:
            L = hostapi.nat.lib()
            for k in range(300):
                rc = L.circl_hip_keytable_set_coalesce(tab.handle, 64 if k % 2 else 0, 0)
                assert rc in (0, EBUSY), rc
                seen["busy" if rc else "ok"] += 1
            stop.set()
            return
        r = np.random.default_rng(t)
        while not stop.is_set():
            lo = int(r.integers(0, 63))
            ct, ss, st = tab.encaps(m[lo:lo + 1], idx[lo:lo + 1])
            assert (ct == ct0[lo:lo + 1]).all() and (ss == ss0[lo:lo + 1]).all() and not st.any()

    _threads(9, body)
    assert seen["ok"] + seen["busy"] == 300
    tab.close()


@pytest.mark.parametrize("scheme_name", ["XWING", "X25519MLKEM768"])
def test_hybrid_tables_submit_and_coalesce_equal_the_plain_calls(scheme_name):
    """The hybrid KEM tables (kem/xwing/xwing.go:259,288; kem/hybrid/hybrid.go:95-99) through the asynchronous queue and through blocking
    coalescing: the bytes of circl_hip_hybrid_encaps / _decaps made alone (those are checked against the oracle by tests/test_gpu_hybrid.py).
    One hybrid launch runs further kernels BEHIND the ML-KEM table call inside it: its batches must not take the tail-flag offer (the flag
    would go up before the X25519 ladder and the combiner ran) -- a premature completion would show here as wrong shared secrets."""
    from circl_amd import hostapi
    scheme = getattr(hostapi, scheme_name)
    S = hostapi.HYBRID_SIZES[scheme]
    rng = np.random.default_rng(scheme)
    nkeys, pool = 3, 48
    pk, sk = hostapi.hybrid_keygen(scheme, rng.integers(0, 256, (nkeys, S["seed"]), dtype=np.uint8))
    idx = rng.integers(0, nkeys, pool).astype(np.uint32)
    es = rng.integers(0, 256, (pool, S["eseed"]), dtype=np.uint8)
    ct0, ss0, st0 = hostapi.hybrid_encaps(scheme, pk[idx], es)
    assert not st0.any()
    ct_in = ct0.copy()
    ct_in[::5, 7] ^= 4                                       # implicit rejection on the lattice half for every fifth item
    ssd0, std0 = hostapi.hybrid_decaps(scheme, sk[idx], ct_in)
    pub = hostapi.KeyTable("hybrid-public", scheme, pk)
    prv = hostapi.KeyTable("hybrid-private", scheme, sk)
    # ---- asynchronous ----
    pub.async_start(32, 0)
    prv.async_start(32, 0)
    wp, wq = Window(pub, 6), Window(prv, 6)
    r = np.random.default_rng(1)
    for k in range(30):
        n = int(r.choice([1, 1, 2, 4]))
        lo = int(r.integers(0, pool - n))
        w, enc = (wp, True) if k % 2 == 0 else (wq, False)
        while not w.room():
            w.reap(block=True)
        ss, st = np.full((n, S["ss"]), 0xAA, np.uint8), np.full(n, 0xAA, np.uint8)
        if enc:
            ct = np.full((n, S["ct"]), 0xAA, np.uint8)
            rc, tk = pub.submit_hybrid_encaps(es[lo:lo + n], ct, ss, st, key_idx=idx[lo:lo + n])
        else:
            ct = None
            rc, tk = prv.submit_hybrid_decaps(ct_in[lo:lo + n], ss, st, key_idx=idx[lo:lo + n])
        if rc == EAGAIN:
            w.reap(block=True)
            continue
        assert rc == 0, rc

        def chk(enc=enc, ct=ct, ss=ss, st=st, lo=lo, n=n):
            if enc:
                assert (ct == ct0[lo:lo + n]).all() and (ss == ss0[lo:lo + n]).all() and not st.any(), (lo, n)
            else:
                assert (ss == ssd0[lo:lo + n]).all() and (st == std0[lo:lo + n]).all(), (lo, n)
        w.push(tk, chk)
        w.reap()
    wp.reap(block=True)
    wq.reap(block=True)
    assert wp.done + wq.done >= 20
    assert pub.async_stop() == 0 and prv.async_stop() == 0
    # ---- blocking, coalesced across four threads ----
    pub.set_coalesce(16)
    prv.set_coalesce(16)

    def body(t):
        rr = np.random.default_rng(100 + t)
        for _ in range(6):
            lo = int(rr.integers(0, pool - 2))
            ct, ss, st = pub.hybrid_encaps(es[lo:lo + 2], idx[lo:lo + 2])
            assert (ct == ct0[lo:lo + 2]).all() and (ss == ss0[lo:lo + 2]).all() and not st.any()
            ss2, st2 = prv.hybrid_decaps(ct_in[lo:lo + 1], idx[lo:lo + 1])
            assert (ss2 == ssd0[lo:lo + 1]).all() and (st2 == std0[lo:lo + 1]).all()
    _threads(4, body)
    calls, items, launches = pub.coalesce_stats()
    assert calls == 24 and items == 48 and 1 <= launches <= calls
    pub.close()
    prv.close()


def test_call_queues_equal_the_oracle_and_the_plain_calls():
    """circl_hip_queue: keys that come WITH the call (a TLS 1.3 server encapsulates to the client's ephemeral key: kem/hybrid/hybrid.go:271-300 ->
    kem/mlkem/mlkem768/kyber.go:359-370).  ML-KEM-768 encapsulation and decapsulation against the oracle -- every item under its own key, a
    non-canonical public key (kem.ErrPubKey, status 1) and a private key whose stored hash does not match (kem.ErrPrivKey, status 2) among them --
    and X25519MLKEM768 encapsulation against circl_hip_hybrid_encaps; two threads submit and poll each queue."""
    from circl_amd import hostapi
    rng = np.random.default_rng(77)
    pool = 120
    ek, dk = orc.mlkem_keygen(768, rng.integers(0, 256, (pool, 64), dtype=np.uint8))
    ek[5, 0:2] = 0xFF                                        # coefficient 0 = 0xFFF >= q: UnpackMLKEM rejects it (cpapke.go:45-55)
    dk[9, -40] ^= 1                                          # H(ek) inside dk no longer matches
    m = rng.integers(0, 256, (pool, 32), dtype=np.uint8)
    ct0, ss0, st0 = orc.mlkem_encaps(768, ek, m)
    assert st0[5] == 1 and st0.sum() == 1
    ct_in = ct0.copy()
    ct_in[::4, 20] ^= 8
    ssd0, std0 = orc.mlkem_decaps(768, dk, ct_in)
    assert std0[9] == 2
    S = hostapi.HYBRID_SIZES[hostapi.X25519MLKEM768]
    hpk, _hsk = hostapi.hybrid_keygen(hostapi.X25519MLKEM768, rng.integers(0, 256, (pool, S["seed"]), dtype=np.uint8))
    hes = rng.integers(0, 256, (pool, S["eseed"]), dtype=np.uint8)
    hct0, hss0, hst0 = hostapi.hybrid_encaps(hostapi.X25519MLKEM768, hpk, hes)
    qe = hostapi.CallQueue("mlkem-encaps", 768, 64)
    qd = hostapi.CallQueue("mlkem-decaps", 768, 32, eventfd=True)
    qh = hostapi.CallQueue("hybrid-encaps", hostapi.X25519MLKEM768, 32)
    assert qd.eventfd() >= 0 and qe.eventfd() == -1

    def body(t):
        r = np.random.default_rng(t)
        we, wd, wh = Window(qe, 8), Window(qd, 8), Window(qh, 4)
        for k in range(45):
            n = int(r.choice([1, 1, 2, 3]))
            lo = int(r.integers(0, pool - n))
            kind = k % 3
            w = (we, wd, wh)[kind]
            while not w.room():
                w.reap(block=True)
            if kind == 0:
                ct, ss, st = np.full((n, 1088), 0xAA, np.uint8), np.full((n, 32), 0xAA, np.uint8), np.full(n, 0xAA, np.uint8)
                rc, tk = qe.submit(ek[lo:lo + n], m[lo:lo + n], ct, ss, st)
                chk = lambda ct=ct, ss=ss, st=st, lo=lo, n=n: (_eq(ct, ct0[lo:lo + n]), _eq(ss, ss0[lo:lo + n]), _eq(st, st0[lo:lo + n]))
            elif kind == 1:
                ss, st = np.full((n, 32), 0xAA, np.uint8), np.full(n, 0xAA, np.uint8)
                rc, tk = qd.submit(dk[lo:lo + n], ct_in[lo:lo + n], None, ss, st)
                chk = lambda ss=ss, st=st, lo=lo, n=n: (_eq(ss, ssd0[lo:lo + n]), _eq(st, std0[lo:lo + n]))
            else:
                ct, ss = np.full((n, S["ct"]), 0xAA, np.uint8), np.full((n, S["ss"]), 0xAA, np.uint8)
                rc, tk = qh.submit(hpk[lo:lo + n], hes[lo:lo + n], ct, ss, None)   # (status is optional)
                chk = lambda ct=ct, ss=ss, lo=lo, n=n: (_eq(ct, hct0[lo:lo + n]), _eq(ss, hss0[lo:lo + n]))
            if rc == EAGAIN:
                w.reap(block=True)
                continue
            assert rc == 0, rc
            w.push(tk, chk)
            w.reap()
        for w in (we, wd, wh):
            w.reap(block=True)
        assert we.done + wd.done + wh.done >= 30
    _threads(2, body)
    calls, items, launches = qe.stats()
    assert calls >= 20 and items >= calls and 1 <= launches <= calls
    ss = np.zeros((1, 32), np.uint8)
    rc, _ = qd.submit(dk[:1], ct_in[:1], None, ss, None)     # left outstanding: close finishes it
    assert rc == 0
    assert qe.poll([1 << 40])[0] == EPARAM                   # a ticket this queue never issued
    rc, _ = qe.submit(ek[:40], m[:40], np.zeros((40, 1088), np.uint8), np.zeros((40, 32), np.uint8), None)
    assert rc == EPARAM                                      # more items than a submitted call may hold (max_items / 4)
    assert qe.close() == 0 and qd.close() == 0 and qh.close() == 0   # (qd finishes its outstanding ticket first)
    assert (ss == ssd0[:1]).all()


def _eq(a, b):
    assert (a == b).all(), (a, b)
