"""Many concurrent ONE-ITEM callers through one resident key table -- the shape of every consumer of the reference's kem.Scheme /
sign.Scheme (kem/hybrid/hybrid.go:95-99, kem/xwing/xwing.go:259,288, hpke/algs.go:283-285, kem/mlkem/mlkem768/kyber.go:347-386,
sign/mldsa/mldsa65/dilithium.go:305): circl_hip_keytable_set_coalesce merges their calls into shared launches.  Whatever batch a
call ends up in, its bytes must be the oracle's: T threads x random call sizes, checked call by call."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_threads(T, body):
    errs = []
    gate = threading.Barrier(T)

    def wrap(t):
        try:
            gate.wait()
            body(t)
        except Exception as e:  # noqa: BLE001 -- reported below, with the thread's number
            errs.append((t, repr(e)))
    th = [threading.Thread(target=wrap, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs[:3]


@pytest.mark.parametrize("param,max_items,wait_us", [(768, 256, 0), (1024, 64, 200), (512, 8, 0)])
def test_mlkem_coalesced_calls_equal_the_oracle_call_after_call(param, max_items, wait_us):
    from circl_amd import hostapi
    rng = np.random.default_rng(param + max_items)
    nkeys, pool, T, rounds = 7, 600, 12, 25
    ek, dk = orc.mlkem_keygen(param, rng.integers(0, 256, (nkeys, 64), dtype=np.uint8))
    m = rng.integers(0, 256, (pool, 32), dtype=np.uint8)
    idx = rng.integers(0, nkeys, pool).astype(np.uint32)
    ct0, ss0, _ = orc.mlkem_encaps(param, ek[idx], m)
    ct_bad = ct0.copy()
    ct_bad[::3, 17] ^= 4                                   # implicit rejection for every third item
    ssd0, _ = orc.mlkem_decaps(param, dk[idx], ct_bad)
    pub = hostapi.KeyTable("mlkem-public", param, ek)
    prv = hostapi.KeyTable("mlkem-private", param, dk)
    pub.set_coalesce(max_items, wait_us)
    prv.set_coalesce(max_items, wait_us)

    def body(t):
        r = np.random.default_rng(1000 + t)
        for _ in range(rounds):
            n = int(r.choice([1, 1, 1, 2, 3, max(1, max_items // 4), max_items // 4 + 1, 90]))   # joins a batch / too big: its own call
            lo = int(r.integers(0, pool - n))
            ct, ss, st = pub.encaps(m[lo:lo + n], idx[lo:lo + n])
            assert (st == 0).all() and (ct == ct0[lo:lo + n]).all() and (ss == ss0[lo:lo + n]).all(), (t, n, lo)
            got, st = prv.decaps(ct_bad[lo:lo + n], idx[lo:lo + n])
            assert (st == 0).all() and (got == ssd0[lo:lo + n]).all(), (t, n, lo)
    _run_threads(T, body)
    calls, items, launches = pub.coalesce_stats()
    assert calls > 0 and items >= calls and 0 < launches <= calls
    # an absent index vector = entry 0 for every item, alone and next to callers that bring one
    ct1, ss1, _ = pub.encaps(m[:3])
    ct10, ss10, _ = orc.mlkem_encaps(param, np.tile(ek[:1], (3, 1)), m[:3])
    assert (ct1 == ct10).all() and (ss1 == ss10).all()
    # switching it off again: the same bytes through the ordinary path
    pub.set_coalesce(0)
    ct2, ss2, _ = pub.encaps(m[:5], idx[:5])
    assert (ct2 == ct0[:5]).all() and (ss2 == ss0[:5]).all()
    with pytest.raises(Exception):                           # an index beyond the table is still refused
        prv.decaps(ct_bad[:2], np.array([0, nkeys], np.uint32))
    pub.close()
    prv.close()


def test_a_private_table_reports_its_bad_entry_to_every_coalesced_caller():
    from circl_amd import hostapi
    rng = np.random.default_rng(77)
    ek, dk = orc.mlkem_keygen(768, rng.integers(0, 256, (4, 64), dtype=np.uint8))
    dk_bad = dk.copy()
    dk_bad[2, -40] ^= 1                                     # entry 2: stored H(ek) no longer matches (kem.ErrPrivKey)
    m = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    idx = (np.arange(40) % 4).astype(np.uint32)
    ct, ss0, _ = orc.mlkem_encaps(768, ek[idx], m)
    prv = hostapi.KeyTable("mlkem-private", 768, dk_bad)
    prv.set_coalesce(32)

    def body(t):
        for i in range(t, 40, 8):
            got, st = prv.decaps(ct[i:i + 1], idx[i:i + 1])
            if idx[i] == 2:
                assert st[0] == 2 and not got.any()
            else:
                assert st[0] == 0 and (got[0] == ss0[i]).all()
    _run_threads(8, body)
    prv.close()


@pytest.mark.parametrize("param", [65, 44])
def test_mldsa_coalesced_verifications_equal_the_oracle(param):
    from circl_amd import hostapi
    rng = np.random.default_rng(param)
    nkeys, pool, T = 3, 60, 8
    pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (nkeys, 32), dtype=np.uint8))
    idx = rng.integers(0, nkeys, pool).astype(np.uint32)
    msgs = [bytes(rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8)) for _ in range(pool)]
    ctxs = [bytes(rng.integers(0, 256, int(rng.integers(0, 20)), dtype=np.uint8)) for _ in range(pool)]
    sig = orc.mldsa_sign(param, sk[idx], msgs, ctxs)
    sig[::4, 100] ^= 1                                      # a quarter of the signatures are bad
    want = orc.mldsa_verify(param, pk[idx], sig, msgs, ctxs)
    assert 0 < want.sum() < pool
    tab = hostapi.KeyTable("mldsa-public", param, pk)
    tab.set_coalesce(64)

    def body(t):
        r = np.random.default_rng(t)
        for _ in range(10):
            n = int(r.choice([1, 1, 2, 5]))
            lo = int(r.integers(0, pool - n))
            ok = tab.verify(sig[lo:lo + n], msgs[lo:lo + n], ctxs[lo:lo + n], idx[lo:lo + n])
            assert (ok == want[lo:lo + n]).all(), (t, n, lo)
            if n == 1:                                       # a caller without contexts next to callers with them
                ok = tab.verify(sig[lo:lo + 1], msgs[lo:lo + 1], None, idx[lo:lo + 1])
                assert ok[0] == (1 if (len(ctxs[lo]) == 0 and want[lo]) else 0)
    _run_threads(T, body)
    # the signer of the same keys: one message per call from every thread, deterministic and hedged (rnd) next to each other
    signer = hostapi.KeyTable("mldsa-private", param, sk)
    signer.set_coalesce(32)
    rnd = rng.integers(0, 256, (pool, 32), dtype=np.uint8)
    want_det = orc.mldsa_sign(param, sk[idx], msgs, ctxs)
    want_rnd = orc.mldsa_sign(param, sk[idx], msgs, ctxs, rnd)

    def sign_body(t):
        r = np.random.default_rng(50 + t)
        for _ in range(6):
            n = int(r.choice([1, 1, 3]))
            lo = int(r.integers(0, pool - n))
            if r.integers(0, 2):
                got = signer.sign(msgs[lo:lo + n], ctxs[lo:lo + n], None, idx[lo:lo + n])
                assert (got == want_det[lo:lo + n]).all(), (t, n, lo)
            else:
                got = signer.sign(msgs[lo:lo + n], ctxs[lo:lo + n], rnd[lo:lo + n], idx[lo:lo + n])
                assert (got == want_rnd[lo:lo + n]).all(), (t, n, lo)
    _run_threads(T, sign_body)
    assert signer.coalesce_stats()[0] > 0
    with pytest.raises(Exception):                           # a context of 256 bytes is refused on the host, coalescing or not
        signer.sign(msgs[:1], [bytes(256)], None, idx[:1])
    signer.close()
    # a message too long for a shared batch takes the ordinary path
    long_msg = [bytes(rng.integers(0, 256, 300000, dtype=np.uint8))]
    s1 = orc.mldsa_sign(param, sk[:1], long_msg, [b""])
    assert tab.verify(s1, long_msg, [b""], np.zeros(1, np.uint32))[0] == 1
    tab.close()


def test_zero_copy_and_copied_small_calls_give_the_same_bytes():
    """A tiny host-buffer call lets its kernels read / write the page-locked staging areas themselves (no copy is enqueued);
    CIRCL_HIP_ZEROCOPY_KB=0 takes the copies.  Both in fresh processes (the knob is read once), every family, against the oracle."""
    code = r"""
import numpy as np
from circl_amd import hostapi
from oracle import orc
rng = np.random.default_rng(5)
for n in (1, 3, 40):
    ek, dk = orc.mlkem_keygen(768, rng.integers(0, 256, (n, 64), dtype=np.uint8))
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ct, ss, st = hostapi.mlkem_encaps(768, ek, m)
    ct0, ss0, _ = orc.mlkem_encaps(768, ek, m)
    assert (ct == ct0).all() and (ss == ss0).all() and not st.any()
    ssd, st = hostapi.mlkem_decaps(768, dk, ct)
    assert (ssd == ss0).all() and not st.any()
    pub, prv = hostapi.KeyTable("mlkem-public", 768, ek), hostapi.KeyTable("mlkem-private", 768, dk)
    idx = np.arange(n, dtype=np.uint32)[::-1].copy()
    ct1, ss1, _ = pub.encaps(m, idx)
    ct10, ss10, _ = orc.mlkem_encaps(768, ek[idx], m)
    assert (ct1 == ct10).all() and (ss1 == ss10).all()
    assert (prv.decaps(ct1, idx)[0] == ss10).all()
    pk, sk = orc.mldsa_keygen(65, rng.integers(0, 256, (n, 32), dtype=np.uint8))
    msgs = [bytes(rng.integers(0, 256, 1 + 13 * i, dtype=np.uint8)) for i in range(n)]
    sig = hostapi.mldsa_sign(65, sk, msgs)
    assert (sig == orc.mldsa_sign(65, sk, msgs, [b""] * n)).all()
    sig[0, 9] ^= 1
    ok = hostapi.mldsa_verify(65, pk, sig, msgs)
    assert ok.tolist() == [0] + [1] * (n - 1)
print("ok")
"""
    for kb in ("0", "64", "4096"):
        env = dict(os.environ, CIRCL_HIP_ZEROCOPY_KB=kb, PYTHONPATH=ROOT)
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "ok" in out.stdout, (kb, out.stdout[-2000:], out.stderr[-2000:])


def test_process_wide_coalescing_of_calls_that_bring_their_own_keys():
    """circl_hip_set_coalesce: the TLS-server shape -- every handshake encapsulates ONCE, to the peer's ephemeral key (kem/hybrid/hybrid.go:
    271-300 -> kem/mlkem/mlkem768/kyber.go:359-370), so there is no resident key to hang a batch on.  Threads calling the ordinary host
    entry points with one or a few items each share launches; every call's bytes are checked against the oracle.  In a process of its
    own (the switch is process-wide)."""
    code = r"""
import threading
import numpy as np
from circl_amd import hostapi, _native as nat
from oracle import orc, hybrid as ohyb
assert nat.lib().circl_hip_set_coalesce(128, 0) == 0
rng = np.random.default_rng(11)
pool = 96
ek, dk = orc.mlkem_keygen(768, rng.integers(0, 256, (pool, 64), dtype=np.uint8))
m = rng.integers(0, 256, (pool, 32), dtype=np.uint8)
ct0, ss0, _ = orc.mlkem_encaps(768, ek, m)
ct_bad = ct0.copy(); ct_bad[::3, 5] ^= 1
ssd0, _ = orc.mlkem_decaps(768, dk, ct_bad)
ek[7, 0] = 0xff; ek[7, 1] |= 0x0f                       # one non-canonical public key: its caller alone gets status 1 (kem.ErrPubKey)
pk, sk = orc.mldsa_keygen(65, rng.integers(0, 256, (pool, 32), dtype=np.uint8))
msgs = [bytes(rng.integers(0, 256, 1 + 3 * i, dtype=np.uint8)) for i in range(pool)]
ctxs = [bytes(rng.integers(0, 256, i % 9, dtype=np.uint8)) for i in range(pool)]
sig = orc.mldsa_sign(65, sk, msgs, ctxs)
sig[::5, 77] ^= 2
okv = orc.mldsa_verify(65, pk, sig, msgs, ctxs)
hs = hostapi.HYBRID_SIZES[hostapi.X25519MLKEM768]
hseed = rng.integers(0, 256, (16, hs["seed"]), dtype=np.uint8)
hpk, hsk = hostapi.hybrid_keygen(hostapi.X25519MLKEM768, hseed)
hes = rng.integers(0, 256, (16, hs["eseed"]), dtype=np.uint8)
hct0, hss0, _ = ohyb.hybrid_encaps(hpk, hes)
errs = []
def body(t):
    try:
        r = np.random.default_rng(t)
        for _ in range(12):
            n = int(r.choice([1, 1, 1, 2, 5]))
            lo = int(r.integers(0, pool - n))
            ct, ss, st = hostapi.mlkem_encaps(768, ek[lo:lo + n], m[lo:lo + n])
            for i in range(n):
                if lo + i == 7:
                    assert st[i] == 1 and not ct[i].any() and not ss[i].any()
                else:
                    assert st[i] == 0 and (ct[i] == ct0[lo + i]).all() and (ss[i] == ss0[lo + i]).all()
            got, st = hostapi.mlkem_decaps(768, dk[lo:lo + n], ct_bad[lo:lo + n])
            assert not st.any() and (got == ssd0[lo:lo + n]).all()
            ok = hostapi.mldsa_verify(65, pk[lo:lo + n], sig[lo:lo + n], msgs[lo:lo + n], ctxs[lo:lo + n])
            assert (ok == okv[lo:lo + n]).all()
        for i in range(t % 4, 16, 4):
            ct, ss, st = hostapi.hybrid_encaps(hostapi.X25519MLKEM768, hpk[i:i + 1], hes[i:i + 1])
            assert not st.any() and (ct[0] == hct0[i]).all() and (ss[0] == hss0[i]).all()
            ss2, st2 = hostapi.hybrid_decaps(hostapi.X25519MLKEM768, hsk[i:i + 1], ct)
            assert not st2.any() and (ss2[0] == hss0[i]).all()
    except Exception as e:
        errs.append((t, repr(e)))
th = [threading.Thread(target=body, args=(t,)) for t in range(10)]
[x.start() for x in th]; [x.join() for x in th]
assert not errs, errs[:3]
# a NULL key array is still an error, not a batch of zero keys
import ctypes as C
out = np.zeros(2000, np.uint8)
assert nat.lib().circl_hip_mlkem_encaps(768, None, out.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), None, 1, 0) == nat.EPARAM
print("ok")
"""
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])
