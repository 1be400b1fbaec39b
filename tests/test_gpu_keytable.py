"""Key tables that live across calls (circl_hip_*_keytable_new, include/circl_hip.h): the reference's parsed key objects.
kem.Scheme.UnmarshalBinaryPublicKey / UnmarshalBinaryPrivateKey keep A^T and H(ek) in the object (kem/mlkem/mlkem768/kyber.go:39-43,
:219-228, :247-263), sign.Scheme.UnmarshalBinaryPublicKey keeps A and tr (sign/mldsa/mldsa65/internal/dilithium.go:114-126); every
later Encapsulate / Decapsulate / Verify on the object reuses them.  Here: a table built once, used by many calls, against the oracle."""
import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("param", [512, 768, 1024])
def test_mlkem_tables_match_the_oracle_call_after_call(param):
    from circl_amd import hostapi
    rng = np.random.default_rng(param)
    nkeys = 23
    ek, dk = orc.mlkem_keygen(param, rng.integers(0, 256, (nkeys, 64), dtype=np.uint8))
    dk_bad = dk.copy()
    dk_bad[5, -40] ^= 1                                  # entry 5: stored H(ek) no longer matches (kem.ErrPrivKey)
    pub = hostapi.KeyTable("mlkem-public", param, ek)
    prv = hostapi.KeyTable("mlkem-private", param, dk_bad)
    assert prv.key_status.tolist() == [2 if i == 5 else 0 for i in range(nkeys)]
    for n in (1, 7, 300, 2500, 40000):                   # small routes, big routes, and a host call of more than one chunk
        m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        idx = rng.integers(0, nkeys, n).astype(np.uint32)
        idx[:2] = [nkeys - 1, 0][:min(n, 2)]
        ct, ss, st = pub.encaps(m, idx)
        ct0, ss0, _ = orc.mlkem_encaps(param, ek[idx], m)
        assert (st == 0).all() and (ct == ct0).all() and (ss == ss0).all(), n
        ct[::3, 11] ^= 8                                  # implicit rejection for every third item
        got, st = prv.decaps(ct, idx)
        want, _ = orc.mlkem_decaps(param, dk[idx], ct)
        bad = idx == 5
        assert (st[bad] == 2).all() and not got[bad].any() and (st[~bad] == 0).all() and (got[~bad] == want[~bad]).all(), n
        # no index vector: every item uses entry 0 -- a table of one key object
        ct1, ss1, st1 = pub.encaps(m)
        ct10, ss10, _ = orc.mlkem_encaps(param, np.tile(ek[:1], (n, 1)), m)
        assert (st1 == 0).all() and (ct1 == ct10).all() and (ss1 == ss10).all(), n
        ct1[::3, 11] ^= 8
        got, st = prv.decaps(ct1)
        want, _ = orc.mlkem_decaps(param, np.tile(dk[:1], (n, 1)), ct1)
        assert (st == 0).all() and (got == want).all(), n
    # an index beyond the table, a table of the wrong kind
    with pytest.raises(Exception):
        pub.encaps(rng.integers(0, 256, (3, 32), dtype=np.uint8), np.array([0, nkeys, 1], np.uint32))
    with pytest.raises(Exception):
        prv.encaps(rng.integers(0, 256, (3, 32), dtype=np.uint8))
    pub.close()
    prv.close()
    with pytest.raises(Exception):                       # a freed table is refused, not dereferenced
        pub.encaps(rng.integers(0, 256, (3, 32), dtype=np.uint8))


def test_mlkem_table_reports_a_non_canonical_public_key_per_item():
    from circl_amd import hostapi
    rng = np.random.default_rng(9)
    ek, _ = orc.mlkem_keygen(768, rng.integers(0, 256, (3, 64), dtype=np.uint8))
    ek[1, 0] = 0xff
    ek[1, 1] |= 0x0f                                      # first coefficient of entry 1 = 0xfff >= q (cpapke.go:45-55)
    t = hostapi.KeyTable("mlkem-public", 768, ek)
    idx = np.array([0, 1, 2, 1, 0], np.uint32)
    ct, ss, st = t.encaps(rng.integers(0, 256, (5, 32), dtype=np.uint8), idx)
    assert st.tolist() == [0, 1, 0, 1, 0] and not ct[st == 1].any() and not ss[st == 1].any()


@pytest.mark.parametrize("param", [44, 65, 87, 3])
def test_mldsa_table_matches_the_oracle_call_after_call(param):
    from circl_amd import hostapi
    rng = np.random.default_rng(70 + param)
    nkeys = 9
    r3 = param in (2, 3, 5)
    pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (nkeys, 32), dtype=np.uint8))
    t = hostapi.KeyTable("mldsa-public", param, pk)
    for n in (1, 40, 1500):                               # cooperative pre-pass, lane pairs
        idx = rng.integers(0, nkeys, n).astype(np.uint32)
        msgs = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 300, n)]
        if n > 20:
            msgs[3] = bytes(rng.integers(0, 256, 5000, dtype=np.uint8))  # a long message goes through the pre-pass with the table's tr
        ctxs = None if r3 else [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 30, n)]
        sig = hostapi.mldsa_sign(param, sk[idx], msgs, ctxs=ctxs)
        bad = sig.copy()
        bad[1::5, 7] ^= 2
        ok = t.verify(bad, msgs, ctxs=ctxs, key_idx=idx).astype(bool)
        want = orc.mldsa_verify(param, pk[idx], bad, msgs, ctxs=ctxs).astype(bool)
        assert (ok == want).all() and ok[0::5].all() and not ok[1::5].any(), n
        # entry 0 for every item
        sig0 = hostapi.mldsa_sign(param, np.tile(sk[:1], (n, 1)), msgs, ctxs=ctxs)
        sig0[2::7, 40] ^= 1
        ok = t.verify(sig0, msgs, ctxs=ctxs).astype(bool)
        want = np.ones(n, bool)
        want[2::7] = False
        assert (ok == want).all(), n
    t.close()


@pytest.mark.parametrize("param", [44, 65, 87, 3])
def test_mldsa_prepared_private_key_signs_like_the_oracle(param):
    # the parsed PrivateKey of the reference keeps A and the NTT-domain s1, s2, t0 (internal/dilithium.go:149-179): prepared once
    # here, then every call signs with it -- deterministic and hedged, contexts, short and long messages, small and larger batches
    from circl_amd import hostapi
    rng = np.random.default_rng(900 + param)
    r3 = param in (2, 3, 5)
    pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (1, 32), dtype=np.uint8))
    t = hostapi.KeyTable("mldsa-private", param, sk)
    for n in (1, 33, 700):
        msgs = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 200, n)]
        if n > 20:
            msgs[5] = bytes(rng.integers(0, 256, 4000, dtype=np.uint8))
        ctxs = None if r3 else [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 40, n)]
        rnd = None if r3 else rng.integers(0, 256, (n, 32), dtype=np.uint8)
        if rnd is not None:
            rnd[::2] = 0
        got = t.sign(msgs, ctxs=ctxs, rnd=rnd)
        want = orc.mldsa_sign(param, np.tile(sk, (n, 1)), msgs, ctxs=ctxs, rnd=rnd)
        assert (got == want).all(), n
        assert hostapi.mldsa_verify_shared(param, pk, got, msgs, ctxs=ctxs).all()
    t.close()


# ---- round 4: several prepared private keys, hybrid key objects, replicated tables, batch Public() ---------------------------------
@pytest.mark.parametrize("param", [44, 65, 87, 3])
def test_mldsa_table_of_prepared_private_keys_signs_like_the_oracle(param):
    # a signer with several identities: every key parsed once (internal/dilithium.go:149-179), message i signed with entry key_idx[i]:
    # the same bytes as scheme.Sign with the gathered keys; small batches (speculating rounds, persistent tail), a batch above the
    # lane-pair thresholds, deterministic and hedged, contexts, one long message
    from circl_amd import hostapi
    rng = np.random.default_rng(1300 + param)
    r3 = param in (2, 3, 5)
    nkeys = 11
    pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (nkeys, 32), dtype=np.uint8))
    t = hostapi.KeyTable("mldsa-private", param, sk)
    assert t.nkeys == nkeys
    for n in (1, 5, 90, 1100):
        idx = rng.integers(0, nkeys, n).astype(np.uint32)
        idx[0] = nkeys - 1
        msgs = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 200, n)]
        if n > 20:
            msgs[5] = bytes(rng.integers(0, 256, 4000, dtype=np.uint8))   # the long-message pre-pass takes tr from entry key_idx[5]
        ctxs = None if r3 else [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 40, n)]
        rnd = None if r3 else rng.integers(0, 256, (n, 32), dtype=np.uint8)
        if rnd is not None:
            rnd[::2] = 0
        got = t.sign(msgs, ctxs=ctxs, rnd=rnd, key_idx=idx)
        k = min(n, 160)                                                      # (the oracle signs ~1 ms per signature)
        want = orc.mldsa_sign(param, sk[idx[:k]], msgs[:k], ctxs=None if ctxs is None else ctxs[:k], rnd=None if rnd is None else rnd[:k])
        assert (got[:k] == want).all(), n
        assert hostapi.mldsa_verify(param, pk[idx], got, msgs, ctxs=ctxs).all(), n
        # no index vector: entry 0
        got0 = t.sign(msgs[:k], ctxs=None if ctxs is None else ctxs[:k])
        assert (got0 == orc.mldsa_sign(param, np.tile(sk[:1], (k, 1)), msgs[:k], ctxs=None if ctxs is None else ctxs[:k])).all(), n
    with pytest.raises(Exception):
        t.sign([b"x", b"y"], key_idx=np.array([0, nkeys], np.uint32))
    t.close()


@pytest.mark.parametrize("scheme_name", ["xwing", "x25519mlkem768"])
def test_hybrid_key_tables_match_the_oracle_call_after_call(scheme_name):
    # kem/xwing/xwing.go:20-31 (PrivateKey caches the expanded ML-KEM key, the X25519 scalar and its public point; PublicKey the
    # parsed ML-KEM key), kem/hybrid/hybrid.go:101-114: keys parsed once, then encapsulations / decapsulations call after call
    from circl_amd import hostapi
    from oracle import hybrid as ohyb
    scheme = hostapi.XWING if scheme_name == "xwing" else hostapi.X25519MLKEM768
    S = hostapi.HYBRID_SIZES[scheme]
    enc0 = ohyb.xwing_encaps if scheme == hostapi.XWING else ohyb.hybrid_encaps
    dec0 = ohyb.xwing_decaps if scheme == hostapi.XWING else ohyb.hybrid_decaps
    rng = np.random.default_rng(4100 + scheme)
    nkeys = 6
    pk, sk = hostapi.hybrid_keygen(scheme, rng.integers(0, 256, (nkeys, S["seed"]), dtype=np.uint8))
    pub = hostapi.KeyTable("hybrid-public", scheme, pk)
    prv = hostapi.KeyTable("hybrid-private", scheme, sk)
    assert not prv.key_status.any()
    for n in (1, 9, 300, 2200):
        es = rng.integers(0, 256, (n, S["eseed"]), dtype=np.uint8)
        idx = rng.integers(0, nkeys, n).astype(np.uint32)
        idx[0] = nkeys - 1
        ct, ss, st = pub.hybrid_encaps(es, idx)
        k = min(n, 200)
        ct0, ss0, st0 = enc0(pk[idx[:k]], es[:k])
        assert not st.any() and (ct[:k] == ct0).all() and (ss[:k] == ss0).all(), n
        cta, ssa, sta = hostapi.hybrid_encaps(scheme, pk[idx], es)          # and the per-item entry point on the gathered keys, all items
        assert (ct == cta).all() and (ss == ssa).all() and (st == sta).all(), n
        ct2 = ct.copy()
        ct2[::3, 17] ^= 4                                                    # ML-KEM half: implicit rejection
        got, st = prv.hybrid_decaps(ct2, idx)
        want, stw = hostapi.hybrid_decaps(scheme, sk[idx], ct2)
        assert (st == stw).all() and (got == want).all(), n
        good, st = prv.hybrid_decaps(ct, idx)
        assert not st.any() and (good == ss).all(), n
        r0 = dec0(sk[idx[:k]], ct2[:k])                                      # (X-Wing has no failing decapsulation: the oracle returns ss alone)
        want0, st0 = r0 if isinstance(r0, tuple) else (r0, np.zeros(k, np.uint8))
        assert (got[:k] == want0).all() and (stw[:k] == st0).all(), n
        # entry 0 for every item
        ct1, ss1, st1 = pub.hybrid_encaps(es)
        ctb, ssb, _ = hostapi.hybrid_encaps(scheme, np.tile(pk[:1], (n, 1)), es)
        assert (ct1 == ctb).all() and (ss1 == ssb).all() and not st1.any(), n
        got1, st1 = prv.hybrid_decaps(ct1)
        assert (got1 == ss1).all() and not st1.any(), n
    if scheme == hostapi.X25519MLKEM768:
        # a private key whose ML-KEM half fails its hash check (kem.ErrPrivKey), a low-order X25519 share (kem.ErrPubKey)
        sk_bad = sk.copy()
        sk_bad[2, 2400 - 40] ^= 1
        tb = hostapi.KeyTable("hybrid-private", scheme, sk_bad)
        assert tb.key_status.tolist() == [0, 0, 2, 0, 0, 0]
        es = rng.integers(0, 256, (12, S["eseed"]), dtype=np.uint8)
        idx = (np.arange(12) % nkeys).astype(np.uint32)
        ct, ss, _ = pub.hybrid_encaps(es, idx)
        ct[4, 1088:] = 0                                                     # X25519 share = the point of order 4's u-coordinate 0
        got, st = tb.hybrid_decaps(ct, idx)
        want, stw = hostapi.hybrid_decaps(scheme, sk_bad[idx], ct)
        assert (st == stw).all() and (got == want).all() and st[2] == 2 and st[8] == 2 and st[4] == 1
        tb.close()
    with pytest.raises(Exception):
        pub.hybrid_decaps(np.zeros((2, S["ct"]), np.uint8))
    pub.close()
    prv.close()


def test_batch_public_from_private_keys():
    # kem/mlkem/mlkem768/kyber.go:323-328 and sign/mldsa/mldsa65/internal/dilithium.go:473-484: PrivateKey.Public() over a batch
    from circl_amd import hostapi
    rng = np.random.default_rng(77)
    for param in (512, 768, 1024):
        ek, dk = orc.mlkem_keygen(param, rng.integers(0, 256, (50, 64), dtype=np.uint8))
        assert (hostapi.mlkem_public_from_private(param, dk) == ek).all()
    for param in (44, 65, 87, 2, 3, 5):
        for n in (1, 37, 1300):
            pk, sk = hostapi.mldsa_keygen(param, rng.integers(0, 256, (n, 32), dtype=np.uint8))
            assert (hostapi.mldsa_public_from_private(param, sk) == pk).all(), (param, n)
    pk0, sk0 = orc.mldsa_keygen(65, rng.integers(0, 256, (20, 32), dtype=np.uint8))   # oracle-made keys
    assert (hostapi.mldsa_public_from_private(65, sk0) == pk0).all()
