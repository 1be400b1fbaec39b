"""The hybrid KEMs of SURVEY.md 8(f) row f2 through the C ABI (circl_hip_hybrid_*), batch level, against the oracle's
restatement (oracle/hybrid.py).  The X-Wing draft's test-vector transcript (kem/xwing/xwing_test.go:38-85) and the
reference's xkem / schemes tests run through the C++ mirrors in tests/test_gpu_host_mirror.py, which sit on the same ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOW_ORDER = bytes.fromhex("e0eb7a7c3b41b8ae1656e3faf19fc46ada098deb9c32b1fd866205165f49b800")


@pytest.fixture(scope="module")
def api():
    from circl_amd import hostapi
    return hostapi


@pytest.fixture(scope="module")
def ho():
    from oracle import hybrid
    return hybrid


@pytest.mark.parametrize("n", [1, 65, 1500])
def test_xwing_against_oracle(api, ho, n):
    rng = np.random.default_rng(n)
    seeds = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    es = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    pk, sk = api.hybrid_keygen(api.XWING, seeds)
    pk0, sk0, _, _ = ho.xwing_keygen(seeds)
    assert (pk == pk0).all() and (sk == sk0).all()
    if n > 4:
        pk[1, 0:2] = 0xff            # coefficient 0xfff >= q: ML-KEM encapsulation-key check fails
        pk[2, 1184:] = np.frombuffer(LOW_ORDER, np.uint8)  # low-order X25519 point: not an error in X-Wing
    ct, ss, st = api.hybrid_encaps(api.XWING, pk, es)
    ct0, ss0, st0 = ho.xwing_encaps(pk, es)
    assert (st == st0).all() and (ct == ct0).all() and (ss == ss0).all()
    if n > 4:
        assert st[1] == 1 and st[2] == 0 and st.sum() == 1
    ct[:: 3, 5] ^= 1                 # implicit rejection in the ML-KEM half
    ss2, st2 = api.hybrid_decaps(api.XWING, sk, ct)
    assert (ss2 == ho.xwing_decaps(sk, ct)).all() and not st2.any()
    good = np.ones(n, bool)
    good[:: 3] = False
    if n > 4:
        good[1] = good[2] = False    # encapsulated to a modified public key
    assert (ss2[good] == ss[good]).all() and (n < 3 or (ss2[~good] != ss[~good]).any())


@pytest.mark.parametrize("n", [1, 65, 1500])
def test_x25519mlkem768_against_oracle(api, ho, n):
    S = api.X25519MLKEM768
    rng = np.random.default_rng(100 + n)
    seeds = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    es = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk, sk = api.hybrid_keygen(S, seeds)
    pk0, sk0 = ho.hybrid_keygen(seeds)
    assert (pk == pk0).all() and (sk == sk0).all()
    if n > 4:
        pk[1, 0:2] = 0xff
        pk[2, 1184:] = np.frombuffer(LOW_ORDER, np.uint8)  # xkem.go:144-146 -> kem.ErrPubKey
        pk[3, 1184:] = 0
    ct, ss, st = api.hybrid_encaps(S, pk, es)
    ct0, ss0, st0 = ho.hybrid_encaps(pk, es)
    assert (st == st0).all() and (ct == ct0).all() and (ss == ss0).all()
    if n > 4:
        assert list(st[:5]) == [0, 1, 1, 1, 0] and not ct[1:4].any() and not ss[1:4].any()
    if n > 8:
        sk[5, 2400 - 40] ^= 1        # H(ek) stored in dk no longer matches -> kem.ErrPrivKey
        ct[6, 1088:] = np.frombuffer(LOW_ORDER, np.uint8)  # low-order ciphertext point -> kem.ErrPubKey
        ct[7, 9] ^= 4                # implicit rejection: not an error
    ss2, st2 = api.hybrid_decaps(S, sk, ct)
    ss20, st20 = ho.hybrid_decaps(sk, ct)
    assert (st2 == st20).all() and (ss2 == ss20).all()
    if n > 8:
        assert st2[5] == 2 and st2[6] == 1 and st2[7] == 0 and not ss2[5].any() and not ss2[6].any()
        assert (ss2[8:] == ss[8:]).all() and (ss2[7] != ss[7]).any()


@pytest.mark.parametrize("scheme", [1, 2, 3, 4])
def test_round_trip_full_chip_batch(api, scheme):
    n = 1 << 16
    S = api.HYBRID_SIZES[scheme]
    rng = np.random.default_rng(scheme)
    pk, sk = api.hybrid_keygen(scheme, rng.integers(0, 256, (n, S["seed"]), dtype=np.uint8))
    ct, ss, st = api.hybrid_encaps(scheme, pk, rng.integers(0, 256, (n, S["eseed"]), dtype=np.uint8))
    ss2, st2 = api.hybrid_decaps(scheme, sk, ct)
    assert not st.any() and not st2.any() and (ss == ss2).all() and ss.any(axis=1).all()
    assert len(np.unique(ss[:, :8].copy().view(np.uint64))) == n


@pytest.mark.parametrize("scheme", [3, 4])
@pytest.mark.parametrize("n", [1, 300])
def test_kyber_x25519_hybrids_against_oracle(api, ho, scheme, n):
    # hybrid.Kyber768X25519() / Kyber512X25519() (hybrid.go:71-81): X25519 is the first component, round-3 Kyber the second
    S = api.HYBRID_SIZES[scheme]
    rng = np.random.default_rng(1000 * scheme + n)
    seeds = rng.integers(0, 256, (n, S["seed"]), dtype=np.uint8)
    es = rng.integers(0, 256, (n, S["eseed"]), dtype=np.uint8)
    pk, sk = api.hybrid_keygen(scheme, seeds)
    pk0, sk0 = ho.hybrid_keygen(seeds, scheme)
    assert pk.shape[1] == S["pk"] and (pk == pk0).all() and (sk == sk0).all()
    if n > 8:
        pk[2, :32] = np.frombuffer(LOW_ORDER, np.uint8)   # low-order X25519 public key -> kem.ErrPubKey
        pk[3, 32:34] = 0xff                               # round-3 Kyber reduces a coefficient >= q: not an error (kyber.go:248-262)
    ct, ss, st = api.hybrid_encaps(scheme, pk, es)
    ct0, ss0, st0 = ho.hybrid_encaps(pk, es, scheme)
    assert (st == st0).all() and (ct == ct0).all() and (ss == ss0).all()
    if n > 8:
        assert st[2] == 1 and st.sum() == 1 and not ct[2].any()
        ct[5, :32] = np.frombuffer(LOW_ORDER, np.uint8)   # low-order ciphertext point
        ct[6, 40] ^= 1                                    # implicit rejection in the Kyber half
    ss2, st2 = api.hybrid_decaps(scheme, sk, ct)
    ss20, st20 = ho.hybrid_decaps(sk, ct, scheme)
    assert (st2 == st20).all() and (ss2 == ss20).all()
    if n > 8:
        assert st2[5] == 1 and st2[2] == 1 and st2.sum() == 2   # item 2's ciphertext is all zero: its X25519 point is the low-order 0
        assert (ss2[7:] == ss[7:]).all() and (ss2[6] != ss[6]).any()
    else:
        assert (ss2 == ss).all()
