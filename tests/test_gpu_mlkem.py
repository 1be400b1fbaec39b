"""GPU parity of batch ML-KEM against the oracle and the NIST ACVP vectors, through the C ABI
(the test bodies follow kem/mlkem/acvp_test.go and kem/schemes/schemes_test.go)."""
import numpy as np
import pytest

from conftest import hx, load_golden
from circl_amd import hostapi
from oracle import orc

pytestmark = pytest.mark.gpu
PARAMS = {"ML-KEM-512": 512, "ML-KEM-768": 768, "ML-KEM-1024": 1024}


def _keys(p, n, seed=0):
    rng = np.random.default_rng(seed + p)
    ek, dk = orc.mlkem_keygen(p, rng.integers(0, 256, (n, 64), dtype=np.uint8))
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    return ek, dk, m


@pytest.mark.parametrize("name", list(PARAMS))
def test_acvp_encap(name):
    p = PARAMS[name]
    cases = load_golden("mlkem_acvp.json.gz")[name]["encap"]
    ek = np.frombuffer(b"".join(hx(c["ek"]) for c in cases), np.uint8)
    m = np.frombuffer(b"".join(hx(c["m"]) for c in cases), np.uint8)
    ct, ss, st = hostapi.mlkem_encaps(p, ek, m)
    assert (st == 0).all()
    for i, c in enumerate(cases):
        assert ct[i].tobytes() == hx(c["c"]), i
        assert ss[i].tobytes() == hx(c["k"]), i


@pytest.mark.parametrize("name", list(PARAMS))
@pytest.mark.parametrize("n", [1, 7, 64, 1000])
def test_encaps_matches_oracle(name, n):
    # ragged sizes: not a multiple of the items-per-workgroup (16 / 7 / 4) nor of 64
    p = PARAMS[name]
    ek, _, m = _keys(p, n, seed=n)
    ct, ss, st = hostapi.mlkem_encaps(p, ek, m)
    ct0, ss0, st0 = orc.mlkem_encaps(p, ek, m)
    assert (st == st0).all() and (st == 0).all()
    assert (ss == ss0).all()
    assert (ct == ct0).all()


def test_encaps_empty_batch():
    ct, ss, st = hostapi.mlkem_encaps(768, np.zeros((0, 1184), np.uint8), np.zeros((0, 32), np.uint8))
    assert ct.shape == (0, 1088) and ss.shape == (0, 32) and st.shape == (0,)


@pytest.mark.parametrize("name", list(PARAMS))
def test_encaps_rejects_non_canonical_ek(name):
    # cpapke.go:45-55 UnpackMLKEM -> kem.ErrPubKey ; neighbours in the same workgroup are unaffected
    p = PARAMS[name]
    ek, _, m = _keys(p, 40, seed=3)
    bad = ek.copy()
    for i in (0, 5, 17, 39):
        bad[i, 3 * (i % 100)] = 0xFF
        bad[i, 3 * (i % 100) + 1] |= 0x0F
    bad[9, 382] |= 0xF0
    bad[9, 383] = 0xFF  # last coefficient of the first polynomial
    ct, ss, st = hostapi.mlkem_encaps(p, bad, m)
    ct0, ss0, st0 = orc.mlkem_encaps(p, bad, m)
    assert st.tolist() == st0.tolist() and st0.sum() == 5
    assert (ct == ct0).all() and (ss == ss0).all()


def test_shared_key_batch_matches_oracle():
    # the reference benchmark's shape (kem/schemes/schemes_test.go:28-38): one ek, many m
    ek, _, _ = _keys(768, 1, seed=77)
    m = np.random.default_rng(78).integers(0, 256, (2048, 32), dtype=np.uint8)
    eks = np.repeat(ek, 2048, axis=0)
    ct, ss, st = hostapi.mlkem_encaps(768, eks, m)
    ct0, ss0, _ = orc.mlkem_encaps(768, eks, m)
    assert (ct == ct0).all() and (ss == ss0).all() and (st == 0).all()


def test_large_batch_sampled_parity_and_fourth_block_streams():
    # 2^16 distinct keys: ~0.83 % of the 9 * 2^16 matrix streams need a 4th SHAKE128 block
    n = 1 << 16
    ek, _, m = _keys(768, n, seed=99)
    ct, ss, st = hostapi.mlkem_encaps(768, ek, m)
    ct0, ss0, _ = orc.mlkem_encaps(768, ek, m)
    assert (st == 0).all()
    assert (ct == ct0).all() and (ss == ss0).all()


# ---- key generation and decapsulation --------------------------------------------------------

@pytest.mark.parametrize("name", list(PARAMS))
def test_acvp_keygen(name):
    # kem/mlkem/acvp_test.go:35-82
    import hashlib
    p = PARAMS[name]
    cases = load_golden("mlkem_acvp.json.gz")[name]["keygen"]
    seeds = np.frombuffer(b"".join(hx(c["d"]) + hx(c["z"]) for c in cases), np.uint8).reshape(-1, 64)
    ek, dk = hostapi.mlkem_keygen(p, seeds)
    for i, c in enumerate(cases):
        assert hashlib.sha256(ek[i].tobytes()).hexdigest() == c["ek_sha256"], i
        assert hashlib.sha256(dk[i].tobytes()).hexdigest() == c["dk_sha256"], i


@pytest.mark.parametrize("name", list(PARAMS))
def test_acvp_decap(name):
    # kem/mlkem/acvp_test.go:126-160 (includes implicit-rejection ciphertexts)
    p = PARAMS[name]
    for g in load_golden("mlkem_acvp.json.gz")[name]["decap"]:
        n = len(g["cases"])
        ss, st = hostapi.mlkem_decaps(p, hx(g["dk"]) * n, b"".join(hx(c["c"]) for c in g["cases"]))
        assert (st == 0).all()
        for i, c in enumerate(g["cases"]):
            assert ss[i].tobytes() == hx(c["k"]), i


@pytest.mark.parametrize("name", list(PARAMS))
@pytest.mark.parametrize("n", [1, 5, 129, 1000])
def test_keygen_matches_oracle(name, n):
    p = PARAMS[name]
    seeds = np.random.default_rng(n + p).integers(0, 256, (n, 64), dtype=np.uint8)
    ek, dk = hostapi.mlkem_keygen(p, seeds)
    ek0, dk0 = orc.mlkem_keygen(p, seeds)
    assert (ek == ek0).all()
    assert (dk == dk0).all()


@pytest.mark.parametrize("name", list(PARAMS))
@pytest.mark.parametrize("n", [1, 9, 300])
def test_keygen_encaps_decaps_roundtrip_on_device(name, n):
    # kem/schemes/schemes_test.go:53-167 round trip, every step on the GPU, checked against the oracle
    p = PARAMS[name]
    rng = np.random.default_rng(1000 + n)
    ek, dk = hostapi.mlkem_keygen(p, rng.integers(0, 256, (n, 64), dtype=np.uint8))
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ct, ss, st = hostapi.mlkem_encaps(p, ek, m)
    ss2, st2 = hostapi.mlkem_decaps(p, dk, ct)
    assert (st == 0).all() and (st2 == 0).all()
    assert (ss == ss2).all()
    ss3, _ = orc.mlkem_decaps(p, dk, ct)
    assert (ss2 == ss3).all()


@pytest.mark.parametrize("name", list(PARAMS))
def test_decaps_implicit_rejection_and_bad_keys(name):
    p = PARAMS[name]
    n = 64
    ek, dk, m = _keys(p, n, seed=5)
    ct, ss, _ = orc.mlkem_encaps(p, ek, m)
    ct = ct.copy()
    dk = dk.copy()
    EK, DK, CT = orc.KEM_SIZES[p]
    for i in range(0, n, 3):            # corrupted ciphertexts: not an error, pseudo-random key
        ct[i, (7 * i) % CT] ^= 1 << (i % 8)
    for i in (1, 10, 40):               # dk whose stored H(ek) does not match -> kem.ErrPrivKey
        dk[i, DK - 64 + (i % 32)] ^= 0x80
    for i in (2, 11):                   # non-canonical coefficient inside the embedded ek AND a fixed-up
        off = DK - 64 - EK              # hash: the reference reduces it mod q and carries on
        dk[i, off] = 0xFF
        dk[i, off + 1] |= 0x0F
        import hashlib
        dk[i, DK - 64:DK - 32] = np.frombuffer(hashlib.sha3_256(dk[i, off:off + EK].tobytes()).digest(), np.uint8)
    got, st = hostapi.mlkem_decaps(p, dk, ct)
    want, st0 = orc.mlkem_decaps(p, dk, ct)
    assert st.tolist() == st0.tolist() and sorted(set(st0.tolist())) == [0, 2]
    assert (got == want).all()
    good = [i for i in range(n) if i % 3 and i not in (1, 10, 40, 2, 11)]
    assert (got[good] == ss[good]).all()


def test_decaps_large_batch_matches_encaps():
    n = 1 << 15
    rng = np.random.default_rng(31)
    ek, dk = hostapi.mlkem_keygen(768, rng.integers(0, 256, (n, 64), dtype=np.uint8))
    ct, ss, _ = hostapi.mlkem_encaps(768, ek, rng.integers(0, 256, (n, 32), dtype=np.uint8))
    ss2, st = hostapi.mlkem_decaps(768, dk, ct)
    assert (st == 0).all() and (ss == ss2).all()
    idx = rng.choice(n, 2048, replace=False)
    want, _ = orc.mlkem_decaps(768, dk[idx], ct[idx])
    assert (ss2[idx] == want).all()
