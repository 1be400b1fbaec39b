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
