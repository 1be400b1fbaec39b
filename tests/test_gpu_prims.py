"""GPU parity of the building blocks against the oracle, mirroring the reference's unit tests
(TestNTTAgainstGeneric, TestInvNTTAgainstGeneric, TestMulHat, f1600x_test.go, sha3_test.go).
All calls go through the C ABI."""
import hashlib

import numpy as np
import pytest

from conftest import hx, load_golden
from circl_amd import hostapi
from oracle import orc

pytestmark = pytest.mark.gpu
Q = 3329


def test_keccak_f1600_zero_and_random():
    want0 = np.array(load_golden("fixed_vectors.json.gz")["keccak_f1600_of_zero"], dtype=np.uint64)
    rng = np.random.default_rng(11)
    st = rng.integers(0, 1 << 63, (1000, 25), dtype=np.uint64) * 2 + rng.integers(0, 2, (1000, 25), dtype=np.uint64)
    st[0] = 0
    got = hostapi.keccak_f1600(st)
    assert (got[0] == want0).all()
    for i in range(0, 1000, 37):
        assert (got[i] == orc.keccak_f1600(st[i])).all()
    got12 = hostapi.keccak_f1600(st[:64], rounds=12)
    for i in range(0, 64, 9):
        assert (got12[i] == orc.keccak_f1600(st[i], 12)).all()


@pytest.mark.parametrize("rate,ds,name", [(168, 0x1F, "shake_128"), (136, 0x1F, "shake_256"), (136, 6, "sha3_256"), (72, 6, "sha3_512")])
def test_sponge_vs_hashlib(rate, ds, name):
    rng = np.random.default_rng(rate + ds)
    for inlen in (0, 1, 33, 34, 71, 72, 73, 135, 136, 137, 167, 168, 169, 500):
        msgs = rng.integers(0, 256, (70, max(inlen, 1)), dtype=np.uint8)[:, :inlen]
        outlen = {"sha3_256": 32, "sha3_512": 64}.get(name, 400)
        got = hostapi.shake(rate, ds, np.ascontiguousarray(msgs).reshape(70, inlen), outlen)
        for i in (0, 1, 63, 64, 69):
            h = getattr(hashlib, name)(msgs[i].tobytes())
            want = h.digest(outlen) if name.startswith("shake") else h.digest()
            assert got[i].tobytes() == want, (name, inlen, i)


def test_sha3_kats_on_device():
    kats = load_golden("sha3_kats.json.gz")
    for alg, rate, ds in (("SHA3-256", 136, 6), ("SHAKE128", 168, 0x1F)):
        for k in kats[alg][:12]:
            msg = np.frombuffer(hx(k["msg"]), np.uint8).reshape(1, -1)
            d = hx(k["digest"])
            assert hostapi.shake(rate, ds, msg, len(d))[0].tobytes() == d


def test_ntt_against_oracle():
    rng = np.random.default_rng(5)
    p = rng.integers(-Q + 1, Q, (300, 256)).astype(np.int16)
    got = hostapi.kyber_ntt(p)
    for i in range(300):
        assert (got[i] == orc.kyber_normalize(orc.kyber_ntt(p[i]))).all(), i


def test_invntt_against_oracle():
    rng = np.random.default_rng(6)
    p = rng.integers(-Q + 1, Q, (300, 256)).astype(np.int16)
    got = hostapi.kyber_ntt(p, inverse=True)
    for i in range(300):
        assert (got[i] == orc.kyber_normalize(orc.kyber_invntt(orc.kyber_normalize(p[i])))).all(), i


def test_ntt_roundtrip_times_r():
    rng = np.random.default_rng(7)
    p = rng.integers(0, Q, (64, 256)).astype(np.int16)
    back = hostapi.kyber_ntt(hostapi.kyber_ntt(p), inverse=True)
    assert (back.astype(np.int64) == (p.astype(np.int64) << 16) % Q).all()


def test_mulhat_against_oracle():
    rng = np.random.default_rng(8)
    a = rng.integers(0, Q, (200, 256)).astype(np.int16)
    b = rng.integers(0, Q, (200, 256)).astype(np.int16)
    got = hostapi.kyber_mulhat(a, b)
    for i in range(200):
        assert (got[i] == orc.kyber_normalize(orc.kyber_mulhat(a[i], b[i]))).all(), i
