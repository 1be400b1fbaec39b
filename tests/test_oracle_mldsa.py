"""Pins the CPU oracle (oracle/dilithium.c) against the reference's ML-DSA vectors
(SURVEY.md section 8c): NIST ACVP keyGen / sigGen / sigVer and Wycheproof verify.  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import hx, load_golden
from oracle import orc

PARAMS = {"ML-DSA-44": 44, "ML-DSA-65": 65, "ML-DSA-87": 87}
Q = 8380417


def test_fixed_uniform_vector():
    # sign/mldsa/mldsa65/internal/sample_test.go:12-63
    want = load_golden("fixed_vectors.json.gz")["dilithium_uniform_seed_i_nonce30000"]
    assert orc.dilithium_uniform(bytes(range(32)), 30000).tolist() == want


def test_zetas_table():
    # sign/internal/dilithium/ntt.go:19-57 (first / last entries)
    z = orc.dilithium_zetas()
    assert z[0] == 4193792 and z[1] == 25847 and z[2] == 5771523 and z[255] == 1976782


def test_ntt_roundtrip():
    # sign/internal/dilithium/ntt_test.go:25-51: InvNTT(NTT(p)) == p * R mod q
    rng = np.random.default_rng(3)
    R = (1 << 32) % Q
    for _ in range(50):
        p = rng.integers(0, Q, 256).astype(np.uint32)
        t = orc.dilithium_ntt(p)
        assert (t < 18 * Q).all()
        t = orc.dilithium_normalize(orc.dilithium_invntt(orc.dilithium_normalize(t)))
        assert (t.astype(np.uint64) == p.astype(np.uint64) * R % Q).all()


@pytest.mark.parametrize("name", list(PARAMS))
def test_acvp_keygen(name):
    p = PARAMS[name]
    cases = load_golden("mldsa_acvp.json.gz")[name]["keygen"]
    assert len(cases) == 25
    seeds = np.frombuffer(b"".join(hx(c["seed"]) for c in cases), np.uint8)
    pk, sk = orc.mldsa_keygen(p, seeds)
    for i, c in enumerate(cases):
        assert hashlib.sha256(pk[i].tobytes()).hexdigest() == c["pk_sha256"]
        assert hashlib.sha256(sk[i].tobytes()).hexdigest() == c["sk_sha256"]


@pytest.mark.parametrize("name", list(PARAMS))
def test_acvp_siggen(name):
    # acvp_test.go:81-121: Sign_internal(sk, message, rnd)
    p = PARAMS[name]
    cases = load_golden("mldsa_acvp.json.gz")[name]["siggen"]
    assert len(cases) == 20  # all of them, deterministic and hedged
    for c in cases:
        sig = orc.mldsa_sign_one(p, hx(c["sk"]), hx(c["message"]), rnd=hx(c["rnd"]), internal=True)
        assert hashlib.sha256(sig).hexdigest() == c["sig_sha256"]


@pytest.mark.parametrize("name", list(PARAMS))
def test_acvp_sigver(name):
    # acvp_test.go:122-158: Verify_internal; only some cases are positive
    p = PARAMS[name]
    groups = load_golden("mldsa_acvp.json.gz")[name]["sigver"]
    n = pos = 0
    for g in groups:
        for c in g["cases"]:
            got = orc.mldsa_verify_one(p, hx(g["pk"]), hx(c["message"]), hx(c["signature"]), internal=True)
            assert got == c["passed"]
            n += 1
            pos += got
    assert n == 15 and 0 < pos < 15


@pytest.mark.parametrize("name", list(PARAMS))
def test_wycheproof_verify(name):
    # sign/schemes/wycheproof_test.go:116-151 (public Verify with contexts)
    p = PARAMS[name]
    PK = orc.DSA_SIZES[p][0]
    groups = load_golden("mldsa_wycheproof_verify.json.gz")[name]
    n = valid = 0
    for g in groups:
        pk = hx(g["pk"])
        for t in g["tests"]:
            n += 1
            if len(pk) != PK:  # UnmarshalBinaryPublicKey fails -> only invalid cases allowed
                assert t["result"] == "invalid"
                continue
            ok = orc.mldsa_verify_one(p, pk, hx(t["msg"]), hx(t["sig"]), ctx=hx(t["ctx"]))
            assert ok == (t["result"] == "valid"), (t["id"], t["comment"])
            valid += ok
    assert n >= 60 and valid >= 40


@pytest.mark.parametrize("name", list(PARAMS))
def test_sign_verify_roundtrip_batch(name):
    # mldsa65/internal/dilithium_test.go:94-129
    p = PARAMS[name]
    rng = np.random.default_rng(p)
    n = 16
    pk, sk = orc.mldsa_keygen(p, rng.integers(0, 256, (n, 32), dtype=np.uint8))
    msgs = [rng.integers(0, 256, int(rng.integers(0, 100)), dtype=np.uint8).tobytes() for _ in range(n)]
    ctxs = [b"ctx%d" % i for i in range(n)]
    sig = orc.mldsa_sign(p, sk, msgs, ctxs)
    assert orc.mldsa_verify(p, pk, sig, msgs, ctxs).all()
    bad = sig.copy()
    bad[:, 40] ^= 1
    assert not orc.mldsa_verify(p, pk, bad, msgs, ctxs).any()
    assert not orc.mldsa_verify(p, pk, sig, msgs, [b"other"] * n).any()


def _wycheproof_sign_cases(name):
    p = PARAMS[name]
    SK = orc.DSA_SIZES[p][1]
    for g in load_golden("mldsa_wycheproof_sign.json.gz")[name]:
        if g["seed"] is not None:
            _, sk = orc.mldsa_keygen(p, np.frombuffer(hx(g["seed"]), np.uint8))
            sk = sk[0].tobytes()
        else:
            sk = hx(g["sk"])
        for t in g["tests"]:
            # sign/schemes/wycheproof_test.go:79-84 skips these two (keys the reference does not reject)
            if t["comment"] in ("private key with s1 vector out of range", "private key with s2 vector out of range"):
                continue
            yield p, SK, sk, t


@pytest.mark.parametrize("name", list(PARAMS))
def test_wycheproof_sign(name):
    # sign/schemes/wycheproof_test.go:57-115: deterministic Sign with contexts, from seeds and from packed keys
    n = 0
    for p, SK, sk, t in _wycheproof_sign_cases(name):
        ctx = hx(t["ctx"])
        if len(sk) != SK or len(ctx) > 255:
            assert t["result"] == "invalid"   # UnmarshalBinaryPrivateKey / ErrContextTooLong
            continue
        assert t["result"] == "valid", t["comment"]
        sig = orc.mldsa_sign_one(p, sk, hx(t["msg"]), ctx=ctx)
        assert hashlib.sha256(sig).hexdigest() == t["sig_sha256"], (t["id"], t["comment"])
        n += 1
    assert n >= 100


@pytest.mark.parametrize("name,param,want", [
    # sign/dilithium/kat_test.go:25-35: SHA-256 over 100 keygen + deterministic-sign transcripts
    ("Dilithium2", 2, "38ed991c5ca11e39ab23945ca37af89e059d16c5474bf8ba96b15cb4e948af2a"),
    ("Dilithium3", 3, "8196b32212753f525346201ffec1c7a0a852596fa0b57bd4e2746231dab44d55"),
    ("Dilithium5", 5, "7ded97a6e6c809b43b54c248171d7504fa6a0cab651bf288bb00034782667481"),
    ("ML-DSA-44", 44, "14f92c48abc0d63ea263cce3c83183c8360c6ede7cbd5b65bd7c6f31e38f0ea5"),
    ("ML-DSA-65", 65, "595a8eff6988159c94eb5398294458c5d27d21c994fb64cadbee339173abcf63"),
    ("ML-DSA-87", 87, "35e2ce3d88b3311517bf8d41aa2cd24aa0fbda2bb8052ca8af4ad8d7c7344074"),
])
def test_kat_transcript_hash(name, param, want):
    import hashlib
    from drbg import DRBG
    name_in_kat = {"ML-DSA-44": "Dilithium2", "ML-DSA-65": "Dilithium3", "ML-DSA-87": "Dilithium5"}.get(name, name)
    g = DRBG(bytes(range(48)))
    f = hashlib.sha256()
    f.update(("# %s\n\n" % name_in_kat).encode())
    seeds, msgs, eseeds = [], [], []
    for i in range(100):
        seed = g.fill(48)
        msgs.append(g.fill(33 * (i + 1)))
        seeds.append(seed)
        eseeds.append(DRBG(seed).fill(32))
    pk, sk = orc.mldsa_keygen(param, np.frombuffer(b"".join(eseeds), np.uint8).reshape(100, 32))
    sig = orc.mldsa_sign(param, sk, msgs)   # empty context, rnd = 0: deterministic (mldsa65/dilithium.go:56-99 with nil opts)
    ok = orc.mldsa_verify(param, pk, sig, msgs)
    assert ok.all()
    sig_size = orc.DSA_SIZES[param][2]
    for i in range(100):
        f.update(b"count = %d\n" % i)
        f.update(b"seed = %s\n" % seeds[i].hex().upper().encode())
        f.update(b"mlen = %d\n" % len(msgs[i]))
        f.update(b"msg = %s\n" % msgs[i].hex().upper().encode())
        f.update(b"pk = %s\n" % pk[i].tobytes().hex().upper().encode())
        f.update(b"sk = %s\n" % sk[i].tobytes().hex().upper().encode())
        f.update(b"smlen = %d\n" % (len(msgs[i]) + sig_size))
        f.update(b"sm = %s%s\n\n" % (sig[i].tobytes().hex().upper().encode(), msgs[i].hex().upper().encode()))
    assert f.hexdigest() == want
