"""Pins the CPU oracle (oracle/kyber.c, oracle/keccak.c) against every vector the reference's own
tests hold for the ML-KEM path.  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import hx, load_golden
from drbg import DRBG
from oracle import orc

PARAMS = {"ML-KEM-512": 512, "ML-KEM-768": 768, "ML-KEM-1024": 1024}
Q = 3329


def test_keccak_permutation_of_zero():
    # simd/keccakf1600/f1600x_test.go:9-19
    want = np.array(load_golden("fixed_vectors.json.gz")["keccak_f1600_of_zero"], dtype=np.uint64)
    assert (orc.keccak_f1600(np.zeros(25, np.uint64)) == want).all()


def test_keccak_turbo_12_rounds_is_last_12():
    # keccakf.go:20-24: turbo starts at round 12.  24 rounds == 12 "front" rounds then 12 turbo rounds
    # is not expressible without the front half, so check the weaker property that the variants differ
    # and that turbo is deterministic.
    a = np.arange(25, dtype=np.uint64)
    assert (orc.keccak_f1600(a, 12) == orc.keccak_f1600(a, 12)).all()
    assert (orc.keccak_f1600(a, 12) != orc.keccak_f1600(a, 24)).any()


@pytest.mark.parametrize("alg,rate,ds", [("SHA3-256", 136, 6), ("SHA3-512", 72, 6), ("SHAKE128", 168, 0x1F), ("SHAKE256", 136, 0x1F)])
def test_sha3_short_msg_kats(alg, rate, ds):
    # internal/sha3/testdata/keccakKats.json.deflate (internal/sha3/sha3_test.go:55-90)
    kats = load_golden("sha3_kats.json.gz")[alg]
    assert len(kats) >= 16
    for k in kats:
        d = hx(k["digest"])
        assert orc.sponge(hx(k["msg"]), len(d), rate, ds) == d


def test_sponge_matches_hashlib_on_block_edges():
    rng = np.random.default_rng(1)
    for n in (0, 1, 71, 72, 73, 135, 136, 137, 167, 168, 169, 1184, 1120):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert orc.sha3_256(d) == hashlib.sha3_256(d).digest()
        assert orc.sha3_512(d) == hashlib.sha3_512(d).digest()
        assert orc.shake128(d, 505) == hashlib.shake_128(d).digest(505)
        assert orc.shake256(d, 273) == hashlib.shake_256(d).digest(273)


def test_fixed_sampler_vectors():
    # pke/kyber/internal/common/sample_test.go:23-138, seed[i] = i
    fx = load_golden("fixed_vectors.json.gz")
    seed = bytes(range(32))
    assert orc.kyber_noise(seed, 37, 3).tolist() == fx["kyber_noise3_seed_i_nonce37"]
    assert orc.kyber_noise(seed, 37, 2).tolist() == fx["kyber_noise2_seed_i_nonce37"]
    assert orc.kyber_uniform(seed, 1, 0).tolist() == fx["kyber_uniform_seed_i_x1_y0"]


def test_zetas_table():
    # ntt.go:16-28: Zetas[i] = 17^brv7(i) * 2^16 mod q
    z = orc.kyber_zetas()
    assert z[0] == 2285 and z[1] == 2571 and z[127] == 1628 and z[64] == 2226


@pytest.mark.parametrize("name", list(PARAMS))
def test_acvp_keygen(name):
    # kem/mlkem/acvp_test.go:35-82
    p = PARAMS[name]
    cases = load_golden("mlkem_acvp.json.gz")[name]["keygen"]
    assert len(cases) == 25
    seeds = np.frombuffer(b"".join(hx(c["d"]) + hx(c["z"]) for c in cases), np.uint8).reshape(-1, 64)
    ek, dk = orc.mlkem_keygen(p, seeds)
    for i, c in enumerate(cases):
        assert hashlib.sha256(ek[i].tobytes()).hexdigest() == c["ek_sha256"]
        assert hashlib.sha256(dk[i].tobytes()).hexdigest() == c["dk_sha256"]


@pytest.mark.parametrize("name", list(PARAMS))
def test_acvp_encap(name):
    # kem/mlkem/acvp_test.go:83-125
    p = PARAMS[name]
    cases = load_golden("mlkem_acvp.json.gz")[name]["encap"]
    assert len(cases) == 25
    ek = np.frombuffer(b"".join(hx(c["ek"]) for c in cases), np.uint8)
    m = np.frombuffer(b"".join(hx(c["m"]) for c in cases), np.uint8)
    ct, ss, st = orc.mlkem_encaps(p, ek, m)
    assert (st == 0).all()
    for i, c in enumerate(cases):
        assert ct[i].tobytes() == hx(c["c"])
        assert ss[i].tobytes() == hx(c["k"])


@pytest.mark.parametrize("name", list(PARAMS))
def test_acvp_decap(name):
    # kem/mlkem/acvp_test.go:126-160 (includes implicit-rejection ciphertexts)
    p = PARAMS[name]
    groups = load_golden("mlkem_acvp.json.gz")[name]["decap"]
    total = 0
    for g in groups:
        dk = hx(g["dk"])
        n = len(g["cases"])
        total += n
        ss, st = orc.mlkem_decaps(p, dk * n, b"".join(hx(c["c"]) for c in g["cases"]))
        assert (st == 0).all()
        for i, c in enumerate(g["cases"]):
            assert ss[i].tobytes() == hx(c["k"])
    assert total == 10


@pytest.mark.parametrize("name,want", [
    # kem/kyber/kat_test.go:31-33
    ("ML-KEM-512", "a30184edee53b3b009356e1e31d7f9e93ce82550e3c622d7192e387b0cc84f2e"),
    ("ML-KEM-768", "729367b590637f4a93c68d5e4a4d2e2b4454842a52c9eec503e3a0d24cb66471"),
    ("ML-KEM-1024", "3fba7327d0320cb6134badf2a1bcb963a5b3c0026c7dece8f00d6a6155e47b33"),
])
def test_kat_transcript_hash(name, want):
    # kem/kyber/kat_test.go:42-94: 100 keygen + encaps + decaps through the NIST DRBG
    p = PARAMS[name]
    g = DRBG(bytes(range(48)))
    f = hashlib.sha256()
    f.update(("# %s\n\n" % name.replace("ML-KEM-", "Kyber")).encode())
    for i in range(100):
        seed = g.fill(48)
        f.update(b"count = %d\n" % i)
        f.update(b"seed = %s\n" % seed.hex().upper().encode())
        g2 = DRBG(seed)
        kseed = g2.fill(64)
        eseed = g2.fill(32)
        ek, dk = orc.mlkem_keygen(p, kseed, threads=1)
        ct, ss, st = orc.mlkem_encaps(p, ek, eseed, threads=1)
        ss2, st2 = orc.mlkem_decaps(p, dk, ct, threads=1)
        assert st[0] == 0 and st2[0] == 0 and (ss == ss2).all()
        f.update(b"pk = %s\n" % ek.tobytes().hex().upper().encode())
        f.update(b"sk = %s\n" % dk.tobytes().hex().upper().encode())
        f.update(b"ct = %s\n" % ct.tobytes().hex().upper().encode())
        f.update(b"ss = %s\n\n" % ss.tobytes().hex().upper().encode())
    assert f.hexdigest() == want


@pytest.mark.parametrize("name,want", [
    # kem/kyber/kat_test.go:25-27 -- round-3 Kyber
    ("Kyber1024", "89248f2f33f7f4f7051729111f3049c409a933ec904aedadf035f30fa5646cd5"),
    ("Kyber768", "a1e122cad3c24bc51622e4c242d8b8acbcd3f618fee4220400605ca8f9ea02c2"),
    ("Kyber512", "e9c2bd37133fcb40772f81559f14b1f58dccd1c816701be9ba6214d43baf4547"),
])
def test_kat_transcript_hash_round3_kyber(name, want):
    # kem/kyber/kat_test.go:42-94; the reference implementation calls randombytes twice for the key seed
    p = int(name[len("Kyber"):])
    g = DRBG(bytes(range(48)))
    f = hashlib.sha256()
    f.update(("# %s\n\n" % name).encode())
    for i in range(100):
        seed = g.fill(48)
        f.update(b"count = %d\n" % i)
        f.update(b"seed = %s\n" % seed.hex().upper().encode())
        g2 = DRBG(seed)
        kseed = g2.fill(32) + g2.fill(32)
        eseed = g2.fill(32)
        ek, dk = orc.kyber_r3_keygen(p, kseed, threads=1)
        ct, ss = orc.kyber_r3_encaps(p, ek, eseed, threads=1)
        ss2 = orc.kyber_r3_decaps(p, dk, ct, threads=1)
        assert (ss == ss2).all()
        f.update(b"pk = %s\n" % ek.tobytes().hex().upper().encode())
        f.update(b"sk = %s\n" % dk.tobytes().hex().upper().encode())
        f.update(b"ct = %s\n" % ct.tobytes().hex().upper().encode())
        f.update(b"ss = %s\n\n" % ss.tobytes().hex().upper().encode())
    assert f.hexdigest() == want


def test_round3_kyber_implicit_rejection_and_lenient_keys():
    # kem/kyber/kyber768/kyber.go:184-196: a modified ciphertext yields KDF(z || H(c)), no error;
    # :248-262: coefficients in {q..4095} of a public key are accepted and reduced
    rng = np.random.default_rng(11)
    ek, dk = orc.kyber_r3_keygen(768, rng.integers(0, 256, (4, 64), dtype=np.uint8))
    ct, ss = orc.kyber_r3_encaps(768, ek, rng.integers(0, 256, (4, 32), dtype=np.uint8))
    bad = ct.copy()
    bad[:, 5] ^= 1
    ssb = orc.kyber_r3_decaps(768, dk, bad)
    for i in range(4):
        z = dk[i, 2400 - 32:].tobytes()
        want = orc.sponge(z + orc.sha3_256(bad[i].tobytes()), 32, 136, 0x1f)
        assert ssb[i].tobytes() == want and ssb[i].tobytes() != ss[i].tobytes()
    # first coefficient of the key +q (still < 4096 when the coefficient is < 767): different bytes, same ring element
    ek2 = ek.copy()
    for i in range(4):
        c0 = int(ek2[i, 0]) | ((int(ek2[i, 1]) & 15) << 8)
        if c0 + 3329 < 4096:
            c1 = c0 + 3329
            ek2[i, 0] = c1 & 255
            ek2[i, 1] = (ek2[i, 1] & 0xf0) | (c1 >> 8)
    seeds = rng.integers(0, 256, (4, 32), dtype=np.uint8)
    ct2, ss2 = orc.kyber_r3_encaps(768, ek2, seeds)   # must not fail
    assert ct2.shape == (4, 1088)


# ---- algebraic properties the reference re-checks (ntt_test.go, poly_test.go) ----

def _rand_abs_le_q(rng):
    return rng.integers(-Q + 1, Q, 256).astype(np.int16)


def test_ntt_roundtrip_is_times_2_16():
    # ntt_test.go:83-109 TestNTT
    rng = np.random.default_rng(7)
    for _ in range(100):
        p = _rand_abs_le_q(rng)
        q = orc.kyber_normalize(p)
        t = orc.kyber_ntt(p)
        assert (np.abs(t.astype(np.int32)) <= 7 * Q).all()
        t = orc.kyber_normalize(orc.kyber_invntt(orc.kyber_normalize(t)))
        assert (t.astype(np.int64) == (q.astype(np.int64) << 16) % Q).all()


def test_mulhat_is_negacyclic_product():
    # poly_test.go:92-133 TestMulHat: InvNTT(MulHat(NTT a, NTT b)) == schoolbook a*b mod (x^256+1)
    rng = np.random.default_rng(8)
    for _ in range(20):
        a = rng.integers(0, Q, 256).astype(np.int16)
        b = rng.integers(0, Q, 256).astype(np.int16)
        full = np.convolve(a.astype(np.int64), b.astype(np.int64))
        want = full[:256].copy()
        want[:255] -= full[256:]
        want %= Q
        ah, bh = orc.kyber_normalize(orc.kyber_ntt(a)), orc.kyber_normalize(orc.kyber_ntt(b))
        ph = orc.kyber_mulhat(ah, bh)           # a*b*R^-1 in the NTT domain
        got = orc.kyber_normalize(orc.kyber_invntt(orc.kyber_normalize(ph)))  # * R  -> a*b
        assert (got.astype(np.int64) == want).all()


@pytest.mark.parametrize("d", [4, 5, 10, 11])
def test_compress_is_exact_rounding(d):
    # poly_test.go:351-378 TestCompressFullInputFirstCoeff: multiply-shift == round(x*2^d/q) mod 2^d
    for base in range(0, Q, 256):
        xs = np.arange(base, min(base + 256, Q))
        p = np.zeros(256, np.int16)
        p[: len(xs)] = xs
        m = orc.kyber_compress(p, d)
        bits = np.unpackbits(m, bitorder="little")[: 256 * d].reshape(256, d)
        got = (bits.astype(np.int64) << np.arange(d)).sum(axis=1)[: len(xs)]
        want = ((xs.astype(np.int64) << d) + Q // 2) // Q % (1 << d)
        assert (got == want).all()
        back = orc.kyber_decompress(m, d)[: len(xs)].astype(np.int64)
        assert (back == ((want * Q + (1 << (d - 1))) >> d)).all()


def test_encaps_rejects_non_canonical_ek():
    # cpapke.go:45-55 UnpackMLKEM -> kem.ErrPubKey
    ek, _ = orc.mlkem_keygen(768, np.arange(64, dtype=np.uint8))
    bad = ek.copy()
    bad[0, 0] = 0xFF
    bad[0, 1] |= 0x0F  # first coefficient = 0xfff >= q
    ct, ss, st = orc.mlkem_encaps(768, np.concatenate([ek, bad]), np.zeros((2, 32), np.uint8))
    assert st.tolist() == [0, 1]
    assert not ct[1].any() and not ss[1].any()


def test_decaps_rejects_bad_hash_and_implicit_rejection():
    # kyber.go:219-228 ErrPrivKey ; kyber.go:171-181 implicit rejection = SHAKE256(z || ct)[:32]
    ek, dk = orc.mlkem_keygen(768, np.arange(64, dtype=np.uint8))
    ct, ss, _ = orc.mlkem_encaps(768, ek, np.full((1, 32), 7, np.uint8))
    bad_dk = dk.copy()
    bad_dk[0, 2400 - 64] ^= 1
    _, st = orc.mlkem_decaps(768, bad_dk, ct)
    assert st[0] == 2
    bad_ct = ct.copy()
    bad_ct[0, 5] ^= 0x10
    ss2, st = orc.mlkem_decaps(768, dk, bad_ct)
    assert st[0] == 0 and (ss2 != ss).any()
    assert ss2[0].tobytes() == hashlib.shake_256(dk[0, -32:].tobytes() + bad_ct[0].tobytes()).digest(32)


def test_turboshake128_reference_vectors():
    # internal/sha3/sha3_test.go:264-284 TestTurboShake128
    assert orc.sponge_rounds(b"", 64, 168, 0x07, 12).hex() == (
        "5a223ad30b3b8c66a243048cfced430f54e7529287d15150b973133adfac6a2ffe2708e73061e09a4000168ba9c8ca1813198f7bbed4984b4185f2c2580ee623")
    assert orc.sponge_rounds(b"", 10032, 168, 0x07, 12)[-32:].hex() == "7593a28020a3c4ae0d605fd61f5eb56eccd27cc3d12ff09f78369772a460c55d"
    assert orc.sponge_rounds(b"\xff", 32, 168, 0x06, 12).hex() == "8ec9c66465ed0d4a6c35d13506718d687a25cb05c74cca1e42501abd83874a67"
    assert orc.sponge_rounds(b"abc", 77, 136, 0x1F, 24) == hashlib.shake_256(b"abc").digest(77)
