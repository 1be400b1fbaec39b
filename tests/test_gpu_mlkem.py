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


# ---- round-3 Kyber (kem/kyber/kyber{512,768,1024}), SURVEY 8f row f3 ----

@pytest.mark.gpu
@pytest.mark.parametrize("name,want", [
    # kem/kyber/kat_test.go:25-27, replayed through the HIP path (the 100 DRBG-derived seeds form one batch)
    ("Kyber1024", "89248f2f33f7f4f7051729111f3049c409a933ec904aedadf035f30fa5646cd5"),
    ("Kyber768", "a1e122cad3c24bc51622e4c242d8b8acbcd3f618fee4220400605ca8f9ea02c2"),
    ("Kyber512", "e9c2bd37133fcb40772f81559f14b1f58dccd1c816701be9ba6214d43baf4547"),
])
def test_round3_kyber_kat_transcript_hash_on_gpu(name, want):
    import hashlib
    from drbg import DRBG
    p = int(name[len("Kyber"):])
    g = DRBG(bytes(range(48)))
    seeds, kseeds, eseeds = [], [], []
    for i in range(100):
        seed = g.fill(48)
        g2 = DRBG(seed)
        seeds.append(seed)
        kseeds.append(g2.fill(32) + g2.fill(32))
        eseeds.append(g2.fill(32))
    ks = np.frombuffer(b"".join(kseeds), dtype=np.uint8).reshape(100, 64)
    es = np.frombuffer(b"".join(eseeds), dtype=np.uint8).reshape(100, 32)
    ek, dk = hostapi.kyber_keygen(p, ks)
    ct, ss = hostapi.kyber_encaps(p, ek, es)
    ss2 = hostapi.kyber_decaps(p, dk, ct)
    assert (ss == ss2).all()
    f = hashlib.sha256()
    f.update(("# %s\n\n" % name).encode())
    for i in range(100):
        f.update(b"count = %d\n" % i)
        f.update(b"seed = %s\n" % seeds[i].hex().upper().encode())
        f.update(b"pk = %s\n" % ek[i].tobytes().hex().upper().encode())
        f.update(b"sk = %s\n" % dk[i].tobytes().hex().upper().encode())
        f.update(b"ct = %s\n" % ct[i].tobytes().hex().upper().encode())
        f.update(b"ss = %s\n\n" % ss[i].tobytes().hex().upper().encode())
    assert f.hexdigest() == want


@pytest.mark.gpu
@pytest.mark.parametrize("param", [512, 768, 1024])
def test_round3_kyber_matches_oracle_incl_rejection_and_lenient_keys(param):
    rng = np.random.default_rng(param + 3)
    n = 777   # ragged: not a multiple of any group size
    seeds = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    ek, dk = hostapi.kyber_keygen(param, seeds)
    ek_o, dk_o = orc.kyber_r3_keygen(param, seeds)
    assert (ek == ek_o).all() and (dk == dk_o).all()
    # non-canonical keys: add q to the first coefficient where it still fits 12 bits (kyber.go:248-262: accepted)
    ek2 = ek.copy()
    for i in range(0, n, 3):
        c0 = int(ek2[i, 0]) | ((int(ek2[i, 1]) & 15) << 8)
        if c0 + 3329 < 4096:
            c1 = c0 + 3329
            ek2[i, 0] = c1 & 255
            ek2[i, 1] = (ek2[i, 1] & 0xf0) | (c1 >> 8)
    es = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ct, ss = hostapi.kyber_encaps(param, ek2, es)
    ct_o, ss_o = orc.kyber_r3_encaps(param, ek2, es)
    assert (ct == ct_o).all() and (ss == ss_o).all()
    # decapsulation of honest and of tampered ciphertexts (implicit rejection, kyber.go:184-196)
    ct_h, ss_h = hostapi.kyber_encaps(param, ek, es)
    bad = ct_h.copy()
    bad[::2, 7] ^= 0x40
    got = hostapi.kyber_decaps(param, dk, bad)
    want = orc.kyber_r3_decaps(param, dk, bad)
    assert (got == want).all()
    assert (got[1::2] == ss_h[1::2]).all() and not (got[::2] == ss_h[::2]).all(axis=1).any()


# ---- shared-key encapsulation (one ek for the batch: kem/schemes/schemes_test.go:28-38 BenchmarkEncapsulate's shape) ----

@pytest.mark.gpu
@pytest.mark.parametrize("name", list(PARAMS))
@pytest.mark.parametrize("n", [1, 7, 9, 1000, 4099])
def test_encaps_shared_key_matches_oracle(name, n):
    p = PARAMS[name]
    rng = np.random.default_rng(n + p)
    ek, dk = orc.mlkem_keygen(p, rng.integers(0, 256, (1, 64), dtype=np.uint8))
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ct, ss, st = hostapi.mlkem_encaps_shared(p, ek, m)
    ct0, ss0, st0 = orc.mlkem_encaps(p, np.tile(ek, (n, 1)), m)
    assert (st == 0).all() and (ct == ct0).all() and (ss == ss0).all()
    ss2, st2 = hostapi.mlkem_decaps(p, np.tile(dk, (n, 1)), ct)
    assert (st2 == 0).all() and (ss2 == ss).all()


@pytest.mark.gpu
def test_encaps_shared_key_rejects_non_canonical_key():
    rng = np.random.default_rng(3)
    ek, _ = orc.mlkem_keygen(768, rng.integers(0, 256, (1, 64), dtype=np.uint8))
    bad = ek.copy()
    bad[0, 0] = 0xff
    bad[0, 1] |= 0x0f          # first coefficient = 0xfff >= q  (cpapke.go:45-55)
    ct, ss, st = hostapi.mlkem_encaps_shared(768, bad, rng.integers(0, 256, (50, 32), dtype=np.uint8))
    assert (st == 1).all() and not ct.any() and not ss.any()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(PARAMS))
@pytest.mark.parametrize("n", [1, 10, 2051])
def test_decaps_shared_key_matches_oracle(name, n):
    p = PARAMS[name]
    rng = np.random.default_rng(n * 3 + p)
    ek, dk = orc.mlkem_keygen(p, rng.integers(0, 256, (1, 64), dtype=np.uint8))
    ct, ss, _ = orc.mlkem_encaps(p, np.tile(ek, (n, 1)), rng.integers(0, 256, (n, 32), dtype=np.uint8))
    ct[::3, 9] ^= 2                                  # implicit rejection for every third item
    got, st = hostapi.mlkem_decaps_shared(p, dk, ct)
    want, st0 = orc.mlkem_decaps(p, np.tile(dk, (n, 1)), ct)
    assert (st == 0).all() and (got == want).all()
    assert (got[1::3] == ss[1::3]).all()
    bad = dk.copy()
    bad[0, -40] ^= 1                                 # stored H(ek) no longer matches: kem.ErrPrivKey for the whole batch
    got, st = hostapi.mlkem_decaps_shared(p, bad, ct)
    assert (st == 2).all() and not got.any()


@pytest.mark.gpu
def test_all_devices_sharding_gives_the_same_results():
    # device = CIRCL_HIP_ALL_DEVICES (-1): contiguous split over every visible GPU, one host thread each, no collective
    # (SURVEY 8e).  With one GPU it is a single shard; with more the outputs must still land in the caller's order.
    from circl_amd import _native as nat
    rng = np.random.default_rng(44)
    n = 3001
    seeds = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    ek, dk = hostapi.mlkem_keygen(768, seeds, device=nat.ALL_DEVICES)
    ek0, dk0 = orc.mlkem_keygen(768, seeds)
    assert (ek == ek0).all() and (dk == dk0).all()
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ct, ss, st = hostapi.mlkem_encaps(768, ek, m, device=nat.ALL_DEVICES)
    ct0, ss0, _ = orc.mlkem_encaps(768, ek, m)
    assert (st == 0).all() and (ct == ct0).all() and (ss == ss0).all()
    ss2, st2 = hostapi.mlkem_decaps(768, dk, ct, device=nat.ALL_DEVICES)
    assert (st2 == 0).all() and (ss2 == ss).all()
    pk, sk = orc.mldsa_keygen(65, seeds[:200, :32])
    msgs = [bytes([i % 256]) * (i % 50) for i in range(200)]
    sig = hostapi.mldsa_sign(65, sk, msgs, device=nat.ALL_DEVICES)
    assert (sig == orc.mldsa_sign(65, sk, msgs)).all()
    assert hostapi.mldsa_verify(65, pk, sig, msgs, device=nat.ALL_DEVICES).all()
