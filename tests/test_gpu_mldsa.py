"""GPU parity of batch ML-DSA verification against the oracle, NIST ACVP sigVer and Wycheproof,
through the C ABI (test bodies follow sign/mldsa/mldsa65/acvp_test.go:122-158 and
sign/schemes/wycheproof_test.go:116-151)."""
import numpy as np
import pytest

from conftest import hx, load_golden
from circl_amd import hostapi
from oracle import orc

pytestmark = pytest.mark.gpu
PARAMS = {"ML-DSA-44": 44, "ML-DSA-65": 65, "ML-DSA-87": 87}
Q = 8380417


def test_dilithium_ntt_against_oracle():
    # sign/internal/dilithium/ntt_test.go:53-88 (asm vs generic), here device vs oracle
    rng = np.random.default_rng(41)
    p = rng.integers(0, Q, (200, 256)).astype(np.uint32)
    got = hostapi.dilithium_ntt(p)
    inv = hostapi.dilithium_ntt(p, inverse=True)
    for i in range(200):
        assert (got[i] == orc.dilithium_normalize(orc.dilithium_ntt(p[i]))).all(), i
        assert (inv[i] == orc.dilithium_normalize(orc.dilithium_invntt(p[i]))).all(), i
    back = hostapi.dilithium_ntt(got, inverse=True)
    assert (back.astype(np.uint64) == p.astype(np.uint64) * ((1 << 32) % Q) % Q).all()


@pytest.mark.parametrize("name", list(PARAMS))
def test_acvp_sigver(name):
    p = PARAMS[name]
    pks, sigs, msgs, want = [], [], [], []
    for g in load_golden("mldsa_acvp.json.gz")[name]["sigver"]:
        for c in g["cases"]:
            pks.append(hx(g["pk"])); sigs.append(hx(c["signature"])); msgs.append(hx(c["message"])); want.append(c["passed"])
    ok = hostapi.mldsa_verify_internal(p, b"".join(pks), b"".join(sigs), msgs)
    assert ok.tolist() == [int(w) for w in want]
    assert 0 < sum(want) < len(want)


@pytest.mark.parametrize("name", list(PARAMS))
def test_wycheproof_verify(name):
    p = PARAMS[name]
    PK, SIG = hostapi.DSA_SIZES[p]
    pks, sigs, msgs, ctxs, want, ids = [], [], [], [], [], []
    for g in load_golden("mldsa_wycheproof_verify.json.gz")[name]:
        pk = hx(g["pk"])
        for t in g["tests"]:
            sig, ctx = hx(t["sig"]), hx(t["ctx"])
            if len(pk) != PK or len(sig) != SIG:
                # the Go wrapper rejects these before any arithmetic (UnmarshalBinaryPublicKey /
                # sig.Unpack length check, dilithium.go:90-93); the fixed-row C ABI never sees them
                assert t["result"] == "invalid"
                continue
            pks.append(pk); sigs.append(sig); msgs.append(hx(t["msg"])); ctxs.append(ctx)
            want.append(t["result"] == "valid"); ids.append(t["id"])
    ok = hostapi.mldsa_verify(p, b"".join(pks), b"".join(sigs), msgs, ctxs)
    bad = [(i, w) for i, o, w in zip(ids, ok.tolist(), want) if bool(o) != w]
    assert not bad, bad
    assert sum(want) >= 40 and len(want) - sum(want) >= 5
    # cross-check the whole set against the oracle as well
    assert ok.tolist() == orc.mldsa_verify(p, b"".join(pks), b"".join(sigs), msgs, ctxs).tolist()


def _signed_batch(p, n, seed):
    rng = np.random.default_rng(seed)
    pk, sk = orc.mldsa_keygen(p, rng.integers(0, 256, (n, 32), dtype=np.uint8))
    msgs = [rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8).tobytes() for _ in range(n)]
    ctxs = [rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8).tobytes() for _ in range(n)]
    sig = orc.mldsa_sign(p, sk, msgs, ctxs)
    return pk, sig, msgs, ctxs


@pytest.mark.parametrize("name", list(PARAMS))
@pytest.mark.parametrize("n", [1, 3, 130])
def test_verify_matches_oracle_with_corruptions(name, n):
    p = PARAMS[name]
    PK, SIG = hostapi.DSA_SIZES[p]
    pk, sig, msgs, ctxs = _signed_batch(p, n, seed=n * 7 + p)
    sig = sig.copy()
    rng = np.random.default_rng(5)
    ct = {44: 32, 65: 48, 87: 64}[p]
    for i in range(0, n, 4):
        kind = (i // 4) % 5
        if kind == 0:
            sig[i, int(rng.integers(0, ct))] ^= 1                      # c~
        elif kind == 1:
            sig[i, ct + int(rng.integers(0, 600))] ^= 0x40             # z
        elif kind == 2:
            sig[i, SIG - 1] = 0xFF                                     # hint switch-over point > omega
        elif kind == 3:
            msgs[i] = msgs[i] + b"x"                                   # different message
        else:
            sig[i, ct:ct + 3] = 0                                      # z coefficient out of range / garbage
    ok = hostapi.mldsa_verify(p, pk, sig, msgs, ctxs)
    want = orc.mldsa_verify(p, pk, sig, msgs, ctxs)
    assert ok.tolist() == want.tolist()
    untouched = [i for i in range(n) if i % 4]
    assert ok[untouched].all()
    assert not ok[::4].all() or n == 1


def test_verify_without_contexts_and_long_messages():
    p = 65
    rng = np.random.default_rng(9)
    n = 20
    pk, sk = orc.mldsa_keygen(p, rng.integers(0, 256, (n, 32), dtype=np.uint8))
    msgs = [rng.integers(0, 256, m, dtype=np.uint8).tobytes() for m in (0, 1, 69, 70, 71, 72, 73, 135, 205, 206, 207, 208, 341, 342, 343, 1000, 5000, 32, 32, 32)]
    sig = orc.mldsa_sign(p, sk, msgs)
    assert hostapi.mldsa_verify(p, pk, sig, msgs).all()
    assert hostapi.mldsa_verify(p, pk, sig, msgs, [b""] * n).all()
    assert not hostapi.mldsa_verify(p, pk, sig, msgs, [b"c"] * n).any()
    # ctx longer than 255 bytes -> false (mldsa65/dilithium.go:116-118)
    assert not hostapi.mldsa_verify(p, pk[:1], sig[:1], msgs[:1], [b"a" * 256]).any()


# ---- key generation ----------------------------------------------------------

@pytest.mark.parametrize("name", list(PARAMS))
def test_acvp_keygen(name):
    # sign/mldsa/mldsa65/acvp_test.go:39-79
    import hashlib
    p = PARAMS[name]
    cases = load_golden("mldsa_acvp.json.gz")[name]["keygen"]
    seeds = np.frombuffer(b"".join(hx(c["seed"]) for c in cases), np.uint8).reshape(-1, 32)
    pk, sk = hostapi.mldsa_keygen(p, seeds)
    for i, c in enumerate(cases):
        assert hashlib.sha256(pk[i].tobytes()).hexdigest() == c["pk_sha256"], i
        assert hashlib.sha256(sk[i].tobytes()).hexdigest() == c["sk_sha256"], i


@pytest.mark.parametrize("name", list(PARAMS))
@pytest.mark.parametrize("n", [1, 5, 300])
def test_keygen_matches_oracle_and_keys_work(name, n):
    p = PARAMS[name]
    rng = np.random.default_rng(n + p)
    seeds = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk, sk = hostapi.mldsa_keygen(p, seeds)
    pk0, sk0 = orc.mldsa_keygen(p, seeds)
    assert (pk == pk0).all()
    assert (sk == sk0).all()
    # sign with the oracle using the GPU-made secret keys, verify on the GPU with the GPU-made public keys
    msgs = [b"m%d" % i for i in range(n)]
    sig = orc.mldsa_sign(p, sk, msgs)
    assert hostapi.mldsa_verify(p, pk, sig, msgs).all()


# ---- signing -------------------------------------------------------------------

@pytest.mark.parametrize("name", list(PARAMS))
def test_acvp_siggen(name):
    # sign/mldsa/mldsa65/acvp_test.go:81-121: Sign_internal(sk, message, rnd), deterministic and hedged
    import hashlib
    p = PARAMS[name]
    cases = load_golden("mldsa_acvp.json.gz")[name]["siggen"]
    assert len(cases) == 20  # every case of both groups
    sk = b"".join(hx(c["sk"]) for c in cases)
    msgs = [hx(c["message"]) for c in cases]
    rnd = np.frombuffer(b"".join(hx(c["rnd"]) for c in cases), np.uint8).reshape(-1, 32)
    sig = hostapi.mldsa_sign(p, sk, msgs, rnd=rnd, internal=True)
    for i, c in enumerate(cases):
        assert hashlib.sha256(sig[i].tobytes()).hexdigest() == c["sig_sha256"], i


@pytest.mark.parametrize("name", list(PARAMS))
@pytest.mark.parametrize("n", [1, 3, 200])
def test_sign_matches_oracle_and_verifies(name, n):
    # mldsa65/internal/dilithium_test.go:94-129 sign -> verify, and bit-exact signatures vs the oracle
    p = PARAMS[name]
    rng = np.random.default_rng(n * 3 + p)
    pk, sk = orc.mldsa_keygen(p, rng.integers(0, 256, (n, 32), dtype=np.uint8))
    msgs = [rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8).tobytes() for _ in range(n)]
    ctxs = [rng.integers(0, 256, int(rng.integers(0, 60)), dtype=np.uint8).tobytes() for _ in range(n)]
    rnd = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    rnd[::2] = 0  # half deterministic, half hedged
    sig = hostapi.mldsa_sign(p, sk, msgs, ctxs, rnd=rnd)
    want = orc.mldsa_sign(p, sk, msgs, ctxs, rnd=rnd)
    assert (sig == want).all()
    assert hostapi.mldsa_verify(p, pk, sig, msgs, ctxs).all()
    det = hostapi.mldsa_sign(p, sk[:3], msgs[:3])            # no contexts, deterministic
    assert (det == orc.mldsa_sign(p, sk[:3], msgs[:3])).all()


def test_sign_rejects_long_context():
    from circl_amd import _native as nat
    pk, sk = orc.mldsa_keygen(65, np.zeros((1, 32), np.uint8))
    with pytest.raises(nat.CirclHipError):
        hostapi.mldsa_sign(65, sk, [b"m"], [b"c" * 256])


@pytest.mark.parametrize("name", list(PARAMS))
def test_wycheproof_sign_on_device(name):
    # sign/schemes/wycheproof_test.go:57-115 through the batch ABI: one batch per parameter set
    import hashlib
    from test_oracle_mldsa import _wycheproof_sign_cases
    sks, msgs, ctxs, want = [], [], [], []
    for p, SK, sk, t in _wycheproof_sign_cases(name):
        ctx = hx(t["ctx"])
        if len(sk) != SK or len(ctx) > 255:
            continue
        sks.append(sk); msgs.append(hx(t["msg"])); ctxs.append(ctx); want.append(t["sig_sha256"])
    sig = hostapi.mldsa_sign(PARAMS[name], b"".join(sks), msgs, ctxs)
    got = [hashlib.sha256(sig[i].tobytes()).hexdigest() for i in range(len(want))]
    assert got == want and len(want) >= 100


# ---- KAT transcripts and round-3 Dilithium2/3/5 (sign/dilithium/mode{2,3,5}) ----

@pytest.mark.parametrize("name,param,want", [
    # sign/dilithium/kat_test.go:25-35 replayed through the HIP path (keygen, deterministic sign, verify)
    ("Dilithium2", 2, "38ed991c5ca11e39ab23945ca37af89e059d16c5474bf8ba96b15cb4e948af2a"),
    ("Dilithium3", 3, "8196b32212753f525346201ffec1c7a0a852596fa0b57bd4e2746231dab44d55"),
    ("Dilithium5", 5, "7ded97a6e6c809b43b54c248171d7504fa6a0cab651bf288bb00034782667481"),
    ("ML-DSA-44", 44, "14f92c48abc0d63ea263cce3c83183c8360c6ede7cbd5b65bd7c6f31e38f0ea5"),
    ("ML-DSA-65", 65, "595a8eff6988159c94eb5398294458c5d27d21c994fb64cadbee339173abcf63"),
    ("ML-DSA-87", 87, "35e2ce3d88b3311517bf8d41aa2cd24aa0fbda2bb8052ca8af4ad8d7c7344074"),
])
def test_kat_transcript_hash_on_gpu(name, param, want):
    import hashlib
    from drbg import DRBG
    name_in_kat = {"ML-DSA-44": "Dilithium2", "ML-DSA-65": "Dilithium3", "ML-DSA-87": "Dilithium5"}.get(name, name)
    g = DRBG(bytes(range(48)))
    seeds, msgs, eseeds = [], [], []
    for i in range(100):
        seed = g.fill(48)
        msgs.append(g.fill(33 * (i + 1)))
        seeds.append(seed)
        eseeds.append(DRBG(seed).fill(32))
    pk, sk = hostapi.mldsa_keygen(param, np.frombuffer(b"".join(eseeds), np.uint8).reshape(100, 32))
    sig = hostapi.mldsa_sign(param, sk, msgs)
    assert hostapi.mldsa_verify(param, pk, sig, msgs).all()
    f = hashlib.sha256()
    f.update(("# %s\n\n" % name_in_kat).encode())
    for i in range(100):
        f.update(b"count = %d\n" % i)
        f.update(b"seed = %s\n" % seeds[i].hex().upper().encode())
        f.update(b"mlen = %d\n" % len(msgs[i]))
        f.update(b"msg = %s\n" % msgs[i].hex().upper().encode())
        f.update(b"pk = %s\n" % pk[i].tobytes().hex().upper().encode())
        f.update(b"sk = %s\n" % sk[i].tobytes().hex().upper().encode())
        f.update(b"smlen = %d\n" % (len(msgs[i]) + sig.shape[1]))
        f.update(b"sm = %s%s\n\n" % (sig[i].tobytes().hex().upper().encode(), msgs[i].hex().upper().encode()))
    assert f.hexdigest() == want


@pytest.mark.parametrize("param", [2, 3, 5])
def test_round3_dilithium_matches_oracle_with_corruptions(param):
    rng = np.random.default_rng(100 + param)
    n = 131
    seeds = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk, sk = hostapi.mldsa_keygen(param, seeds)
    pk_o, sk_o = orc.mldsa_keygen(param, seeds)
    assert (pk == pk_o).all() and (sk == sk_o).all()
    msgs = [rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8).tobytes() for _ in range(n)]
    sig = hostapi.mldsa_sign(param, sk, msgs)
    assert (sig == orc.mldsa_sign(param, sk, msgs)).all()
    SIG = sig.shape[1]
    bad = sig.copy()
    for i in range(0, n, 4):
        kind = (i // 4) % 4
        if kind == 0:
            bad[i, int(rng.integers(0, 32))] ^= 1          # c~ (32 bytes in round 3)
        elif kind == 1:
            bad[i, 32 + int(rng.integers(0, 600))] ^= 0x40  # z
        elif kind == 2:
            bad[i, SIG - 1] = 0xFF                          # hint switch-over point > omega
        else:
            msgs[i] = msgs[i] + b"x"
    ok = hostapi.mldsa_verify(param, pk, bad, msgs)
    assert ok.tolist() == orc.mldsa_verify(param, pk, bad, msgs).tolist()
    assert ok[[i for i in range(n) if i % 4]].all() and not ok[::4].any()


# ---- shared-key verification (one public key for the batch: the reference's parsed-key case) ----

@pytest.mark.parametrize("param", [44, 65, 87, 3])
@pytest.mark.parametrize("n", [1, 5, 333])
def test_verify_shared_key_matches_oracle(param, n):
    rng = np.random.default_rng(param * 1000 + n)
    pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (1, 32), dtype=np.uint8))
    msgs = [rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8).tobytes() for _ in range(n)]
    ctxs = None if param < 10 else [rng.integers(0, 256, int(rng.integers(0, 30)), dtype=np.uint8).tobytes() for _ in range(n)]
    sig = orc.mldsa_sign(param, np.tile(sk, (n, 1)), msgs, ctxs)
    ct = {44: 32, 65: 48, 87: 64, 3: 32}[param]
    for i in range(0, n, 3):
        if (i // 3) % 2 == 0:
            sig[i, ct + 10 + i % 200] ^= 0x10     # z
        else:
            msgs[i] = msgs[i] + b"!"
    ok = hostapi.mldsa_verify_shared(param, pk, sig, msgs, ctxs)
    want = orc.mldsa_verify(param, np.tile(pk, (n, 1)), sig, msgs, ctxs)
    assert ok.tolist() == want.tolist()
    assert ok[[i for i in range(n) if i % 3]].all() and not ok[::3].any()


@pytest.mark.parametrize("param", [44, 65, 87, 3])
@pytest.mark.parametrize("n", [1, 5, 40, 700])
def test_sign_shared_key_matches_oracle(param, n):
    # n < 16 takes the persistent kernel, larger batches the phase-split path with its speculative tail
    rng = np.random.default_rng(param * 77 + n)
    pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (1, 32), dtype=np.uint8))
    msgs = [rng.integers(0, 256, int(rng.integers(0, 120)), dtype=np.uint8).tobytes() for _ in range(n)]
    ctxs = None if param < 10 else [rng.integers(0, 256, int(rng.integers(0, 20)), dtype=np.uint8).tobytes() for _ in range(n)]
    rnd = None if param < 10 else rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sig = hostapi.mldsa_sign_shared(param, sk, msgs, ctxs, rnd)
    want = orc.mldsa_sign(param, np.tile(sk, (n, 1)), msgs, ctxs, rnd)
    assert (sig == want).all()
    assert hostapi.mldsa_verify_shared(param, pk, sig, msgs, ctxs).all()


@pytest.mark.gpu
def test_concurrent_device_sign_calls_do_not_share_state():
    # The C ABI is re-entrant (SURVEY 8b "Threading"): three host threads sign different batches at the same time on one
    # device through the device-pointer entry point, each on its own stream and workspace.  The per-round count read-backs
    # of the batched signer must not be shared between the calls.
    import ctypes as C
    import threading
    import torch
    from circl_amd import _native as nat
    L = nat.lib()
    param = 65
    PK, SK, SIG = orc.DSA_SIZES[param]
    rng = np.random.default_rng(77)
    jobs = []
    for t, n in enumerate((700, 450, 900)):
        pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (n, 32), dtype=np.uint8))
        msgs = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n)]
        jobs.append({"n": n, "sk": sk, "msgs": msgs, "want": orc.mldsa_sign(param, sk, msgs)})
    for j in jobs:
        n = j["n"]
        j["d_sk"] = torch.from_numpy(j["sk"]).cuda()
        j["d_msg"] = torch.from_numpy(np.frombuffer(b"".join(j["msgs"]) + bytes(16), dtype=np.uint8).copy()).cuda()
        j["d_off"] = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64).cuda()
        j["d_rnd"] = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
        j["sig"] = torch.empty((n, SIG), dtype=torch.uint8, device="cuda")
        wsb = L.circl_hip_mldsa_sign_workspace_size(param, n)
        j["ws"], j["wsb"] = torch.empty(wsb, dtype=torch.uint8, device="cuda"), wsb
        j["stream"] = torch.cuda.Stream()
    torch.cuda.synchronize()

    def run(j):
        torch.cuda.set_device(0)
        for _ in range(3):
            j["rc"] = L.circl_hip_mldsa_sign_dev(param, j["d_sk"].data_ptr(), j["d_msg"].data_ptr(), j["d_off"].data_ptr(), None, None,
                                                 j["d_rnd"].data_ptr(), 0, j["sig"].data_ptr(), j["n"], j["ws"].data_ptr(), j["wsb"],
                                                 C.c_void_p(j["stream"].cuda_stream))
            if j["rc"] != 0:
                return

    threads = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    for j in jobs:
        assert j["rc"] == 0
        assert (j["sig"].cpu().numpy() == j["want"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("param,shared", [(65, False), (44, False), (87, True)])
def test_sign_speculative_rounds_match_oracle(param, shared):
    # 3000 items: above the hand-over to the persistent tail (2 items per CU) and below the speculation target, so the
    # batched rounds run with k = 1, 2, 3, ... attempts per item and the commit picks the lowest successful one --
    # the signature must still be the one of the reference's sequential rejection loop.
    rng = np.random.default_rng(1234 + param)
    n = 3000
    nk = 1 if shared else n
    pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (nk, 32), dtype=np.uint8))
    msgs = [rng.integers(0, 256, 1 + (i % 90), dtype=np.uint8).tobytes() for i in range(n)]
    if shared:
        want = orc.mldsa_sign(param, np.repeat(sk, n, axis=0), msgs)
        got = hostapi.mldsa_sign_shared(param, sk[0], msgs)
    else:
        want = orc.mldsa_sign(param, sk, msgs)
        got = hostapi.mldsa_sign(param, sk, msgs)
    assert (got == want).all()
