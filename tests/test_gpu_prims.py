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


def test_xof_service_variable_length_and_turboshake():
    # batched XOF over ragged messages; TurboSHAKE128 vectors of internal/sha3/sha3_test.go:264-284
    rng = np.random.default_rng(77)
    msgs = [rng.integers(0, 256, int(k), dtype=np.uint8).tobytes() for k in [0, 1, 135, 136, 137, 167, 168, 169, 1000] + list(rng.integers(0, 600, 300))]
    got = hostapi.xof(136, 0x1F, msgs, 200)
    for i, m in enumerate(msgs):
        assert got[i].tobytes() == hashlib.shake_256(m).digest(200), i
    got = hostapi.xof(168, 0x1F, msgs, 33)
    for i in (0, 5, 8, 100):
        assert got[i].tobytes() == hashlib.shake_128(msgs[i]).digest(33)
    t = hostapi.xof(168, 0x07, [b""], 10032, rounds=12)[0].tobytes()
    assert t[:64].hex() == "5a223ad30b3b8c66a243048cfced430f54e7529287d15150b973133adfac6a2ffe2708e73061e09a4000168ba9c8ca1813198f7bbed4984b4185f2c2580ee623"
    assert t[-32:].hex() == "7593a28020a3c4ae0d605fd61f5eb56eccd27cc3d12ff09f78369772a460c55d"
    assert hostapi.xof(168, 0x06, [b"\xff"], 32, rounds=12)[0].tobytes().hex() == "8ec9c66465ed0d4a6c35d13506718d687a25cb05c74cca1e42501abd83874a67"
    tm = hostapi.xof(136, 0x0B, msgs[:50], 64, rounds=12)
    for i in range(50):
        assert tm[i].tobytes() == orc.sponge_rounds(msgs[i], 64, 136, 0x0B, 12)


# ---- KangarooTwelve (xof/k12) ----

@pytest.mark.gpu
def test_k12_id_vectors_and_oracle():
    from test_oracle_k12 import BIG, VECTORS, ptn
    from oracle import k12 as ok12
    # xof/k12/k12_test.go:46-70, all vectors of one output length in one batch each
    for outlen in (32, 16):
        vs = [v for v in VECTORS if v[2] == outlen]
        got = hostapi.k12([v[0] for v in vs], outlen, [v[1] for v in vs])
        assert [g.tobytes().hex() for g in got] == [v[3] for v in vs]
    # the 24 MB vector (k12_test.go:59): 2946 leaves in one TurboSHAKE128 batch
    assert hostapi.k12([ptn(BIG[0])], 32)[0].tobytes().hex() == BIG[1]
    # random lengths around the chunk boundaries, with and without contexts, long outputs
    rng = np.random.default_rng(12)
    lens = [0, 1, 167, 168, 169, 8190, 8191, 8192, 8193, 16383, 16384, 16385, 40000, 100000]
    msgs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]
    ctxs = [rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8).tobytes() for _ in lens]
    got = hostapi.k12(msgs, 200, ctxs)
    for g, m, c in zip(got, msgs, ctxs):
        assert g.tobytes() == ok12.k12(m, c, 200)
    got = hostapi.k12(msgs, 64)
    for g, m in zip(got, msgs):
        assert g.tobytes() == ok12.k12(m, b"", 64)


# ---- error behaviour of the C ABI on a GPU box ----

@pytest.mark.gpu
def test_abi_error_codes():
    import ctypes as C
    import torch
    from circl_amd import _native as nat
    L = nat.lib()
    n = 64
    ek = torch.zeros((n, 1184), dtype=torch.uint8, device="cuda")
    m = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    ct = torch.empty((n, 1088), dtype=torch.uint8, device="cuda")
    ss = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    st = torch.empty(n, dtype=torch.uint8, device="cuda")
    wsb = L.circl_hip_mlkem_workspace_size(768, n)
    ws = torch.empty(wsb + 64, dtype=torch.uint8, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda param, ekp, wsp, wsbytes: L.circl_hip_mlkem_encaps_dev(param, ekp, m.data_ptr(), ct.data_ptr(), ss.data_ptr(), st.data_ptr(), n, wsp, wsbytes, stream)
    assert call(768, ek.data_ptr(), ws.data_ptr(), wsb) == nat.OK
    assert call(769, ek.data_ptr(), ws.data_ptr(), wsb) == nat.EPARAM                 # unknown parameter set
    # the workspace size is MONOTONE in n and a call only needs the part its own n needs (include/circl_hip.h): without room for the row
    # cache of the small-batch routes the call takes the big-batch routes -- same bytes; below per-item slots + scratch + 128 KB it fails
    torch.cuda.synchronize()
    ct_small = ct.clone()
    assert call(768, ek.data_ptr(), ws.data_ptr(), wsb - 1) == nat.OK
    torch.cuda.synchronize()
    assert (ct == ct_small).all().item()
    assert call(768, ek.data_ptr(), ws.data_ptr(), wsb - 48 * 8192) == nat.OK          # exactly the minimum for 64 items (a 16-entry cache)
    assert call(768, ek.data_ptr(), ws.data_ptr(), wsb - 48 * 8192 - 1) == nat.EWORKSPACE  # workspace too small
    sizes = [L.circl_hip_mlkem_workspace_size(768, k) for k in (1, 2, 17, 1000, 20000, 32768, 32769, 65536, 1 << 20)]
    assert sizes == sorted(sizes)
    assert call(768, ek.data_ptr() + 1, ws.data_ptr(), wsb) == nat.EWORKSPACE          # misaligned input
    assert call(768, ek.data_ptr(), ws.data_ptr() + 8, wsb) == nat.EWORKSPACE          # misaligned workspace
    assert L.circl_hip_mlkem_workspace_size(100, n) == 0 and L.circl_hip_mldsa_workspace_size(1, n) == 0
    assert L.circl_hip_mlkem_ek_size(5) == 0 and L.circl_hip_mldsa_sig_size(66) == 0
    # host-buffer entry points: bad device index, unknown parameter set; n = 0 is a no-op
    a = np.zeros((1, 1184), np.uint8)
    b = np.zeros((1, 32), np.uint8)
    c = np.zeros((1, 1088), np.uint8)
    s1 = np.zeros(1, np.uint8)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    assert L.circl_hip_mlkem_encaps(768, p(a), p(b), p(c), p(b), p(s1), 1, 99) == nat.ENODEV
    assert L.circl_hip_mlkem_encaps(7, p(a), p(b), p(c), p(b), p(s1), 1, 0) == nat.EPARAM
    assert L.circl_hip_mlkem_encaps(768, p(a), p(b), p(c), p(b), p(s1), 0, 0) == nat.OK
    assert L.circl_hip_xof(100, 0x1f, 24, p(a), p(np.zeros(2, np.uint64)), p(c), 32, 1, 0) == nat.EPARAM   # not a sponge rate
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_keccak_coop_equals_lane_per_state_form():
    # the wave-cooperative permutation (a rare serial path of the ML-DSA kernels) against the oracle and the main form
    from circl_amd import hostapi
    from oracle import orc
    rng = np.random.default_rng(41)
    st = rng.integers(0, 2**63, (300, 25), dtype=np.uint64) * 2 + rng.integers(0, 2, (300, 25), dtype=np.uint64)
    st[0] = 0
    a = hostapi.keccak_f1600_coop(st)
    assert (a == hostapi.keccak_f1600(st)).all()
    assert (a[:20] == np.stack([orc.keccak_f1600(s) for s in st[:20]])).all()


@pytest.mark.gpu
def test_keccak_split_equals_lane_per_state_form():
    # the two-lanes-per-state permutation (hashing of small and medium batches) against the oracle and the main form; an odd
    # count leaves a lane pair without a state, 301 states span more than one workgroup
    from circl_amd import hostapi
    from oracle import orc
    rng = np.random.default_rng(43)
    st = rng.integers(0, 2**63, (301, 25), dtype=np.uint64) * 2 + rng.integers(0, 2, (301, 25), dtype=np.uint64)
    st[0] = 0
    st[1] = 0xFFFFFFFFFFFFFFFF
    a = hostapi.keccak_f1600_split(st)
    assert (a == hostapi.keccak_f1600(st)).all()
    assert (a[:20] == np.stack([orc.keccak_f1600(s) for s in st[:20]])).all()
    assert (hostapi.keccak_f1600_split(st[:1]) == a[:1]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("param", [44, 65, 87, 3])
def test_sample_in_ball_both_forms_vs_oracle(param):
    # sample.go:299-339 PolyDeriveUniformBall: the block-parallel form the kernels use, the reference-order scan that is its
    # fallback, and the oracle agree on random challenge seeds
    from circl_amd import hostapi
    from oracle import orc
    rng = np.random.default_rng(param)
    ct = {44: 32, 65: 48, 87: 64, 3: 32}[param]
    seeds = rng.integers(0, 256, (4000, ct), dtype=np.uint8)
    fast = hostapi.mldsa_sample_in_ball(param, seeds)
    slow = hostapi.mldsa_sample_in_ball(param, seeds, sequential=True)
    assert (fast == slow).all()
    tau = {44: 39, 65: 49, 87: 60, 3: 49}[param]
    assert ((fast != 0).sum(axis=1) == tau).all()
    for i in range(0, 4000, 40):
        assert (fast[i] == orc.dilithium_ball(param, seeds[i])).all()
