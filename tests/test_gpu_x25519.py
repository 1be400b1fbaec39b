"""Batch X25519 on the GPU (replaces dh/x25519) through the C ABI, against the reference's own
vectors (dh/x25519/key_test.go: TestRFC7748Kat, TestRFC7748Times, TestBase, TestWycheproof) and the oracle."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from circl_amd import hostapi
    return hostapi


@pytest.fixture(scope="module")
def orc():
    from oracle import orc
    return orc


def _rows(hexes):
    return np.frombuffer(b"".join(bytes.fromhex(h) for h in hexes), np.uint8).reshape(-1, 32).copy()


def test_rfc7748_kat(api):
    G = load_golden("x25519.json.gz")["rfc7748_kat"]
    out, ok = api.x25519(_rows(v["scalar"] for v in G), _rows(v["input"] for v in G))
    assert [bytes(r).hex() for r in out] == [v["output"] for v in G] and ok.all()


def test_rfc7748_times(api):
    # key_test.go:53-84: iterate k, u = X25519(k, u), k; one lane of a wave per call -- latency, not throughput
    want = {v["times"]: v["key"] for v in load_golden("x25519.json.gz")["rfc7748_times"]}
    u = np.zeros((1, 32), np.uint8)
    u[0, 0] = 9
    k = u.copy()
    for t in range(1, 1001):
        r, _ = api.x25519(k, u)
        u, k = k, r
        if t in want:
            assert bytes(k[0]).hex() == want[t]


def test_wycheproof(api):
    G = load_golden("x25519.json.gz")["wycheproof"]
    out, ok = api.x25519(_rows(v["private"] for v in G), _rows(v["public"] for v in G))
    for v, r, o in zip(G, out, ok):
        assert bytes(r).hex() == v["shared"], v["tcId"]
        assert o or v["result"] == "acceptable", v["tcId"]
    assert not ok.all()


def test_low_order_points_and_non_canonical_forms(api, orc):
    p = 2**255 - 19
    lows = [0, 1, p - 1,
            int.from_bytes(bytes.fromhex("e0eb7a7c3b41b8ae1656e3faf19fc46ada098deb9c32b1fd866205165f49b800"), "little"),
            int.from_bytes(bytes.fromhex("5f9c95bca3508c24b1d0b1559c83ef5b04445cc4581c8e86d8224eddd09f1157"), "little")]
    encs = []
    for u in lows:
        encs += [u, u | 1 << 255] + ([u + p, (u + p) | 1 << 255] if u + p < 2**255 else [])
    others = [2, 9, p - 2, p + 2, 2**255 - 1, (1 << 255) | 9, p + 18]
    pts = np.frombuffer(b"".join(e.to_bytes(32, "little") for e in encs + others), np.uint8).reshape(-1, 32).copy()
    k = np.tile(np.arange(32, dtype=np.uint8), (len(pts), 1))
    out, ok = api.x25519(k, pts)
    assert not ok[:len(encs)].any() and not out[:len(encs)].any()
    assert ok[len(encs):].all() and out[len(encs):].any(axis=1).all()
    o2, k2 = orc.x25519(k, pts)
    assert (out == o2).all() and (ok == k2).all()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 20000])
def test_against_oracle_and_base_point(api, orc, n):
    rng = np.random.default_rng(n)
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    u = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if n >= 64:  # extreme scalars and points in the mix
        k[0] = 0
        k[1] = 255
        u[2] = 255
        u[3] = 0
        u[3, 0] = 9
    out, ok = api.x25519(k, u)
    o2, k2 = orc.x25519(k, u)
    assert (out == o2).all() and (ok == k2).all()
    pub, okb = api.x25519(k)                      # KeyGen: key_test.go:100-112 TestBase
    assert (pub == orc.x25519(k)[0]).all() and okb.all()
    nine = np.zeros((n, 32), np.uint8)
    nine[:, 0] = 9
    assert (api.x25519(k, nine)[0] == pub).all()


def test_diffie_hellman_agreement_large_batch(api):
    # size-independent property at a batch that fills the chip: X25519(a, X25519(b, 9)) == X25519(b, X25519(a, 9))
    n = 1 << 18
    rng = np.random.default_rng(7)
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pa, _ = api.x25519(a)
    pb, _ = api.x25519(b)
    s1, ok1 = api.x25519(a, pb)
    s2, ok2 = api.x25519(b, pa)
    assert (s1 == s2).all() and ok1.all() and ok2.all() and s1.any(axis=1).all()


def test_misaligned_host_buffers(api, orc):
    import ctypes as C
    from circl_amd import _native as nat
    n = 3000
    rng = np.random.default_rng(5)
    raw_k = np.zeros(n * 32 + 3, np.uint8)
    raw_u = np.zeros(n * 32 + 5, np.uint8)
    raw_o = np.zeros(n * 32 + 1, np.uint8)
    k, u, o = raw_k[3:].reshape(n, 32), raw_u[5:].reshape(n, 32), raw_o[1:].reshape(n, 32)
    k[:] = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    u[:] = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ok = np.zeros(n, np.uint8)
    vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    nat.check(nat.lib().circl_hip_x25519(vp(k), vp(u), vp(o), vp(ok), n, 0), "x25519")
    o2, k2 = orc.x25519(k.copy(), u.copy())
    assert (o == o2).all() and (ok == k2).all()
