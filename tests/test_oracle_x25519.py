"""Pins oracle/x25519.c (the checker of the GPU X25519 of SURVEY.md 8(f) row f2) against the vectors the reference's own
tests hold: dh/x25519/key_test.go TestRFC7748Kat (:24-46), TestRFC7748Times (:53-84), TestBase (:100-112),
TestWycheproof (:114-150)."""
import numpy as np

from conftest import load_golden
from oracle import orc

G = load_golden("x25519.json.gz")


def test_rfc7748_kat():
    for v in G["rfc7748_kat"]:
        out, _ = orc.x25519(bytes.fromhex(v["scalar"]), bytes.fromhex(v["input"]))
        assert bytes(out[0]).hex() == v["output"]


def test_rfc7748_times():
    for v in G["rfc7748_times"]:
        if v["times"] > 1000:
            continue  # the reference skips the 10^6 case too unless -long
        u = bytes([9]) + bytes(31)
        k = u
        for _ in range(v["times"]):
            r, _ = orc.x25519(k, u)
            u, k = k, bytes(r[0])
        assert k.hex() == v["key"]


def test_base_point_is_shared_with_nine():
    rng = np.random.default_rng(3)
    s = rng.integers(0, 256, (256, 32), dtype=np.uint8)
    nine = np.zeros((256, 32), np.uint8)
    nine[:, 0] = 9
    assert (orc.x25519(s)[0] == orc.x25519(s, nine)[0]).all()


def test_wycheproof():
    n_low = 0
    for v in G["wycheproof"]:
        out, ok = orc.x25519(bytes.fromhex(v["private"]), bytes.fromhex(v["public"]))
        assert bytes(out[0]).hex() == v["shared"], v["tcId"]
        assert ok[0] or v["result"] == "acceptable", v["tcId"]
        n_low += not ok[0]
    assert n_low > 0


def test_low_order_points_and_their_non_canonical_forms():
    # dh/x25519/curve.go:71-96 + key.go:24-31: the check reduces mod p first, so u + p (where it fits in 255 bits) and
    # a set top bit are caught as well; the shared secret of a low-order point is zero
    p = 2**255 - 19
    lows = [0, 1, p - 1,
            int.from_bytes(bytes.fromhex("e0eb7a7c3b41b8ae1656e3faf19fc46ada098deb9c32b1fd866205165f49b800"), "little"),
            int.from_bytes(bytes.fromhex("5f9c95bca3508c24b1d0b1559c83ef5b04445cc4581c8e86d8224eddd09f1157"), "little")]
    k = bytes(range(32))
    for u in lows:
        encs = [u, u | 1 << 255] + ([u + p, (u + p) | 1 << 255] if u + p < 2**255 else [])
        for enc in encs:
            out, ok = orc.x25519(k, enc.to_bytes(32, "little"))
            assert not ok[0] and not out.any(), hex(enc)
    out, ok = orc.x25519(k, (2).to_bytes(32, "little"))
    assert ok[0] and out.any()
