"""Pins oracle/k12.py (KangarooTwelve draft -10) against the I-D vectors the reference tests hold:
xof/k12/k12_test.go:46-70."""
import pytest

from oracle import k12

CHUNK = 8192


def ptn(n):
    return bytes(i % 0xfb for i in range(n))


VECTORS = [
    # (msg, ctx, outlen, hex)  -- k12_test.go:47-61 (I-D test vectors)
    (b"", b"", 32, "1ac2d450fc3b4205d19da7bfca1b37513c0803577ac7167f06fe2ce1f0ef39e5"),
    (ptn(17), b"", 32, "6bf75fa2239198db4772e36478f8e19b0f371205f6a9a93a273f51df37122888"),
    (ptn(17**2), b"", 32, "0c315ebcdedbf61426de7dcf8fb725d1e74675d7f5327a5067f367b108ecb67c"),
    (ptn(17**3), b"", 32, "cb552e2ec77d9910701d578b457ddf772c12e322e4ee7fe417f92c758f0d59d0"),
    (ptn(17**4), b"", 32, "8701045e22205345ff4dda05555cbb5c3af1a771c2b89baef37db43d9998b9fe"),
    (ptn(17**5), b"", 32, "844d610933b1b9963cbdeb5ae3b6b05cc7cbd67ceedf883eb678a0a8e0371682"),
    (b"", ptn(1), 32, "fab658db63e94a246188bf7af69a133045f46ee984c56e3c3328caaf1aa1a583"),
    (b"\xff", ptn(41), 32, "d848c5068ced736f4462159b9867fd4c20b808acc3d5bc48e0b06ba0a3762ec4"),
    (b"\xff" * 3, ptn(41**2), 32, "c389e5009ae57120854c2e8c64670ac01358cf4c1baf89447a724234dc7ced74"),
    (b"\xff" * 7, ptn(41**3), 32, "75d2f86a2e644566726b4fbcfc5657b9dbcf070c7b0dca06450ab291d7443bcf"),
    # corner cases, k12_test.go:64-69
    (ptn(CHUNK), b"", 16, "48f256f6772f9edfb6a8b661ec92dc93"),
    (ptn(CHUNK + 1), b"", 16, "bb66fe72eaea5179418d5295ee134485"),
    (ptn(2 * CHUNK), b"", 16, "82778f7f7234c83352e76837b721fbdb"),
    (ptn(2 * CHUNK + 1), b"", 16, "5f8d2b943922b451842b4e82740d0236"),
    (ptn(3 * CHUNK), b"", 16, "f4082a8fe7d1635aa042cd1da63bf235"),
    (ptn(3 * CHUNK + 1), b"", 16, "38cb940999aca742d69dd79298c6051c"),
]
# ptn(17**6) = 24 MB: k12_test.go:59 -- kept for the GPU path (tests/test_gpu_prims.py), too slow through ctypes one-shots here
BIG = (17**6, "3c390782a8a4e89fa6367f72feaaf13255c8d95878481d3cd8ce85f58e880af8")


@pytest.mark.parametrize("idx", range(len(VECTORS)))
def test_k12_id_vectors(idx):
    msg, ctx, outlen, want = VECTORS[idx]
    assert k12.k12(msg, ctx, outlen).hex() == want


def test_length_encode():
    assert k12.length_encode(0) == b"\x00"
    assert k12.length_encode(12) == b"\x0c\x01"
    assert k12.length_encode(65538) == b"\x01\x00\x02\x03"
