"""oracle/k12.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

KangarooTwelve draft -10 restated from xof/k12/k12.go on top of the oracle's TurboSHAKE128
(oracle/keccak.c orc_sponge_oneshot_rounds, 12 rounds):

  S = M || C || length_encode(|C|)                                  k12.go:331-343 (Read: context + its length)
  |S| <= 8192:  TurboSHAKE128(S, D=0x07, L)                         k12.go:60-66  (stalk created with 0x07)
  otherwise:    chunks S_0 (8192 B), S_1, ...;  CV_i = TurboSHAKE128(S_i, 0x0B, 32)       k12.go:136-140, :150-160
                node = S_0 || 03 00 00 00 00 00 00 00 || CV_1 .. CV_{n-1} || length_encode(n-1) || FF FF
                TurboSHAKE128(node, D=0x06, L)                      k12.go:141-142, :383-395
  length_encode(x) = big-endian bytes of x without leading zeros, then their count          k12.go:333-342

Pinned by the I-D test vectors of xof/k12/k12_test.go:46-70 (tests/test_oracle_k12.py).
"""
from oracle import orc

CHUNK = 8192


def length_encode(x: int) -> bytes:
    b = x.to_bytes(8, "big").lstrip(b"\0")
    return b + bytes([len(b)])


def turboshake128(data: bytes, ds: int, outlen: int) -> bytes:
    return orc.sponge_rounds(data, outlen, 168, ds, 12)


def k12(msg: bytes, ctx: bytes, outlen: int) -> bytes:
    s = bytes(msg) + bytes(ctx) + length_encode(len(ctx))
    if len(s) <= CHUNK:
        return turboshake128(s, 0x07, outlen)
    node = bytearray(s[:CHUNK]) + b"\x03" + b"\0" * 7
    n = 0
    for off in range(CHUNK, len(s), CHUNK):
        node += turboshake128(s[off:off + CHUNK], 0x0B, 32)
        n += 1
    node += length_encode(n) + b"\xff\xff"
    return turboshake128(bytes(node), 0x06, outlen)
