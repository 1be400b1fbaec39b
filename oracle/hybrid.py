"""TEST INFRASTRUCTURE ONLY (see oracle.h): X-Wing (kem/xwing/xwing.go) and X25519MLKEM768 (kem/hybrid/hybrid.go + xkem.go) restated over
the oracle's SHAKE256 / SHA3-256 / ML-KEM-768 / X25519.  X-Wing is pinned by the draft's transcript digest through
tests/xwing_test.cpp (kem/xwing/xwing_test.go:38-85); the pieces are pinned by the tests/test_oracle_*.py files."""
import numpy as np

from . import orc

LABEL = b"\\.//^\\"


def _shake256(rows, outlen):
    return np.stack([np.frombuffer(orc.sponge(bytes(r), outlen, 136, 0x1f), np.uint8) for r in rows])


def _sha3_256(rows):
    return np.stack([np.frombuffer(orc.sponge(bytes(r), 32, 136, 0x06), np.uint8) for r in rows])


# ---- X-Wing ----
def xwing_keygen(seeds):                                   # xwing.go:107-132
    ex = _shake256(seeds, 96)
    ek, dk = orc.mlkem_keygen(768, ex[:, :64].copy())
    pkx, _ = orc.x25519(ex[:, 64:].copy())
    return np.concatenate([ek, pkx], axis=1), seeds.copy(), dk, ex[:, 64:].copy()


def xwing_encaps(pk, eseeds):                              # xwing.go:223-265
    ek, pkx = pk[:, :1184].copy(), pk[:, 1184:].copy()
    ctm, ssm, st = orc.mlkem_encaps(768, ek, eseeds[:, :32].copy())
    ekx = eseeds[:, 32:].copy()
    ctx, _ = orc.x25519(ekx)
    ssx, _ = orc.x25519(ekx, pkx)
    ss = _sha3_256(np.concatenate([ssm, ssx, ctx, pkx, np.tile(np.frombuffer(LABEL, np.uint8), (len(pk), 1))], axis=1))
    ct = np.concatenate([ctm, ctx], axis=1)
    ss[st != 0] = 0
    ct[st != 0] = 0
    return ct, ss, st


def xwing_decaps(sk, ct):                                  # xwing.go:270-299
    _, _, dk, skx = xwing_keygen(sk)
    ssm, _ = orc.mlkem_decaps(768, dk, ct[:, :1088].copy())
    ctx = ct[:, 1088:].copy()
    ssx, _ = orc.x25519(skx, ctx)
    pkx, _ = orc.x25519(skx)
    return _sha3_256(np.concatenate([ssm, ssx, ctx, pkx, np.tile(np.frombuffer(LABEL, np.uint8), (len(sk), 1))], axis=1))


# ---- X25519MLKEM768 ----
def hybrid_keygen(seeds):                                  # hybrid.go:236-250, xkem.go:112-123
    ex = _shake256(seeds, 96)
    ek, dk = orc.mlkem_keygen(768, ex[:, :64].copy())
    skx = _shake256(ex[:, 64:], 32)
    pkx, _ = orc.x25519(skx)
    return np.concatenate([ek, pkx], axis=1), np.concatenate([dk, skx], axis=1)


def hybrid_encaps(pk, eseeds):                             # hybrid.go:271-300, xkem.go:160-178
    ex = _shake256(eseeds, 64)
    ctm, ssm, st = orc.mlkem_encaps(768, pk[:, :1184].copy(), ex[:, :32].copy())
    skx = _shake256(ex[:, 32:], 32)
    ctx, _ = orc.x25519(skx)
    ssx, ok = orc.x25519(skx, pk[:, 1184:].copy())
    status = np.where((st != 0) | (ok == 0), 1, 0).astype(np.uint8)
    ct, ss = np.concatenate([ctm, ctx], axis=1), np.concatenate([ssm, ssx], axis=1)
    ct[status != 0] = 0
    ss[status != 0] = 0
    return ct, ss, status


def hybrid_decaps(sk, ct):                                 # hybrid.go:302-323, xkem.go:180-196
    ssm, st = orc.mlkem_decaps(768, sk[:, :2400].copy(), ct[:, :1088].copy())
    ssx, ok = orc.x25519(sk[:, 2400:].copy(), ct[:, 1088:].copy())
    status = np.where(st != 0, st, np.where(ok == 0, 1, 0)).astype(np.uint8)
    ss = np.concatenate([ssm, ssx], axis=1)
    ss[status != 0] = 0
    return ss, status
