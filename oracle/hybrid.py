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


# ---- kem/hybrid's concatenation scheme: X25519MLKEM768 (ML-KEM-768 first), Kyber768-X25519 / Kyber512-X25519 (X25519 first,
# round-3 Kyber second): hybrid.go:63-99 ----
SCHEMES = {2: dict(param=768, r3=False, x_first=False), 3: dict(param=768, r3=True, x_first=True), 4: dict(param=512, r3=True, x_first=True)}


def _kem(param, r3):
    ek, dk, ct = orc.KEM_SIZES[param]
    if r3:
        return ek, dk, ct, orc.kyber_r3_keygen, (lambda e, m: orc.kyber_r3_encaps(param, e, m) + (np.zeros(len(e), np.uint8),)), \
            (lambda d, c: (orc.kyber_r3_decaps(param, d, c), np.zeros(len(d), np.uint8)))
    return ek, dk, ct, orc.mlkem_keygen, (lambda e, m: orc.mlkem_encaps(param, e, m)), (lambda d, c: orc.mlkem_decaps(param, d, c))


def _cat(x_first, kem_part, x_part):
    return np.concatenate([x_part, kem_part] if x_first else [kem_part, x_part], axis=1)


def _split(x_first, rows, kem_bytes):
    return (rows[:, 32:].copy(), rows[:, :32].copy()) if x_first else (rows[:, :kem_bytes].copy(), rows[:, kem_bytes:].copy())


def hybrid_keygen(seeds, scheme=2):                        # hybrid.go:236-250, xkem.go:112-123
    S = SCHEMES[scheme]
    EK, DK, CT, keygen, _, _ = _kem(S["param"], S["r3"])
    ex = _shake256(seeds, 96)
    kseed, xseed = (ex[:, 32:], ex[:, :32]) if S["x_first"] else (ex[:, :64], ex[:, 64:])
    ek, dk = keygen(S["param"], kseed.copy())
    skx = _shake256(xseed, 32)
    pkx, _ = orc.x25519(skx)
    return _cat(S["x_first"], ek, pkx), _cat(S["x_first"], dk, skx)


def hybrid_encaps(pk, eseeds, scheme=2):                   # hybrid.go:271-300, xkem.go:160-178
    S = SCHEMES[scheme]
    EK, DK, CT, _, encaps, _ = _kem(S["param"], S["r3"])
    ex = _shake256(eseeds, 64)
    m, xseed = (ex[:, 32:], ex[:, :32]) if S["x_first"] else (ex[:, :32], ex[:, 32:])
    ek, pkx = _split(S["x_first"], pk, EK)
    ctm, ssm, st = encaps(ek, m.copy())
    skx = _shake256(xseed, 32)
    ctx, _ = orc.x25519(skx)
    ssx, ok = orc.x25519(skx, pkx)
    status = np.where((st != 0) | (ok == 0), 1, 0).astype(np.uint8)
    ct, ss = _cat(S["x_first"], ctm, ctx), _cat(S["x_first"], ssm, ssx)
    ct[status != 0] = 0
    ss[status != 0] = 0
    return ct, ss, status


def hybrid_decaps(sk, ct, scheme=2):                       # hybrid.go:302-323, xkem.go:180-196
    S = SCHEMES[scheme]
    EK, DK, CT, _, _, decaps = _kem(S["param"], S["r3"])
    dk, skx = _split(S["x_first"], sk, DK)
    ctm, ctx = _split(S["x_first"], ct, CT)
    ssm, st = decaps(dk, ctm)
    ssx, ok = orc.x25519(skx, ctx)
    status = np.where(st != 0, st, np.where(ok == 0, 1, 0)).astype(np.uint8)
    ss = _cat(S["x_first"], ssm, ssx)
    ss[status != 0] = 0
    return ss, status
