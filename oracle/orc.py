"""ctypes binding of oracle/liborc.so (TEST INFRASTRUCTURE ONLY).

The oracle is a scalar C restatement of cloudflare/circl's generic Go ML-KEM / ML-DSA
path (see oracle/kyber.c, oracle/dilithium.c, oracle/keccak.c for file:line citations).
It is the *checker* for the HIP path and the reported CPU baseline; it is never the
thing shipped or measured as the product.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liborc.so")
_VLIB = os.path.join(_HERE, "liborcvec.so")  # oracle/vec: the batch-vectorised encapsulation of bench.py's cpu_baseline

KEM_SIZES = {512: (800, 1632, 768), 768: (1184, 2400, 1088), 1024: (1568, 3168, 1568)}  # ek, dk, ct
DSA_SIZES = {44: (1312, 2560, 2420), 65: (1952, 4032, 3309), 87: (2592, 4896, 4627),  # pk, sk, sig
             # round-3 Dilithium2/3/5 (sign/dilithium/mode{2,3,5}): tr is 32 bytes, c~ is 32 bytes
             2: (1312, 2528, 2420), 3: (1952, 4000, 3293), 5: (2592, 4864, 4595)}


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("keccak.c", "kyber.c", "dilithium.c", "batch.c", "x25519.c", "keccak.h", "oracle.h")]
    vsrcs = [os.path.join(_HERE, "vec", f) for f in ("mlkem_vec.c", "dispatch.c")] + [os.path.join(_HERE, "Makefile")]
    if (force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
            or not os.path.exists(_VLIB) or any(os.path.getmtime(s) > os.path.getmtime(_VLIB) for s in vsrcs)):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_mlkem_ek_size.restype = C.c_size_t
        _lib.orc_mlkem_dk_size.restype = C.c_size_t
        _lib.orc_mlkem_ct_size.restype = C.c_size_t
        _lib.orc_mldsa_pk_size.restype = C.c_size_t
        _lib.orc_mldsa_sk_size.restype = C.c_size_t
        _lib.orc_mldsa_sig_size.restype = C.c_size_t
        _lib.orc_kyber_zetas.restype = C.POINTER(C.c_int16)
        _lib.orc_dilithium_zetas.restype = C.POINTER(C.c_uint32)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(x, shape=None):
    a = np.ascontiguousarray(np.frombuffer(x, dtype=np.uint8) if isinstance(x, (bytes, bytearray)) else x, dtype=np.uint8)
    if shape is not None:
        a = a.reshape(shape)
    return a


def ncpu():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container that sees
    256 hardware threads may be limited to 16 CPUs' worth of time; oversubscribing it makes the threaded oracle slower)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // p)))
        except Exception:
            pass
    return n


# ---------------- hashes ----------------
def keccak_f1600(state, rounds=24):
    a = np.array(state, dtype=np.uint64).copy()
    assert a.shape == (25,)
    lib().orc_keccak_f1600(_p(a), C.c_int(rounds))
    return a


def sponge(data, outlen, rate, ds):
    d = _u8(bytes(data))
    out = np.zeros(outlen, np.uint8)
    lib().orc_sponge_oneshot(_p(out), C.c_size_t(outlen), _p(d), C.c_size_t(len(d)), C.c_uint(rate), C.c_uint8(ds))
    return out.tobytes()


def sponge_rounds(data, outlen, rate, ds, rounds):
    d = _u8(bytes(data) + b"\0")
    out = np.zeros(outlen, np.uint8)
    lib().orc_sponge_oneshot_rounds(_p(out), C.c_size_t(outlen), _p(d), C.c_size_t(len(data)), C.c_uint(rate), C.c_uint8(ds), C.c_int(rounds))
    return out.tobytes()


def sha3_256(d):
    return sponge(d, 32, 136, 0x06)


def sha3_512(d):
    return sponge(d, 64, 72, 0x06)


def shake128(d, n):
    return sponge(d, n, 168, 0x1F)


def shake256(d, n):
    return sponge(d, n, 136, 0x1F)


# ---------------- ML-KEM ----------------
def mlkem_keygen(param, seeds, threads=None):
    """seeds: (n,64) uint8 (d||z) -> ek (n,EK), dk (n,DK)"""
    EK, DK, _ = KEM_SIZES[param]
    seeds = _u8(seeds).reshape(-1, 64)
    n = len(seeds)
    ek = np.zeros((n, EK), np.uint8)
    dk = np.zeros((n, DK), np.uint8)
    r = lib().orc_mlkem_keygen_batch(param, _p(seeds), _p(ek), _p(dk), C.c_size_t(n), threads or ncpu())
    assert r == 0
    return ek, dk


def mlkem_encaps(param, ek, m, threads=None):
    """-> ct (n,CT), ss (n,32), status (n,)"""
    EK, _, CT = KEM_SIZES[param]
    ek = _u8(ek).reshape(-1, EK)
    m = _u8(m).reshape(-1, 32)
    n = len(ek)
    assert len(m) == n
    ct = np.zeros((n, CT), np.uint8)
    ss = np.zeros((n, 32), np.uint8)
    st = np.zeros(n, np.uint8)
    r = lib().orc_mlkem_encaps_batch(param, _p(ek), _p(m), _p(ct), _p(ss), _p(st), C.c_size_t(n), threads or ncpu())
    assert r == 0
    return ct, ss, st


_vlib = None


def vec_isa(want=0):
    """2 = AVX-512 (F/BW/VL/DQ/VBMI/VBMI2), 1 = AVX2, 0 = this CPU has neither (oracle/vec/dispatch.c); `want` caps it."""
    global _vlib
    if _vlib is None:
        build()
        _vlib = C.CDLL(_VLIB)
    return int(_vlib.orcv_isa(int(want)))


def mlkem_encaps_vec(param, ek, m, threads=None, isa=0, shared=False):
    """The batch-vectorised CPU encapsulation (oracle/vec/mlkem_vec.c: W items per vector, Keccak on 4 / 8 states) -- ML-KEM-768 / -1024,
    same bytes as mlkem_encaps (tests/test_oracle_vec.py).  shared: ONE key (ek of one row) for all messages, parsed once per thread --
    the shape of mlkem_encaps_shared.  Only bench.py's cpu_baseline times it. -> ct, ss, status"""
    EK, _, CT = KEM_SIZES[param]
    ek = _u8(ek).reshape(-1, EK)
    m = _u8(m).reshape(-1, 32)
    n = len(m)
    assert len(ek) == (1 if shared else n)
    if not vec_isa(isa):
        raise RuntimeError("oracle/vec needs AVX2")
    ct = np.zeros((n, CT), np.uint8)
    ss = np.zeros((n, 32), np.uint8)
    st = np.zeros(n, np.uint8)
    r = _vlib.orcv_mlkem_encaps2(param, _p(ek), 1 if shared else 0, _p(m), _p(ct), _p(ss), _p(st), C.c_size_t(n), threads or ncpu(), int(isa))
    assert r == 0, r
    return ct, ss, st


def vec_keccak_f1600(states, isa):
    """Keccak-f[1600] of oracle/vec on N = 4 (isa 1) / 8 (isa 2) states at once; states: (N, 25) uint64 -> (N, 25)"""
    vec_isa()
    n = int(_vlib.orcv_states(int(isa)))
    a = np.ascontiguousarray(np.asarray(states, dtype=np.uint64).reshape(n, 25).T)  # word-major
    _vlib.orcv_f1600(int(isa), _p(a))
    return np.ascontiguousarray(a.T)


def mlkem_encaps_shared(param, ek, m, threads=None):
    """n encapsulations to ONE parsed key (the reference's BenchmarkEncapsulate shape) -> ct (n,CT), ss (n,32)"""
    EK, _, CT = KEM_SIZES[param]
    ek = _u8(ek).reshape(-1, EK)
    assert len(ek) == 1
    m = _u8(m).reshape(-1, 32)
    n = len(m)
    ct = np.zeros((n, CT), np.uint8)
    ss = np.zeros((n, 32), np.uint8)
    r = lib().orc_mlkem_encaps_shared_batch(param, _p(ek), _p(m), _p(ct), _p(ss), C.c_size_t(n), threads or ncpu())
    assert r == 0
    return ct, ss


def mlkem_decaps(param, dk, ct, threads=None):
    _, DK, CT = KEM_SIZES[param]
    dk = _u8(dk).reshape(-1, DK)
    ct = _u8(ct).reshape(-1, CT)
    n = len(dk)
    assert len(ct) == n
    ss = np.zeros((n, 32), np.uint8)
    st = np.zeros(n, np.uint8)
    r = lib().orc_mlkem_decaps_batch(param, _p(dk), _p(ct), _p(ss), _p(st), C.c_size_t(n), threads or ncpu())
    assert r == 0
    return ss, st


# ---------------- round-3 Kyber (kem/kyber/kyber{512,768,1024}) ----------------
def kyber_r3_keygen(param, seeds, threads=None):
    """seeds: (n,64) uint8 -> ek (n,EK), dk (n,DK); param 512 | 768 | 1024"""
    EK, DK, _ = KEM_SIZES[param]
    seeds = _u8(seeds).reshape(-1, 64)
    n = len(seeds)
    ek = np.zeros((n, EK), np.uint8)
    dk = np.zeros((n, DK), np.uint8)
    assert lib().orc_kyber_r3_keygen_batch(param, _p(seeds), _p(ek), _p(dk), C.c_size_t(n), threads or ncpu()) == 0
    return ek, dk


def kyber_r3_encaps(param, ek, seeds, threads=None):
    """-> ct (n,CT), ss (n,32); never fails (non-canonical keys are reduced)"""
    EK, _, CT = KEM_SIZES[param]
    ek = _u8(ek).reshape(-1, EK)
    seeds = _u8(seeds).reshape(-1, 32)
    n = len(ek)
    assert len(seeds) == n
    ct = np.zeros((n, CT), np.uint8)
    ss = np.zeros((n, 32), np.uint8)
    assert lib().orc_kyber_r3_encaps_batch(param, _p(ek), _p(seeds), _p(ct), _p(ss), C.c_size_t(n), threads or ncpu()) == 0
    return ct, ss


def kyber_r3_decaps(param, dk, ct, threads=None):
    _, DK, CT = KEM_SIZES[param]
    dk = _u8(dk).reshape(-1, DK)
    ct = _u8(ct).reshape(-1, CT)
    n = len(dk)
    assert len(ct) == n
    ss = np.zeros((n, 32), np.uint8)
    assert lib().orc_kyber_r3_decaps_batch(param, _p(dk), _p(ct), _p(ss), C.c_size_t(n), threads or ncpu()) == 0
    return ss


# Kyber ring primitives (single polynomial, int16[256])
def _poly16(p):
    a = np.array(p, dtype=np.int16).copy()
    assert a.shape == (256,)
    return a


def kyber_ntt(p):
    a = _poly16(p); lib().orc_kyber_ntt(_p(a)); return a


def kyber_invntt(p):
    a = _poly16(p); lib().orc_kyber_invntt(_p(a)); return a


def kyber_normalize(p):
    a = _poly16(p); lib().orc_kyber_normalize(_p(a)); return a


def kyber_mulhat(a, b):
    a = _poly16(a); b = _poly16(b); r = np.zeros(256, np.int16)
    lib().orc_kyber_mulhat(_p(r), _p(a), _p(b)); return r


def kyber_noise(seed, nonce, eta):
    s = _u8(bytes(seed)); r = np.zeros(256, np.int16)
    lib().orc_kyber_noise(_p(r), _p(s), C.c_uint8(nonce), C.c_int(eta)); return r


def kyber_uniform(seed, x, y):
    s = _u8(bytes(seed)); r = np.zeros(256, np.int16)
    lib().orc_kyber_uniform(_p(r), _p(s), C.c_uint8(x), C.c_uint8(y)); return r


def kyber_compress(p, d):
    a = _poly16(p); m = np.zeros(32 * d, np.uint8)
    lib().orc_kyber_compress(_p(m), _p(a), C.c_int(d)); return m


def kyber_decompress(m, d):
    m = _u8(bytes(m)); r = np.zeros(256, np.int16)
    lib().orc_kyber_decompress(_p(r), _p(m), C.c_int(d)); return r


def kyber_zetas():
    return np.ctypeslib.as_array(lib().orc_kyber_zetas(), shape=(128,)).copy()


# ---------------- ML-DSA ----------------
def _blob(items):
    off = np.zeros(len(items) + 1, np.uint64)
    off[1:] = np.cumsum([len(x) for x in items])
    blob = np.frombuffer(b"".join(bytes(x) for x in items) + b"\0", dtype=np.uint8).copy()
    return blob, off


def mldsa_keygen(param, seeds, threads=None):
    PK, SK, _ = DSA_SIZES[param]
    seeds = _u8(seeds).reshape(-1, 32)
    n = len(seeds)
    pk = np.zeros((n, PK), np.uint8)
    sk = np.zeros((n, SK), np.uint8)
    r = lib().orc_mldsa_keygen_batch(param, _p(seeds), _p(pk), _p(sk), C.c_size_t(n), threads or ncpu())
    assert r == 0
    return pk, sk


def mldsa_sign(param, sk, msgs, ctxs=None, rnd=None, threads=None):
    """deterministic when rnd is None (32 zero bytes per item)"""
    _, SK, SIG = DSA_SIZES[param]
    sk = _u8(sk).reshape(-1, SK)
    n = len(sk)
    mb, mo = _blob(msgs)
    cb, co = _blob(ctxs if ctxs is not None else [b""] * n)
    rnd = np.zeros((n, 32), np.uint8) if rnd is None else _u8(rnd).reshape(n, 32)
    sig = np.zeros((n, SIG), np.uint8)
    r = lib().orc_mldsa_sign_batch(param, _p(sk), _p(mb), _p(mo), _p(cb), _p(co), _p(rnd), _p(sig), C.c_size_t(n), threads or ncpu())
    assert r == 0
    return sig


def mldsa_verify(param, pk, sig, msgs, ctxs=None, threads=None):
    PK, _, SIG = DSA_SIZES[param]
    pk = _u8(pk).reshape(-1, PK)
    sig = _u8(sig).reshape(-1, SIG)
    n = len(pk)
    mb, mo = _blob(msgs)
    cb, co = _blob(ctxs if ctxs is not None else [b""] * n)
    ok = np.zeros(n, np.uint8)
    r = lib().orc_mldsa_verify_batch(param, _p(pk), _p(sig), _p(mb), _p(mo), _p(cb), _p(co), _p(ok), C.c_size_t(n), threads or ncpu())
    assert r == 0
    return ok


def mldsa_sign_one(param, sk, msg, ctx=b"", rnd=bytes(32), internal=False):
    _, SK, SIG = DSA_SIZES[param]
    sk = _u8(bytes(sk)); msg_a = _u8(bytes(msg) + b"\0"); ctx_a = _u8(bytes(ctx) + b"\0"); rnd = _u8(bytes(rnd))
    sig = np.zeros(SIG, np.uint8)
    r = lib().orc_mldsa_sign(param, _p(sk), _p(msg_a), C.c_size_t(len(msg)), _p(ctx_a), C.c_size_t(len(ctx)), _p(rnd), C.c_int(int(internal)), _p(sig))
    assert r == 0, r
    return sig.tobytes()


def mldsa_verify_one(param, pk, msg, sig, ctx=b"", internal=False):
    pk = _u8(bytes(pk)); msg_a = _u8(bytes(msg) + b"\0"); ctx_a = _u8(bytes(ctx) + b"\0"); sig_a = _u8(bytes(sig) + b"\0")
    r = lib().orc_mldsa_verify(param, _p(pk), _p(msg_a), C.c_size_t(len(msg)), _p(ctx_a), C.c_size_t(len(ctx)), C.c_int(int(internal)), _p(sig_a), C.c_size_t(len(sig)))
    return bool(r)


def _poly32(p):
    a = np.array(p, dtype=np.uint32).copy()
    assert a.shape == (256,)
    return a


def dilithium_ntt(p):
    a = _poly32(p); lib().orc_dilithium_ntt(_p(a)); return a


def dilithium_invntt(p):
    a = _poly32(p); lib().orc_dilithium_invntt(_p(a)); return a


def dilithium_normalize(p):
    a = _poly32(p); lib().orc_dilithium_normalize(_p(a)); return a


def dilithium_uniform(seed, nonce):
    s = _u8(bytes(seed)); r = np.zeros(256, np.uint32)
    lib().orc_dilithium_uniform(_p(r), _p(s), C.c_uint16(nonce)); return r


def dilithium_ball(param, ctilde):
    """PolyDeriveUniformBall(c~) -> uint32[256] (sample.go:299-339)"""
    c = _u8(bytes(ctilde)); r = np.zeros(256, np.uint32)
    assert lib().orc_dilithium_ball(param, _p(r), _p(c)) == 0
    return r


def dilithium_zetas():
    return np.ctypeslib.as_array(lib().orc_dilithium_zetas(), shape=(256,)).copy()


# ---------------- X25519 (dh/x25519) ----------------
def x25519(scalar, point=None):
    """Batch X25519: scalar (n,32), point (n,32) or None for the base point (KeyGen).  Returns (out (n,32), ok (n,))
    with ok = 0 where the reference's Shared reports a low-order public key."""
    scalar = _u8(scalar).reshape(-1, 32)
    n = scalar.shape[0]
    out = np.zeros((n, 32), np.uint8)
    ok = np.zeros(n, np.uint8)
    pt = None if point is None else _u8(point).reshape(n, 32)
    lib().orc_x25519_batch(_p(scalar), None if pt is None else _p(pt), _p(out), _p(ok), C.c_size_t(n))
    return out, ok
