"""CPU-side checks of the device source itself: the lane-local __host__ __device__ functions of
circl_amd/csrc/*_dev.h are compiled for the host (tests/hostsim/hostsim.hip) and compared with the
oracle / exact arithmetic.  Mirrors the reference's field_test.go / ntt_test.go / poly_test.go."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import load_golden
from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q, DQ = 3329, 8380417


@pytest.fixture(scope="module")
def hs():
    out = os.path.join(ROOT, "build", "libhostsim.so")
    src = os.path.join(ROOT, "tests", "hostsim", "hostsim.hip")
    hdrs = [os.path.join(ROOT, "circl_amd", "csrc", h) for h in ("keccak_dev.h", "kyber_dev.h", "dilithium_dev.h", "x25519_dev.h", "x25519_base_table.h", "lane_ops.h")]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or any(os.path.getmtime(p) > os.path.getmtime(out) for p in [src] + hdrs):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", "-I",
                               os.path.join(ROOT, "circl_amd", "csrc"), src, "-o", out])
    L = C.CDLL(out)
    for f in ("hs_kyber_compress", "hs_kyber_msg_bit", "hs_dil_mont32", "hs_dil_fold", "hs_dil_normalize", "hs_dil_zeta", "hs_dil_r32", "hs_dil_use_hint"):
        getattr(L, f).restype = C.c_uint32
    L.hs_dil_mont32.argtypes = [C.c_uint32, C.c_uint32]
    L.hs_dil_fold.argtypes = [C.c_uint32]
    L.hs_dil_normalize.argtypes = [C.c_uint32]
    L.hs_dil_use_hint.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
    L.hs_dil_exceeds.argtypes = [C.c_uint32, C.c_uint32]
    return L


def test_keccak_host_instantiation(hs):
    want = np.array(load_golden("fixed_vectors.json.gz")["keccak_f1600_of_zero"], dtype=np.uint64)
    a = np.zeros(25, np.uint64)
    hs.hs_keccak_f1600(a.ctypes.data_as(C.c_void_p), 24)
    assert (a == want).all()
    rng = np.random.default_rng(0)
    for rounds in (24, 12):
        st = rng.integers(0, 1 << 63, 25, dtype=np.uint64)
        b = st.copy()
        hs.hs_keccak_f1600(b.ctypes.data_as(C.c_void_p), rounds)
        assert (b == orc.keccak_f1600(st, rounds)).all()


def test_kyber_field_and_zetas(hs):
    # field_test.go: barrettReduce over its whole documented range; the zetas table; and the two-instruction
    # products / reductions with R = 2^32 that replace montReduce on the device (kyber_dev.h)
    z = orc.kyber_zetas()          # the reference's table: zeta^brv * 2^16
    assert [hs.hs_kyber_zeta_plain(i) * 65536 % Q for i in range(128)] == [int(v) % Q for v in z]
    for x in range(-32768, 32768):
        r = hs.hs_kyber_barrett(x)
        assert 0 <= r <= Q and (r - x) % Q == 0
        n = hs.hs_kyber_normalize(x)
        assert n == x % Q
    hs.hs_kyber_mulc.restype = hs.hs_kyber_reduce32.restype = hs.hs_kyber_mulc_limit.restype = C.c_uint32
    hs.hs_kyber_mulc.argtypes = [C.c_uint32, C.c_uint32]
    hs.hs_kyber_reduce32.argtypes = [C.c_uint32]
    lim = hs.hs_kyber_mulc_limit()
    assert lim == (1 << 32) // Q
    rng = np.random.default_rng(1)
    edge = [0, 1, 2, Q - 1, Q, Q + 1, 8 * Q + 3, 128 * Q - 1, lim - 1]
    for w in range(Q):              # every residue as the constant, edge operands
        for b in edge:
            assert hs.hs_kyber_mulc(b, w) == w * b % Q
    for b, w in zip(rng.integers(0, lim, 200000), rng.integers(0, Q, 200000)):
        assert hs.hs_kyber_mulc(int(b), int(w)) == int(w) * int(b) % Q
    inv32 = pow(1 << 32, -1, Q)
    for t in [0, 1, Q, (1 << 31) - 1, 1 << 31, (1 << 32) - 1] + [int(v) for v in rng.integers(0, 1 << 32, 200000, dtype=np.uint64)]:
        assert hs.hs_kyber_reduce32(t) == (-t * inv32) % Q


@pytest.mark.parametrize("d", [4, 5, 10, 11])
def test_compress_exact_for_all_x(hs, d):
    # poly_test.go:351-378, on the extended domain the kernels use: any representative below 4q (2q + 8 for d = 11)
    for x in range(4 * Q if d != 11 else 2 * Q + 8):
        want = (((x % Q) << d) + Q // 2) // Q % (1 << d)
        assert hs.hs_kyber_compress(x, d) == want
    for t in range(1 << d):
        assert hs.hs_kyber_decompress(t, d) == (t * Q + (1 << (d - 1))) >> d
    assert [hs.hs_kyber_msg_bit(x) for x in range(Q)] == [1 if 833 <= x <= 2496 else 0 for x in range(Q)]


def test_cbd_tables(hs):
    assert [hs.hs_kyber_cbd2(t) for t in range(16)] == [bin(t & 3).count("1") - bin(t >> 2).count("1") for t in range(16)]
    assert [hs.hs_kyber_cbd3(t) for t in range(64)] == [bin(t & 7).count("1") - bin(t >> 3).count("1") for t in range(64)]
    # the packed form used by the PRF pass: nibble k of the result is cbd2(nibble k of the word) + 8
    hs.hs_kyber_cbd2_bias8_word.restype = C.c_uint32
    rng = np.random.default_rng(5)
    words = [0, 0xFFFFFFFF, 0x33333333, 0xCCCCCCCC, 0x0F0F0F0F, 0xF0F0F0F0] + [int(x) for x in rng.integers(0, 1 << 32, 2000, dtype=np.uint64)]
    for w in words:
        got = hs.hs_kyber_cbd2_bias8_word(w)
        for k in range(8):
            t = (w >> (4 * k)) & 15
            assert ((got >> (4 * k)) & 15) - 8 == bin(t & 3).count("1") - bin(t >> 2).count("1")


def test_ntt_network_matches_oracle(hs):
    # ntt_test.go:49-109 with the device's layer / layout / twiddle schedule executed on the host, including the
    # element widths of the LDS exchanges and the no-reduction bounds of the lazy arithmetic
    rng = np.random.default_rng(2)
    mx = C.c_uint32()
    inv16 = pow(1 << 16, -1, Q)
    neg_r32 = (-(1 << 32)) % Q

    def run(p, inverse, scale):
        a = np.ascontiguousarray(p, dtype=np.uint32)
        hs.hs_kyber_ntt(a.ctypes.data_as(C.c_void_p), inverse, scale, C.byref(mx))
        return a.astype(np.int64)

    fwd_inputs = [rng.integers(0, Q + 4, 256) for _ in range(30)] + [np.full(256, Q + 3), np.zeros(256, np.int64), np.arange(256) % 2 * (Q + 3)]
    for p in fwd_inputs:
        got = run(p, 0, 0)
        want = orc.kyber_normalize(orc.kyber_ntt(orc.kyber_normalize(p.astype(np.int16))))
        assert (got % Q == want).all()
        assert got.max() < Q + 4 + 7 * Q and mx.value < (1 << 15)     # int16 operands of V_DOT2, 16-bit exchanges
    inv_inputs = [rng.integers(0, Q, 256) for _ in range(30)] + [np.full(256, Q - 1), np.arange(256) % 2 * (Q - 1), (np.arange(256) // 2) % 2 * (Q - 1)]
    for p in inv_inputs:
        ref = orc.kyber_normalize(orc.kyber_invntt(p.astype(np.int16))).astype(np.int64)   # 2^16 times the exact inverse
        got = run(p, 1, 65536)
        assert (got == ref).all() and mx.value < 128 * Q                # canonical output, lazy sums inside the mulc domain
        got = run(p, 1, neg_r32)
        assert (got == ref * inv16 * neg_r32 % Q).all()
    # round trip (ntt_test.go:83-109): InvNTT(NTT(p)) = p * 2^16
    p = rng.integers(0, Q, 256)
    assert (run(orc.kyber_normalize(run(p, 0, 0).astype(np.int16)), 1, 65536) == p * 65536 % Q).all()


def test_mulhat_lane_share(hs):
    # poly_test.go:92-133: the V_DOT2 formulation of the kernels, b lazy (< 2^15), result = -2^-32 times the products
    rng = np.random.default_rng(3)
    a = rng.integers(0, Q, 256).astype(np.int16)
    for b in (rng.integers(0, Q, 256), rng.integers(0, 8 * Q + 4, 256), np.full(256, 8 * Q + 3)):
        plain = orc.kyber_normalize(orc.kyber_mulhat(a, orc.kyber_normalize(b.astype(np.int16)))).astype(np.int64) * 65536 % Q
        want = (-plain * pow(1 << 32, -1, Q)) % Q
        out = (C.c_int * 4)()
        for lane in range(64):
            ai = (C.c_int * 4)(*[int(v) for v in a[4 * lane:4 * lane + 4]])
            bi = (C.c_int * 4)(*[int(v) for v in b[4 * lane:4 * lane + 4]])
            hs.hs_kyber_mulhat4_packed(out, ai, bi, lane)   # V_DOT2 formulation used by the kernels
            assert list(out) == want[4 * lane:4 * lane + 4].tolist()


def test_dilithium_mont32_and_zetas(hs):
    R32 = (1 << 32) % DQ
    assert hs.hs_dil_r32() == R32
    z = orc.dilithium_zetas().astype(np.uint64)   # zeta^brv * 2^32, the reference's table (ntt.go:19-57)
    for i in range(256):
        assert hs.hs_dil_zeta(i) == int(z[i])
    rng = np.random.default_rng(4)
    inv32 = pow(1 << 32, -1, DQ)
    # contract: a b < 2^32 q  ->  result in (0, 2q)
    for a, b in zip(rng.integers(0, 1 << 32, 20000, dtype=np.uint64), rng.integers(0, DQ, 20000)):
        r = hs.hs_dil_mont32(int(a), int(b))
        assert 0 < r < 2 * DQ and r % DQ == int(a) * int(b) * inv32 % DQ
    for a, b in zip(rng.integers(0, 36 * DQ, 20000, dtype=np.uint64), rng.integers(0, 2 * DQ, 20000)):  # lazy operands
        r = hs.hs_dil_mont32(int(a), int(b))
        assert 0 < r < 2 * DQ and r % DQ == int(a) * int(b) * inv32 % DQ
    for x in rng.integers(0, 1 << 32, 20000, dtype=np.uint64):
        f = hs.hs_dil_fold(int(x))
        assert f < 2 * DQ and f % DQ == int(x) % DQ
        assert hs.hs_dil_normalize(int(x)) == int(x) % DQ


def test_dilithium_mont64_lazy_dot_product(hs):
    # the single reduction of a 64-bit lazily accumulated dot product (mac_rows): t 2^-32 mod q in (0, 2q) for t < 2^32 q
    hs.hs_dil_mont64.restype = C.c_uint32
    hs.hs_dil_mont64.argtypes = [C.c_uint64]
    inv32 = pow(1 << 32, -1, DQ)
    rng = np.random.default_rng(12)
    ts = [0, 1, DQ, (1 << 32) - 1, 1 << 32, (DQ << 32) - 1, 7 * (DQ - 1) * (17 * DQ)]
    ts += [int(a) * int(b) for a, b in zip(rng.integers(0, DQ, 3000), rng.integers(0, 17 * DQ, 3000))]
    ts += [int(v) for v in rng.integers(0, DQ << 32, 3000, dtype=np.uint64)]
    for t in ts:
        assert t < (DQ << 32)
        r = hs.hs_dil_mont64(t)
        assert 0 < r < 2 * DQ and r % DQ == t * inv32 % DQ


@pytest.mark.parametrize("g88", [0, 1])
def test_decompose_use_hint_laws(hs, g88):
    # sign/mldsa/mldsa65/internal/rounding_test.go:14-67
    gamma2 = 95232 if g88 else 261888
    alpha = 2 * gamma2
    m = (DQ - 1) // alpha
    a0, a1 = C.c_uint32(), C.c_uint32()
    rng = np.random.default_rng(5)
    for a in np.concatenate([rng.integers(0, DQ, 20000), np.arange(0, 3000), np.arange(DQ - 3000, DQ)]):
        a = int(a)
        hs.hs_dil_decompose(a, g88, C.byref(a0), C.byref(a1))
        r0 = a0.value - DQ if a0.value > (DQ - 1) // 2 else a0.value   # a0 + q representation -> centred
        if a0.value >= DQ:
            r0 = a0.value - DQ
        assert 0 <= a1.value < m or a1.value == 0
        assert (a1.value * alpha + r0 - a) % DQ == 0 and -gamma2 <= r0 <= gamma2
        assert hs.hs_dil_use_hint(a, 0, g88) == a1.value
        up = hs.hs_dil_use_hint(a, 1, g88)
        assert up == ((a1.value + 1) % m if r0 > 0 else (a1.value - 1) % m)
    assert hs.hs_dil_exceeds(5, 6) == 0 and hs.hs_dil_exceeds(DQ - 6, 6) == 1 and hs.hs_dil_exceeds(6, 6) == 1


# ---- x25519_dev.h (the GPU X25519 of SURVEY.md 8(f) row f2): the very source of the kernel, on the host ----
P25519 = 2**255 - 19
_POS = [0, 26, 51, 77, 102, 128, 153, 179, 204, 230]
# the documented bounds: "carried" limbs, and the largest limbs a subtraction of two carried values can produce
_CAR = [(1 << 26) + (1 << 18) if i % 2 == 0 else (1 << 25) + (1 << 18) for i in range(10)]
_SUBMAX = [c + 2 * ((1 << 26) - 1 if i % 2 == 0 else (1 << 25) - 1) for i, c in enumerate(_CAR)]


def _val(limbs):
    return sum(int(x) << p for x, p in zip(limbs, _POS))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def test_x25519_field_arithmetic_at_its_bounds(hs):
    # math/fp25519/fp_test.go's mul / sqr / sub / inv checks against exact integers, with every limb at the bound the
    # ladder can reach (un-carried differences): the 64-bit column sums must not overflow
    rng = np.random.default_rng(0)
    r = np.zeros(10, np.uint32)
    top = np.array(_SUBMAX, dtype=np.uint32)
    for it in range(3000):
        f = top - rng.integers(0, 4, 10).astype(np.uint32) if it % 2 == 0 else np.array([rng.integers(0, m + 1) for m in _SUBMAX], dtype=np.uint32)
        g = top.copy() if it < 10 else np.array([rng.integers(0, m + 1) for m in _SUBMAX], dtype=np.uint32)
        hs.hs_fe_mul(_ptr(r), _ptr(f), _ptr(g))
        assert _val(r) % P25519 == _val(f) * _val(g) % P25519 and all(r[i] <= _CAR[i] for i in range(10))
        hs.hs_fe_sqr(_ptr(r), _ptr(f))
        assert _val(r) % P25519 == _val(f) ** 2 % P25519 and all(r[i] <= _CAR[i] for i in range(10))
        hs.hs_fe_mul_small(_ptr(r), _ptr(f), 121666)
        assert _val(r) % P25519 == _val(f) * 121666 % P25519 and all(r[i] <= _CAR[i] for i in range(10))
        a = np.array([rng.integers(0, m + 1) for m in _CAR], dtype=np.uint32)
        b = np.array(_CAR, dtype=np.uint32) if it < 10 else np.array([rng.integers(0, m + 1) for m in _CAR], dtype=np.uint32)
        hs.hs_fe_sub(_ptr(r), _ptr(a), _ptr(b))
        assert _val(r) % P25519 == (_val(a) - _val(b)) % P25519 and all(r[i] <= _SUBMAX[i] for i in range(10))


def test_x25519_codec_and_inverse(hs):
    rng = np.random.default_rng(1)
    r, w = np.zeros(10, np.uint32), np.zeros(8, np.uint32)
    for it in range(300):
        a = np.array(_CAR, dtype=np.uint32) - np.uint32(it) if it < 40 else np.array([rng.integers(0, m + 1) for m in _CAR], dtype=np.uint32)
        hs.hs_fe_to_words(_ptr(w), _ptr(a))
        assert int.from_bytes(w.tobytes(), "little") == _val(a) % P25519
        hs.hs_fe_inv(_ptr(r), _ptr(a))
        assert _val(r) * _val(a) % P25519 == (1 if _val(a) % P25519 else 0)
    for v in (0, 1, 18, 19, P25519 - 1, P25519, P25519 + 1, P25519 + 18, 2**255 - 1):
        ww = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint32).copy()
        hs.hs_fe_from_words(_ptr(r), _ptr(ww))
        assert _val(r) == v
        hs.hs_fe_to_words(_ptr(w), _ptr(r))
        assert int.from_bytes(w.tobytes(), "little") == v % P25519


def test_x25519_ladder_host_instantiation(hs):
    # dh/x25519/key_test.go: RFC 7748 KATs, Wycheproof, TestBase -- on the device source compiled for the host
    G = load_golden("x25519.json.gz")
    hs.hs_x25519_valid_public.restype = C.c_uint32
    o = np.zeros(32, np.uint8)
    for v in G["rfc7748_kat"]:
        k, u = (np.frombuffer(bytes.fromhex(v[t]), np.uint8).copy() for t in ("scalar", "input"))
        hs.hs_x25519(_ptr(o), _ptr(k), _ptr(u), 0, C.c_size_t(1))
        assert o.tobytes().hex() == v["output"]
    for v in G["wycheproof"]:
        k, u = (np.frombuffer(bytes.fromhex(v[t]), np.uint8).copy() for t in ("private", "public"))
        hs.hs_x25519(_ptr(o), _ptr(k), _ptr(u), 0, C.c_size_t(1))
        assert o.tobytes().hex() == v["shared"], v["tcId"]
        assert hs.hs_x25519_valid_public(_ptr(u)) == orc.x25519(k, u)[1][0]
    rng = np.random.default_rng(2)
    n = 64
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    u = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    out = np.zeros((n, 32), np.uint8)
    hs.hs_x25519(_ptr(out), _ptr(k), _ptr(u), 0, C.c_size_t(n))
    assert (out == orc.x25519(k, u)[0]).all()
    hs.hs_x25519(_ptr(out), _ptr(k), _ptr(u), 1, C.c_size_t(n))
    assert (out == orc.x25519(k)[0]).all()


def test_x25519_fixed_base_comb_host_instantiation(hs):
    # KeyGen through the Edwards comb (x25519_dev.h base_mult + the generated table) == Shared(k, 9): key_test.go:100-112
    rng = np.random.default_rng(3)
    n = 600
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for i, b in enumerate((0, 255, 0x88, 0x77, 0x08, 0xf8, 0x80, 0x7f)):  # digit patterns that exercise every carry / sign case
        k[i] = b
    out = np.zeros((n, 32), np.uint8)
    hs.hs_x25519(_ptr(out), _ptr(k), _ptr(k), 2, C.c_size_t(n))
    assert (out == orc.x25519(k)[0]).all()


def test_x25519_base_table_is_what_the_generator_writes(tmp_path):
    import importlib.util
    import shutil
    spec = importlib.util.spec_from_file_location("gen", os.path.join(ROOT, "tools", "gen_x25519_base_table.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    hdr = os.path.join(ROOT, "circl_amd", "csrc", "x25519_base_table.h")
    keep = tmp_path / "committed.h"
    shutil.copy2(hdr, keep)  # copy2: the time stamp travels too, so the build does not see a changed header afterwards
    try:
        gen.main()
        assert open(hdr).read() == open(keep).read()
    finally:
        shutil.copy2(keep, hdr)


# ---- the laws of tests/lane_laws.py on the HOST instantiation (tests/test_gpu_lane_prims.py runs them on the GPU) ----------
import lane_laws as laws  # noqa: E402


def _host_lane(hs):
    from circl_amd.hostapi import LANE_OPS

    def lane(op, a, b=None, arg=0, two=False):
        a = np.ascontiguousarray(a, dtype=np.uint32).reshape(-1)
        bb = None if b is None else np.ascontiguousarray(np.broadcast_to(np.asarray(b, dtype=np.uint32), a.shape))
        o0 = np.empty(len(a), np.uint32)
        o1 = np.empty(len(a), np.uint32) if two else None
        vp = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)  # noqa: E731
        hs.hs_lane_op_array(LANE_OPS[op], int(arg), vp(a), vp(bb), vp(o0), vp(o1), C.c_size_t(len(a)))
        return (o0, o1) if two else o0
    return lane


@pytest.mark.parametrize("d", [4, 5, 10, 11])
def test_lane_law_compress(hs, d):
    laws.law_compress_is_exact_rounding_for_every_representative(_host_lane(hs), d)
    laws.law_decompress_and_round_trip_bound(_host_lane(hs), d)


def test_lane_laws_kyber(hs):
    lane = _host_lane(hs)
    laws.law_decompress_and_round_trip_bound(lane, 1)
    laws.law_message_bit_normalize_barrett(lane)
    laws.law_mulc_and_reduce32_at_their_bounds(lane)
    laws.law_cbd2_word_and_dot2(lane)


@pytest.mark.parametrize("gamma2", [95232, 261888])
def test_lane_laws_dilithium_rounding(hs, gamma2):
    laws.law_decompose_law_for_every_a(_host_lane(hs), gamma2)
    laws.law_make_hint_use_hint_law(_host_lane(hs), gamma2)


def test_lane_laws_dilithium_field(hs):
    laws.law_power2round_exceeds_normalize(_host_lane(hs))
    laws.law_montgomery_products_at_their_bounds(_host_lane(hs))
