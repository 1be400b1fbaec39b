"""Coefficient-level laws of the Kyber / Dilithium device functions as checks over a `lane(op, a, b=None, arg=0, two=False)`
callable: tests/test_gpu_lane_prims.py passes the DEVICE instantiation (circl_hip_lane_op on the GPU), tests/test_hostsim.py the
HOST instantiation of the same source.  Expectations are exact integer arithmetic (numpy / Python ints), never the oracle.

Reference tests mirrored: pke/kyber/internal/common/poly_test.go:351-378 (compress = exact rounding, decompress bound),
field_test.go (Montgomery / Barrett), sign/mldsa/mldsa65/internal/rounding_test.go:14-67 (decompose, makeHint / useHint),
sign/internal/dilithium/field_test.go (Montgomery reduction, ReduceLe2Q, power2round)."""
import numpy as np

Q = 3329
DQ = 8380417

def law_compress_is_exact_rounding_for_every_representative(lane, d):
    # every representative the kernels may hand over: [0, 4q) (d = 11: [0, 2q + 8), the documented domain of compress_coeff<11>:
    # an inverse transform's output below q plus a noise term q + (-2..2))
    top = 4 * Q if d < 11 else 2 * Q + 8
    x = np.arange(top, dtype=np.uint32)
    got = lane("KYBER_COMPRESS", x, arg=d)
    want = (((x.astype(np.uint64) << d) + Q // 2) // Q) & ((1 << d) - 1)      # poly.go:248-332: round half up, mod 2^d
    assert (got == want).all()
    # and the value only depends on x mod q
    assert (got == got[x % Q]).all()


def law_decompress_and_round_trip_bound(lane, d):
    t = np.arange(1 << d, dtype=np.uint32)
    got = lane("KYBER_DECOMPRESS", t, arg=d)
    assert (got == ((1 << (d - 1)) + t.astype(np.uint64) * Q) >> d).all()
    if d > 1:  # poly_test.go:351-378: |x - Decompress(Compress(x))| mod+- q <= round(q / 2^(d+1))
        x = np.arange(Q, dtype=np.uint32)
        back = lane("KYBER_DECOMPRESS", lane("KYBER_COMPRESS", x, arg=d), arg=d).astype(np.int64)
        diff = (x.astype(np.int64) - back) % Q
        diff = np.minimum(diff, Q - diff)
        assert diff.max() <= (Q + (1 << d)) >> (d + 1)


def law_message_bit_normalize_barrett(lane):
    x = np.arange(Q, dtype=np.uint32)
    assert (lane("KYBER_MSG_BIT", x) == ((x >= 833) & (x <= 2496))).all()           # poly.go:150-165
    v = np.arange(-32768, 32768, dtype=np.int64)
    n, b = lane("KYBER_NORMALIZE", (v & 0xffff).astype(np.uint32), two=True)
    assert (n == v % Q).all()                                                        # poly.go:35-39
    bs = b.astype(np.int32).astype(np.int64)
    assert ((bs - v) % Q == 0).all() and (bs >= 0).all() and (bs <= Q).all()         # field.go:45-64


def law_mulc_and_reduce32_at_their_bounds(lane):
    limit = (1 << 32) // Q
    rng = np.random.default_rng(1)
    a_edges = np.array([0, 1, Q - 1, Q, 7 * Q + 3, 128 * Q - 1, 128 * Q, limit - 1], dtype=np.uint64)
    for a in a_edges:  # every residue w against the edges of the operand range
        w = np.arange(Q, dtype=np.uint32)
        got = lane("KYBER_MULC", np.full(Q, a, np.uint32), w)
        assert (got == (a * w.astype(np.uint64)) % Q).all(), int(a)
    a = rng.integers(0, limit, 1 << 20, dtype=np.uint64)
    w = rng.integers(0, Q, 1 << 20, dtype=np.uint64)
    assert (lane("KYBER_MULC", a.astype(np.uint32), w.astype(np.uint32)) == (a * w) % Q).all()
    t = np.concatenate([rng.integers(0, 1 << 32, 1 << 20, dtype=np.uint64), np.array([0, 1, Q, (1 << 31) - 1, 1 << 31, (1 << 32) - 1], np.uint64)])
    r = lane("KYBER_REDUCE32", t.astype(np.uint32)).astype(np.uint64)
    assert (r < Q).all()
    r32 = (1 << 32) % Q
    assert ((r * r32 + t) % Q == 0).all()                                            # r == -t 2^-32 (mod q)


def law_cbd2_word_and_dot2(lane):
    rng = np.random.default_rng(2)
    w = np.concatenate([rng.integers(0, 1 << 32, 1 << 16, dtype=np.uint64), np.array([0, 0xffffffff, 0x55555555, 0xaaaaaaaa, 0x12345678], np.uint64)])
    got = lane("KYBER_CBD2_WORD", w.astype(np.uint32)).astype(np.uint64)
    for k in range(8):                                                               # sample.go:67-95 on nibble k
        t = (w >> (4 * k)) & 15
        want = ((t & 1) + ((t >> 1) & 1)).astype(np.int64) - (((t >> 2) & 1) + ((t >> 3) & 1)).astype(np.int64) + 8
        assert (((got >> (4 * k)) & 15).astype(np.int64) == want).all()
    a = rng.integers(0, 1 << 32, 1 << 16, dtype=np.uint64)
    b = rng.integers(0, 1 << 32, 1 << 16, dtype=np.uint64)
    s16 = lambda v: ((v & 0xffff) ^ 0x8000).astype(np.int64) - 0x8000                # noqa: E731
    want = (s16(a) * s16(b) + s16(a >> 16) * s16(b >> 16) + 12345) & 0xffffffff
    assert (lane("KYBER_DOT2", a.astype(np.uint32), b.astype(np.uint32), arg=12345) == want).all()


def law_decompose_law_for_every_a(lane, gamma2):
    # rounding_test.go:14-37 TestDecompose, for every a < q
    alpha = 2 * gamma2
    a = np.arange(DQ, dtype=np.uint32)
    a0q, a1 = lane("DIL_DECOMPOSE", a, arg=gamma2, two=True)
    a0 = a0q.astype(np.int64) - DQ
    a1 = a1.astype(np.int64)
    rec = a0 + alpha * a1
    wrap = (a1 == 0) & (rec < 0)
    assert ((-(alpha // 2) <= a0[wrap]) & (a0[wrap] < 0)).all()
    assert ((-(alpha // 2) < a0[~wrap]) & (a0[~wrap] <= alpha // 2)).all()
    rec[wrap] += DQ
    assert (rec == a.astype(np.int64)).all()
    assert (a1 < (16 if gamma2 == 261888 else 44)).all()


def law_make_hint_use_hint_law(lane, gamma2):
    # rounding_test.go:39-67 TestMakeHint: useHint(w - f, makeHint(w0 - f, w1)) == w1 for |f| <= gamma2.  The reference
    # sweeps all (w, f) behind a -very-long flag; here every w < q against the f that sit on the decision boundaries.
    w = np.arange(DQ, dtype=np.uint32)
    w0q, w1 = lane("DIL_DECOMPOSE", w, arg=gamma2, two=True)
    w64, w0 = w.astype(np.int64), w0q.astype(np.int64)
    for fn in (0, 1, 2, gamma2 // 2, gamma2 - 1, gamma2):
        for f in ({fn, (DQ - fn) % DQ}):
            z0 = ((w0 + DQ - f) % DQ).astype(np.uint32)
            hint = lane("DIL_MAKE_HINT", z0, w1, arg=gamma2)
            z0s = z0.astype(np.int64)
            want_hint = ~((z0s <= gamma2) | (z0s > DQ - gamma2) | ((z0s == DQ - gamma2) & (w1 == 0)))   # rounding.go:56-70
            assert (hint == want_hint).all()
            w1p = lane("DIL_USE_HINT", ((w64 + DQ - f) % DQ).astype(np.uint32), hint, arg=gamma2)
            assert (w1p == w1).all(), (gamma2, f)


def law_power2round_exceeds_normalize(lane):
    a = np.arange(DQ, dtype=np.uint32)
    a0q, a1 = lane("DIL_POWER2ROUND", a, two=True)                                   # field.go:35-52
    a0 = a0q.astype(np.int64) - DQ
    assert (a1.astype(np.int64) * 8192 + a0 == a.astype(np.int64)).all() and (a0 > -4096).all() and (a0 <= 4096).all()
    for bound in (1, 78, 196, 95232 - 78, 261888 - 196, (1 << 17) - 78, (1 << 19) - 196, (DQ - 1) // 2):   # poly.go:51-71
        centred = np.minimum(a.astype(np.int64), DQ - a.astype(np.int64))
        assert (lane("DIL_EXCEEDS", a, np.uint32(bound)) == (centred >= bound)).all(), bound
    rng = np.random.default_rng(4)
    t = np.concatenate([rng.integers(0, 1 << 32, 1 << 20, dtype=np.uint64), np.array([0, DQ - 1, DQ, 2 * DQ, (1 << 32) - 1], np.uint64)])
    n, f = lane("DIL_NORMALIZE", t.astype(np.uint32), two=True)
    assert (n == t % DQ).all()
    assert (f < (1 << 24)).all() and ((f.astype(np.int64) - t.astype(np.int64)) % DQ == 0).all()   # field.go:5-13 generalised


def law_montgomery_products_at_their_bounds(lane):
    rng = np.random.default_rng(5)
    r32 = (1 << 32) % DQ
    n = 1 << 20
    a = rng.integers(0, 1 << 32, n, dtype=np.uint64)
    b = rng.integers(0, DQ, n, dtype=np.uint64)                                       # a b < 2^32 q
    a[:6] = [(1 << 32) - 1, (1 << 32) - 1, 0, 1, DQ, 512 * DQ - 1]
    b[:6] = [DQ - 1, 1, DQ - 1, 1, DQ - 1, DQ - 1]
    for x, y in ((a, b), (b, a)):  # either operand may be the large one
        r = lane("DIL_MONT32", x.astype(np.uint32), y.astype(np.uint32)).astype(object)
        xo, yo = x.astype(object), y.astype(object)
        assert all(0 < v < 2 * DQ for v in r[:4096]) and int(max(r)) < 2 * DQ and int(min(r)) > 0
        assert all((int(rv) * r32 - int(xv) * int(yv)) % DQ == 0 for rv, xv, yv in zip(r[:20000], xo[:20000], yo[:20000]))   # field.go:20-24
    # mont64 on lazily accumulated sums: any t < 2^32 q, incl. the largest
    hi = rng.integers(0, DQ, n, dtype=np.uint64)
    lo = rng.integers(0, 1 << 32, n, dtype=np.uint64)
    hi[:3], lo[:3] = [DQ - 1, 0, DQ - 1], [(1 << 32) - 1, 0, 0]
    r = lane("DIL_MONT64", lo.astype(np.uint32), hi.astype(np.uint32))
    assert (r > 0).all() and (r < 2 * DQ).all()
    assert all((int(rv) * r32 - ((int(h) << 32) | int(l))) % DQ == 0 for rv, h, l in zip(r[:20000], hi[:20000], lo[:20000]))
