// dilithium_dev.h -- the Dilithium ring Z_8380417[x]/(x^256+1) on gfx950, one polynomial per
// wavefront, 4 coefficients per lane.
//
// Replaces sign/internal/dilithium {field,ntt,poly}.go and its AVX2 assembler.  The reference
// uses Montgomery arithmetic with R = 2^32 (field.go:20-24), which on CDNA4 would need
// quarter-rate 32x32 multiplies.  Every value here is < 2^24, so we use Montgomery arithmetic
// with R = 2^24 on the full-rate 24-bit multipliers instead (V_MUL_U32_U24 / V_MUL_HI_U32_U24):
//     mont24(a, b) = a b 2^-24 mod q,   a < 2^24, b < q   ->   result < 2q
// in 8 VALU instructions.  Twiddles are stored pre-multiplied by 2^24, so a butterfly computes the
// plain product zeta*b; coefficients are therefore plain residues mod q (no stray Montgomery
// factor), which is all that the packed outputs of ML-DSA depend on.
//
// Register layouts L1..L4 are those of kyber_dev.h; the 8 NTT layers (strides 128..1) are done
// two at a time on register-local pairs, with three re-distributions through 1 KB of LDS.
#pragma once
#include "kyber_dev.h"

namespace circl {
namespace dilithium {

constexpr uint32_t Q = 8380417;  // 2^23 - 2^13 + 1
constexpr int D = 13;

constexpr uint32_t cpow(uint64_t b, uint32_t e) {
    uint64_t r = 1;
    b %= Q;
    while (e) {
        if (e & 1) r = r * b % Q;
        b = b * b % Q;
        e >>= 1;
    }
    return (uint32_t)r;
}
constexpr uint32_t R24 = (1u << 24) % Q;                      // 2^24 mod q
constexpr uint32_t R24SQ = (uint32_t)((uint64_t)R24 * R24 % Q);  // 2^48 mod q
// q^-1 mod 2^24 (Newton iteration; checked by static_assert below), and its negation
constexpr uint32_t qinv24() {
    uint32_t x = 1;
    for (int i = 0; i < 6; i++) x = x * (2 - Q * x);
    return x & 0xffffff;
}
constexpr uint32_t NEG_QINV24 = (0x1000000 - qinv24()) & 0xffffff;
static_assert(((uint64_t)Q * qinv24() & 0xffffff) == 1, "q * qinv == 1 mod 2^24");

CIRCL_HD uint32_t umul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (uint32_t)((uint64_t)(a & 0xffffff) * (b & 0xffffff));
#endif
}
CIRCL_HD uint32_t umulhi24(uint32_t a, uint32_t b) {  // bits 32..47 of the 48-bit product
    // the AMDGPU backend selects V_MUL_HI_U32_U24 for this shape (operands provably 24-bit)
    return (uint32_t)(((uint64_t)(a & 0xffffff) * (uint64_t)(b & 0xffffff)) >> 32);
}

// a b 2^-24 mod q for a < 2^24, b < 2^24 with a*b < 2^24 * q; result < 2q.
CIRCL_HD uint32_t mont24(uint32_t a, uint32_t b) {
    const uint32_t lo = umul24(a, b), hi = umulhi24(a, b);
    const uint32_t m = kyber::umul24_lowbits<NEG_QINV24>(lo);  // only its low 24 bits are used below
    const uint32_t mlo = umul24(m, Q), mhi = umulhi24(m, Q);
    const uint32_t s = lo + mlo;               // low 24 bits are zero by construction
    const uint32_t top = hi + mhi + (s < lo ? 1u : 0u);
    return (s >> 24) | (top << 8);
}
// field.go:5-13 ReduceLe2Q generalised: x < 2^32 -> < 2^24 (and < 2q when x < 2^28)
CIRCL_HD uint32_t fold(uint32_t x) { return (x & 0x7fffff) + umul24(x >> 23, 8191u); }
// field.go:27-31 le2qModQ: x < 2q -> [0,q)
CIRCL_HD uint32_t csubq(uint32_t x) {
    x -= Q;
    return x + ((uint32_t)((int32_t)x >> 31) & Q);
}
CIRCL_HD uint32_t normalize(uint32_t x) { return csubq(fold(fold(x))); }  // any x < 2^32

// ntt.go:19-57: zeta^brv8(k), here times 2^24 (the reference stores them times 2^32)
struct ZetaTable {
    uint32_t v[256];
};
constexpr ZetaTable make_zetas() {
    ZetaTable t{};
    for (int i = 0; i < 256; i++) {
        int brv = 0;
        for (int b = 0; b < 8; b++) brv |= ((i >> b) & 1) << (7 - b);
        t.v[i] = (uint32_t)((uint64_t)cpow(1753, (uint32_t)brv) * R24 % Q);
    }
    return t;
}
static __device__ __constant__ const ZetaTable kZetasDev = make_zetas();
static const ZetaTable kZetasHost = make_zetas();
CIRCL_HD uint32_t zeta(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return kZetasDev.v[i];
#else
    return kZetasHost.v[i];
#endif
}

struct LaneZetas {
    uint32_t f2, f3a, f3b, f4, f5a, f5b, f6, f7a, f7b;
    uint32_t i2, i3a, i3b, i4, i5a, i5b, i6, i7a, i7b;
};
// forward layer t (stride 128>>t): k = 2^t + block; inverse: index 2^(t+1) - 1 - block, with the
// butterfly written as b' = zeta * (b - a)  (ntt.go:191-217: InvZetas[k] = -Zetas[255-k]).
CIRCL_HD LaneZetas load_lane_zetas(int lane) {
    LaneZetas z;
    const int h = lane >> 4, m = lane >> 2;
    z.f2 = zeta(4 + h);
    z.f3a = zeta(8 + 2 * h);
    z.f3b = zeta(9 + 2 * h);
    z.f4 = zeta(16 + m);
    z.f5a = zeta(32 + 2 * m);
    z.f5b = zeta(33 + 2 * m);
    z.f6 = zeta(64 + lane);
    z.f7a = zeta(128 + 2 * lane);
    z.f7b = zeta(129 + 2 * lane);
    z.i2 = zeta(7 - h);
    z.i3a = zeta(15 - 2 * h);
    z.i3b = zeta(14 - 2 * h);
    z.i4 = zeta(31 - m);
    z.i5a = zeta(63 - 2 * m);
    z.i5b = zeta(62 - 2 * m);
    z.i6 = zeta(127 - lane);
    z.i7a = zeta(255 - 2 * lane);
    z.i7b = zeta(254 - 2 * lane);
    return z;
}

// Cooley-Tukey: (a, b) -> (a + zb, a - zb); a may be lazy (< 2^31), b any < 2^32.
CIRCL_HD void ct(uint32_t &a, uint32_t &b, uint32_t z) {
    const uint32_t t = mont24(fold(b), z);  // < 2q
    b = a + (2 * Q - t);
    a = a + t;
}
// Gentleman-Sande: (a, b) -> (a + b, z (b - a)); a < 8q (two layers after a fold to < 2^24)
CIRCL_HD void gs(uint32_t &a, uint32_t &b, uint32_t z) {
    const uint32_t t = b + (8 * Q - a);
    a = a + b;
    b = mont24(fold(t), z);
}

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
template <int FROM, int TO> __device__ __forceinline__ void relayout(uint32_t (&c)[4], uint32_t *xch, int lane) {
    auto idx = [&](int which, int r) {
        return which == 1 ? kyber::idx_l1(lane, r) : which == 2 ? kyber::idx_l2(lane, r) : which == 3 ? kyber::idx_l3(lane, r) : kyber::idx_l4(lane, r);
    };
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) xch[idx(FROM, r)] = c[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = xch[idx(TO, r)];
}

// Poly.NTT (ntt.go:166-183).  In: layout L1, c < 2^24.  Out: layout L4, c < 17q, plain residues.
__device__ __forceinline__ void ntt(uint32_t (&c)[4], const LaneZetas &z, uint32_t *xch, int lane) {
    const uint32_t z1 = zeta(1), z2 = zeta(2), z3 = zeta(3);
    ct(c[0], c[2], z1); ct(c[1], c[3], z1);
    ct(c[0], c[1], z2); ct(c[2], c[3], z3);
    relayout<1, 2>(c, xch, lane);
    ct(c[0], c[2], z.f2); ct(c[1], c[3], z.f2);
    ct(c[0], c[1], z.f3a); ct(c[2], c[3], z.f3b);
    relayout<2, 3>(c, xch, lane);
    ct(c[0], c[2], z.f4); ct(c[1], c[3], z.f4);
    ct(c[0], c[1], z.f5a); ct(c[2], c[3], z.f5b);
    relayout<3, 4>(c, xch, lane);
    ct(c[0], c[2], z.f6); ct(c[1], c[3], z.f6);
    ct(c[0], c[1], z.f7a); ct(c[2], c[3], z.f7b);
}

// Exact inverse transform INCLUDING the factor 1/256 (the reference's InvNTT returns R/256 times
// this, ntt.go:212-216, compensated by its R^-1-carrying MulHat).  In: layout L4, c < 2^24.
// Out: layout L1, c < 2q.
__device__ __forceinline__ void invntt(uint32_t (&c)[4], const LaneZetas &z, uint32_t *xch, int lane) {
    gs(c[0], c[1], z.i7a); gs(c[2], c[3], z.i7b);
    gs(c[0], c[2], z.i6); gs(c[1], c[3], z.i6);
    relayout<4, 3>(c, xch, lane);
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = fold(c[r]);
    gs(c[0], c[1], z.i5a); gs(c[2], c[3], z.i5b);
    gs(c[0], c[2], z.i4); gs(c[1], c[3], z.i4);
    relayout<3, 2>(c, xch, lane);
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = fold(c[r]);
    gs(c[0], c[1], z.i3a); gs(c[2], c[3], z.i3b);
    gs(c[0], c[2], z.i2); gs(c[1], c[3], z.i2);
    relayout<2, 1>(c, xch, lane);
    const uint32_t z1 = zeta(1), z2 = zeta(2), z3 = zeta(3);
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = fold(c[r]);
    gs(c[0], c[1], z3); gs(c[2], c[3], z2);
    gs(c[0], c[2], z1); gs(c[1], c[3], z1);
    // times 256^-1:  mont24(x, 2^24 / 256) = x / 256
    constexpr uint32_t inv256R = (uint32_t)((uint64_t)cpow(256, Q - 2) * R24 % Q);
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = mont24(fold(c[r]), inv256R);
}
#endif

// rounding.go:13-43 decompose for a in [0,q): returns a1, and a0+q through the reference's formula
template <uint32_t GAMMA2> CIRCL_HD void decompose(uint32_t a, uint32_t &a0plusq, uint32_t &a1) {
    constexpr uint32_t ALPHA = 2 * GAMMA2;
    a1 = (a + 127) >> 7;
    if constexpr (ALPHA == 523776) {
        a1 = (a1 * 1025 + (1u << 21)) >> 22;
        a1 &= 15;
    } else {
        a1 = (a1 * 11275 + (1u << 23)) >> 24;
        a1 ^= (uint32_t)((int32_t)(43 - a1) >> 31) & a1;
    }
    a0plusq = a - a1 * ALPHA;
    a0plusq += (uint32_t)((int32_t)(a0plusq - (Q - 1) / 2) >> 31) & Q;
}
// rounding.go:98-135 PolyUseHint for one coefficient
template <uint32_t GAMMA2> CIRCL_HD uint32_t use_hint(uint32_t a, uint32_t hint) {
    uint32_t a0, a1;
    decompose<GAMMA2>(a, a0, a1);
    if (hint == 0) return a1;
    if constexpr (GAMMA2 == 261888) {
        return a0 > Q ? (a1 + 1) & 15 : (a1 - 1) & 15;
    } else {
        if (a0 > Q) return a1 == 43 ? 0 : a1 + 1;
        return a1 == 0 ? 43 : a1 - 1;
    }
}
// poly.go:51-71 exceeds for one coefficient x in [0,q)
CIRCL_HD bool exceeds(uint32_t x, uint32_t bound) {
    int32_t t = (int32_t)((Q - 1) / 2) - (int32_t)x;
    t ^= (t >> 31);
    t = (int32_t)((Q - 1) / 2) - t;
    return (uint32_t)t >= bound;
}

}  // namespace dilithium
}  // namespace circl
