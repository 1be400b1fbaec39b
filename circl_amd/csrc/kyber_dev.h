// kyber_dev.h -- the Kyber ring Z_3329[x]/(x^256+1) on gfx950, one polynomial per wavefront.
//
// Replaces pke/kyber/internal/common {field,ntt,poly,sample}.go and its AVX2 assembler.
// A polynomial lives in the registers of ONE 64-lane wavefront, 4 coefficients per lane.
// Coefficients are 32-bit registers holding small signed values; products use the full-rate
// 24-bit multipliers (V_MUL_I32_I24 / V_MAD_I32_I24) and signed Montgomery reduction with
// R = 2^16 exactly as the reference (field.go:4-32), so every intermediate is congruent to the
// reference's and the packed outputs are bit-identical.
//
// Register layouts (lane l in 0..63, register r in 0..3 -> coefficient index n):
//   L1: n = l + 64 r                              bits 7,6 of n are register-local
//   L2: n = ((l>>4)<<6) | (r<<4) | (l&15)          bits 5,4 local
//   L3: n = ((l>>2)<<4) | (r<<2) | (l&3)           bits 3,2 local
//   L4: n = 4 l + r                                bits 1,0 local ("4 consecutive coefficients")
// The 7 NTT layers (strides 128..2) are done two at a time on register-local pairs; between
// them the wave re-distributes the polynomial through a 512-byte LDS scratch (3 exchanges per
// transform).  L4 is the layout of MulHat (pairs (4l,4l+1) and (4l+2,4l+3) share zeta =
// Zetas[64+l], poly.go:63-100), of the 12-bit codec and of coalesced 8-byte LDS/global access.
#pragma once
#include "keccak_dev.h"

namespace circl {
namespace kyber {

constexpr int Q = 3329;
constexpr int N = 256;

CIRCL_HD int mul24(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b);
#else
    return a * b;
#endif
}
CIRCL_HD int mad24(int a, int b, int c) { return mul24(a, b) + c; }
// (x mod 2^24) * C on the full-rate 24-bit multiplier, for callers that keep only low bits of the product.
// Written as C (or with __umul24, which is C underneath) LLVM's demanded-bits analysis drops the 24-bit
// masks -- they cannot change the low bits -- and then has to select the quarter-rate V_MUL_LO_U32.
template <uint32_t C> CIRCL_HD uint32_t umul24_lowbits(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "i"(C), "v"(x));
    return r;
#else
    return (x & 0xffffffu) * C;
#endif
}

// field.go:4-32 montReduce: x R^-1 mod q, q^-1 = 62209 (mod 2^16).  |x| < 2^31; result is
// x/2^16 + (-q/2, q/2).
CIRCL_HD int mont_reduce(int x) {
    const int m = (int)(int16_t)(uint16_t)umul24_lowbits<62209u>((uint32_t)x);
    return (x - mul24(m, Q)) >> 16;
}
CIRCL_HD int mont_mul(int a, int b) { return mont_reduce(mul24(a, b)); }
// field.go:45-64 barrettReduce for |x| < 2^15: result in [0, q]
CIRCL_HD int barrett(int x) { return x - mul24(mul24(x, 20159) >> 26, Q); }
// field.go:67-74 csubq
CIRCL_HD int csubq(int x) {
    x -= Q;
    return x + ((x >> 31) & Q);
}
// poly.go:35-39 Normalize one coefficient (|x| < 2^15) to [0, q)
CIRCL_HD int normalize(int x) { return csubq(barrett(x)); }

// ntt.go:16-28: Zetas[i] = 17^brv7(i) * 2^16 mod q, computed at compile time.
struct ZetaTable {
    int16_t v[128];
};
constexpr ZetaTable make_zetas() {
    ZetaTable t{};
    for (int i = 0; i < 128; i++) {
        int brv = 0;
        for (int b = 0; b < 7; b++) brv |= ((i >> b) & 1) << (6 - b);
        unsigned z = 1;
        for (int e = 0; e < brv; e++) z = z * 17 % Q;
        t.v[i] = (int16_t)((z << 16) % Q);
    }
    return t;
}
static __device__ __constant__ const ZetaTable kZetasDev = make_zetas();
static const ZetaTable kZetasHost = make_zetas();
CIRCL_HD int zeta(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return kZetasDev.v[i];
#else
    return kZetasHost.v[i];
#endif
}

// Per-lane twiddles, loaded once per kernel.  Forward layer t (stride 128>>t) uses
// k = 2^t + (n >> (8-t)) (ntt.go:117-134); the inverse walks the same table backwards
// (ntt.go:145-193), i.e. index 3*2^t - 1 - k.
struct LaneZetas {
    int f2, f3a, f3b, f4, f5a, f5b, f6;
    int i2, i3a, i3b, i4, i5a, i5b, i6;
};
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
    z.i2 = zeta(7 - h);
    z.i3a = zeta(15 - 2 * h);
    z.i3b = zeta(14 - 2 * h);
    z.i4 = zeta(31 - m);
    z.i5a = zeta(63 - 2 * m);
    z.i5b = zeta(62 - 2 * m);
    z.i6 = zeta(127 - lane);
    return z;
}

CIRCL_HD int idx_l1(int l, int r) { return l + 64 * r; }
CIRCL_HD int idx_l2(int l, int r) { return ((l >> 4) << 6) | (r << 4) | (l & 15); }
CIRCL_HD int idx_l3(int l, int r) { return ((l >> 2) << 4) | (r << 2) | (l & 3); }
CIRCL_HD int idx_l4(int l, int r) { return 4 * l + r; }

// Cooley-Tukey / Gentleman-Sande butterflies (ntt.go:126-131, :168-176)
CIRCL_HD void ct(int &a, int &b, int z) {
    const int t = mont_mul(z, b);
    b = a - t;
    a = a + t;
}
CIRCL_HD void gs(int &a, int &b, int z) {
    const int t = b - a;
    a = a + b;
    b = mont_mul(z, t);
}

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// ---- wave-cooperative code (device only) ------------------------------------------------

// All kernels that use these run ONE wavefront per workgroup, so a workgroup barrier is a
// wave-local ordering point for LDS.
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// Re-distribute a polynomial between register layouts through `xch` (int16[256] in LDS).
template <int FROM, int TO> __device__ __forceinline__ void relayout(int (&c)[4], int16_t *xch, int lane) {
    auto idx = [&](int which, int r) {
        return which == 1 ? idx_l1(lane, r) : which == 2 ? idx_l2(lane, r) : which == 3 ? idx_l3(lane, r) : idx_l4(lane, r);
    };
    wave_sync();  // earlier readers of xch are done
#pragma unroll
    for (int r = 0; r < 4; r++) xch[idx(FROM, r)] = (int16_t)c[r];
    wave_sync();
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = xch[idx(TO, r)];
}

// Poly.NTT (ntt.go:60-135).  In: layout L1, |c| <= q.  Out: layout L4, |c| <= 8q.
__device__ __forceinline__ void ntt(int (&c)[4], const LaneZetas &z, int16_t *xch, int lane) {
    const int z1 = zeta(1), z2 = zeta(2), z3 = zeta(3);
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
}

// Poly.InvNTT (ntt.go:145-193), including the final multiplication by 1441 = 128^-1 R^2.
// In: layout L4, |c| <= q.  Out: layout L1, |c| < q.  The reference Barrett-reduces a lazy
// subset of coefficients (InvNTTReductions); we reduce all four registers after every second
// layer, which is congruent mod q and keeps every value inside int16 for the LDS exchange.
__device__ __forceinline__ void invntt(int (&c)[4], const LaneZetas &z, int16_t *xch, int lane) {
    gs(c[0], c[2], z.i6); gs(c[1], c[3], z.i6);
    relayout<4, 3>(c, xch, lane);
    gs(c[0], c[1], z.i5a); gs(c[2], c[3], z.i5b);
    gs(c[0], c[2], z.i4); gs(c[1], c[3], z.i4);
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = barrett(c[r]);
    relayout<3, 2>(c, xch, lane);
    gs(c[0], c[1], z.i3a); gs(c[2], c[3], z.i3b);
    gs(c[0], c[2], z.i2); gs(c[1], c[3], z.i2);
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = barrett(c[r]);
    relayout<2, 1>(c, xch, lane);
    const int z1 = zeta(1), z2 = zeta(2), z3 = zeta(3);
    gs(c[0], c[1], z3); gs(c[2], c[3], z2);
    gs(c[0], c[2], z1); gs(c[1], c[3], z1);
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = mont_mul(1441, c[r]);
}
#endif  // device

// MulHat accumulation (poly.go:63-100 + vec.go:30-37 PolyDotHat), layout L4, kept lazy:
//   acc[0] += a0 b0 + zeta * mont(a1 b1)     acc[1] += a0 b1 + a1 b0
//   acc[2] += a2 b2 - zeta * mont(a3 b3)     acc[3] += a2 b3 + a3 b2
// with inputs in [0,q]: every term is < 2 q^2, so K <= 4 terms stay far below 2^31; one
// Montgomery reduction per coefficient at the end (mulhat_finish) gives the reference's value
// mod q (which carries the same single factor R^-1).
CIRCL_HD void mulhat_acc(int (&acc)[4], const int (&a)[4], const int (&b)[4], int zeta64) {
    const int t0 = mont_mul(a[1], b[1]);
    const int t1 = mont_mul(a[3], b[3]);
    acc[0] = mad24(a[0], b[0], mad24(t0, zeta64, acc[0]));
    acc[1] = mad24(a[0], b[1], mad24(a[1], b[0], acc[1]));
    acc[2] = mad24(a[2], b[2], mad24(t1, -zeta64, acc[2]));
    acc[3] = mad24(a[2], b[3], mad24(a[3], b[2], acc[3]));
}
// The same accumulation on packed int16 pairs with V_DOT2_I32_I16 (two multiplies and the add per
// instruction).  The b side is prepared once per polynomial and reused for every row of the matrix:
//   p0 = (b0, mont(zeta b1))   q0 = (b1, b0)     p1 = (b2, mont(-zeta b3))   q1 = (b3, b2)
// so that  acc0 += a0 b0 + a1 zeta b1 R^-1,  acc1 += a0 b1 + a1 b0  (and likewise for the second pair).
// The a side is the pair of dwords exactly as they lie in memory (int16 coefficients 4l..4l+3).
struct HatOperand {
    uint32_t p0, q0, p1, q1;
};
CIRCL_HD uint32_t pack16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
CIRCL_HD HatOperand hat_prepare(const int (&b)[4], int zeta64) {
    HatOperand h;
    h.p0 = pack16(b[0], mont_mul(b[1], zeta64));
    h.q0 = pack16(b[1], b[0]);
    h.p1 = pack16(b[2], mont_mul(b[3], -zeta64));
    h.q1 = pack16(b[3], b[2]);
    return h;
}
CIRCL_HD int dot2(uint32_t a, uint32_t b, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short short2v __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), c, false);
#else
    return (int)(int16_t)(a & 0xffff) * (int)(int16_t)(b & 0xffff) + (int)(int16_t)(a >> 16) * (int)(int16_t)(b >> 16) + c;
#endif
}
CIRCL_HD void mulhat_acc_packed(int (&acc)[4], uint32_t a01, uint32_t a23, const HatOperand &b) {
    acc[0] = dot2(a01, b.p0, acc[0]);
    acc[1] = dot2(a01, b.q0, acc[1]);
    acc[2] = dot2(a23, b.p1, acc[2]);
    acc[3] = dot2(a23, b.q1, acc[3]);
}
CIRCL_HD void mulhat_finish(int (&acc)[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] = mont_reduce(acc[r]);
}

// sample.go:31-95 centred binomial: coefficient n of eta=2 is nibble n of the PRF output,
// (b0+b1) - (b2+b3); of eta=3 the 6 bits at 6n, (b0+b1+b2) - (b3+b4+b5).
CIRCL_HD int cbd2_from_nibble(unsigned t) {
    return (int)((t & 1) + ((t >> 1) & 1)) - (int)(((t >> 2) & 1) + ((t >> 3) & 1));
}
// The eta = 2 rule on a whole PRF word: every nibble t becomes cbd2(t) + 8 (a value in 6..10; (b0+b1) + 8 - (b2+b3)
// cannot borrow from the next nibble).  The lane-per-stream PRF pass applies it to its output words -- eight
// coefficients per instruction, 64 streams per wave -- so that the ring phase, where a wave-instruction covers only
// 64 coefficients of ONE polynomial, is left with a bit-field extract and a subtraction per coefficient.
CIRCL_HD uint32_t cbd2_bias8_word(uint32_t w) {
    const uint32_t d = (w & 0x55555555u) + ((w >> 1) & 0x55555555u);
    return ((d & 0x33333333u) | 0x88888888u) - ((d >> 2) & 0x33333333u);
}
CIRCL_HD int cbd3_from_6bits(unsigned t) {
    return (int)((t & 1) + ((t >> 1) & 1) + ((t >> 2) & 1)) - (int)(((t >> 3) & 1) + ((t >> 4) & 1) + ((t >> 5) & 1));
}

// poly.go:248-332 CompressTo arithmetic for x in [0,q): round(x 2^d / q) mod 2^d with the
// reference's multiply-shift constants (proven exact on that domain, poly.go:254-260).
template <int D> CIRCL_HD unsigned compress_coeff(int x) {
    const unsigned y = ((unsigned)x << D) + Q / 2;
    if constexpr (D == 4 || D == 5) {
        return (umul24_lowbits<315u>(y) >> 20) & ((1u << D) - 1);  // y < 2^17: one full-rate 24-bit multiply
    } else {
        // floor(y / q) for y < 2^23 as (y * ceil(2^35 / q)) >> 35: both factors fit the full-rate 24-bit
        // multiplier (V_MUL_HI_U32_U24 gives bits 32..47); the rounding error y * 2492 / (q 2^35) < 2^-15 is
        // below the 1/q slack of a floor.  Same value as the reference's 64-bit multiply-shift; checked for
        // every x in tests/test_hostsim.py.
        static_assert(D == 10 || D == 11, "du");
        const unsigned hi = (unsigned)(((uint64_t)(y & 0xffffffu) * (uint64_t)10321340u) >> 32);
        return (hi >> 3) & ((1u << D) - 1);
    }
}
// poly.go:170-243 Decompress arithmetic
template <int D> CIRCL_HD int decompress_coeff(unsigned t) { return (int)(((1u << (D - 1)) + t * Q) >> D); }

// poly.go:150-165 CompressMessageTo for x in [0,q): bit = 1 iff 833 <= x <= 2496
CIRCL_HD unsigned msg_bit(int x) {
    int t = 1664 - x;
    t = (t >> 31) ^ t;
    t -= 832;
    return ((unsigned)t >> 31) & 1;
}

}  // namespace kyber
}  // namespace circl
