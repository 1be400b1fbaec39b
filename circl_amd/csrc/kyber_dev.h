// kyber_dev.h -- the Kyber ring Z_3329[x]/(x^256+1) on gfx950, one polynomial per wavefront.
//
// Replaces pke/kyber/internal/common {field,ntt,poly,sample}.go and its AVX2 assembler.
// A polynomial lives in the registers of ONE 64-lane wavefront, 4 coefficients per lane, as NON-NEGATIVE
// 32-bit integers.  The reference reduces with signed Montgomery arithmetic, R = 2^16 (field.go:4-32): five
// instructions per twiddle product here (multiply, low-half multiply, sign extension, multiply-subtract, shift).
// On gfx950 V_MUL_LO_U32 / V_MUL_HI_U32 issue at the rate of the 24-bit multipliers (profiles/r01_valu_issue_rates.txt),
// which makes Montgomery reduction with R = 2^32 a TWO-instruction product by a constant:
//     mulc(b, C(w)) = hi32( lo32(b * C(w)) * q )  ==  w b  (mod q),  in [0, q),      C(w) = (-w 2^32 mod q) q^-1 mod 2^32
// (b Z with Z = -w 2^32 mod q is below 2^32 for b < 2^32 / q, so its high word is zero and the quotient word m =
// lo32(b Z q^-1) satisfies m q = b Z + 2^32 h exactly; h = -b Z 2^-32 = w b mod q).  The same two instructions reduce any
// 32-bit accumulator: reduce32(t) = hi32(lo32(t q^-1) q) == -t 2^-32.  Butterflies are then 5 instructions (7 before),
// the inverse transform needs no Barrett steps at all (values stay far below the 2^32 / q bound), and its final
// scaling leaves canonical residues.  Every value is congruent mod q to the reference's, so packed outputs are
// bit-identical; the stray factors (-2^-32 of reduce32) are folded into the constant of the inverse transform's last step.
//
// Register layouts (lane l in 0..63, register r in 0..3 -> coefficient index n):
//   L1: n = l + 64 r                              bits 7,6 of n are register-local
//   L2: n = ((l>>4)<<6) | (r<<4) | (l&15)          bits 5,4 local
//   L3: n = ((l>>2)<<4) | (r<<2) | (l&3)           bits 3,2 local
//   L4: n = 4 l + r                                bits 1,0 local ("4 consecutive coefficients")
// The 7 NTT layers (strides 128..2) are done two at a time on register-local pairs; between
// them the wave re-distributes the polynomial through a 1 KB LDS scratch (3 exchanges per
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
CIRCL_HD uint32_t umulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// field.go:45-64 barrettReduce for |x| < 2^15: result in [0, q]
CIRCL_HD int barrett(int x) { return x - mul24(mul24(x, 20159) >> 26, Q); }
// field.go:67-74 csubq
CIRCL_HD int csubq(int x) {
    x -= Q;
    return x + ((x >> 31) & Q);
}
// poly.go:35-39 Normalize one coefficient (|x| < 2^15) to [0, q)
CIRCL_HD int normalize(int x) { return csubq(barrett(x)); }

// ---- products by constants and reductions with R = 2^32 (see the header comment) ----------------
constexpr uint32_t qinv32() {  // q^-1 mod 2^32 by Newton iteration
    uint32_t x = 1;
    for (int i = 0; i < 6; i++) x = x * (2u - (uint32_t)Q * x);
    return x;
}
constexpr uint32_t QINV32 = qinv32();
static_assert((uint32_t)(QINV32 * (uint32_t)Q) == 1u, "q * qinv == 1 mod 2^32");
constexpr uint32_t R32 = (uint32_t)((1ull << 32) % Q);  // 2^32 mod q
constexpr uint32_t MULC_LIMIT = (uint32_t)((1ull << 32) / Q);  // operands of mulc stay below this (1 290 167)
constexpr uint32_t modq(int64_t x) { return (uint32_t)(((x % Q) + Q) % Q); }
constexpr uint32_t powq(uint32_t b, uint32_t e) {
    uint32_t r = 1;
    for (uint32_t i = 0; i < e; i++) r = r * b % Q;
    return r;
}
constexpr uint32_t invq(uint32_t x) { return powq(x % Q, Q - 2); }
// the constant C(w) of mulc for the residue w
constexpr uint32_t mulc_const(uint32_t w) { return modq(-(int64_t)((uint64_t)(w % Q) * R32 % Q)) * QINV32; }
// w b mod q in [0, q) for b < MULC_LIMIT
CIRCL_HD uint32_t mulc(uint32_t b, uint32_t c) { return umulhi32(b * c, (uint32_t)Q); }
// -t 2^-32 mod q in [0, q) for any 32-bit t
CIRCL_HD uint32_t reduce32(uint32_t t) { return umulhi32(t * QINV32, (uint32_t)Q); }
constexpr uint32_t NEG_R32 = modq(-(int64_t)R32);  // multiplying by it undoes the factor of reduce32

// ntt.go:16-28 tabulates Zetas[i] = 17^brv7(i) 2^16; here the same powers as mulc constants, computed at compile time.
struct ZetaTable {
    uint32_t w[128];  // 17^brv7(i) mod q
    uint32_t c[128];  // mulc_const(w[i])
    uint32_t cn[128]; // mulc_const(-w[i])
};
constexpr ZetaTable make_zetas() {
    ZetaTable t{};
    for (int i = 0; i < 128; i++) {
        int brv = 0;
        for (int b = 0; b < 7; b++) brv |= ((i >> b) & 1) << (6 - b);
        t.w[i] = powq(17, (uint32_t)brv);
        t.c[i] = mulc_const(t.w[i]);
        t.cn[i] = mulc_const(modq(-(int64_t)t.w[i]));
    }
    return t;
}
CIRCL_HD uint32_t zeta_c(int i) {  // mulc constant of Zetas[i]; a function-local table is addressed pc-relative (no GOT)
    constexpr ZetaTable t = make_zetas();
    return t.c[i];
}
CIRCL_HD uint32_t zeta_cn(int i) {
    constexpr ZetaTable t = make_zetas();
    return t.cn[i];
}
CIRCL_HD uint32_t zeta_plain(int i) {
    constexpr ZetaTable t = make_zetas();
    return t.w[i];
}

// Per-lane twiddle constants, loaded once per kernel.  Forward layer t (stride 128>>t) uses
// k = 2^t + (n >> (8-t)) (ntt.go:117-134); the inverse walks the same table backwards
// (ntt.go:145-193), i.e. index 3*2^t - 1 - k.  f6n = -Zetas[64+l] is the second pair's factor in MulHat.
struct LaneZetas {
    uint32_t f2, f3a, f3b, f4, f5a, f5b, f6, f6n;
    uint32_t i2, i3a, i3b, i4, i5a, i5b, i6;
};
CIRCL_HD LaneZetas load_lane_zetas(int lane) {
    LaneZetas z;
    const int h = lane >> 4, m = lane >> 2;
    z.f2 = zeta_c(4 + h);
    z.f3a = zeta_c(8 + 2 * h);
    z.f3b = zeta_c(9 + 2 * h);
    z.f4 = zeta_c(16 + m);
    z.f5a = zeta_c(32 + 2 * m);
    z.f5b = zeta_c(33 + 2 * m);
    z.f6 = zeta_c(64 + lane);
    z.f6n = zeta_cn(64 + lane);
    z.i2 = zeta_c(7 - h);
    z.i3a = zeta_c(15 - 2 * h);
    z.i3b = zeta_c(14 - 2 * h);
    z.i4 = zeta_c(31 - m);
    z.i5a = zeta_c(63 - 2 * m);
    z.i5b = zeta_c(62 - 2 * m);
    z.i6 = zeta_c(127 - lane);
    return z;
}

CIRCL_HD int idx_l1(int l, int r) { return l + 64 * r; }
CIRCL_HD int idx_l2(int l, int r) { return ((l >> 4) << 6) | (r << 4) | (l & 15); }
CIRCL_HD int idx_l3(int l, int r) { return ((l >> 2) << 4) | (r << 2) | (l & 3); }
CIRCL_HD int idx_l4(int l, int r) { return 4 * l + r; }

// Cooley-Tukey butterfly (ntt.go:126-131) on non-negative values: both outputs grow by at most q.
CIRCL_HD void ct(int &a, int &b, uint32_t c) {
    const int t = (int)mulc((uint32_t)b, c);
    b = a + Q - t;
    a = a + t;
}
// Gentleman-Sande butterfly (ntt.go:168-176): a' = a + b, b' = zeta (b - a).  BOUND = a multiple of q that both
// inputs are below; the sum doubles, the product is back in [0, q).
template <int BOUND> CIRCL_HD void gs(int &a, int &b, uint32_t c) {
    const int t = b + BOUND - a;
    a = a + b;
    b = (int)mulc((uint32_t)t, c);
}

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// ---- wave-cooperative code (device only) ------------------------------------------------

// All kernels that use these run ONE wavefront per workgroup, so a workgroup barrier is a
// wave-local ordering point for LDS.
// NW ("no wait"): the buffer belongs to ONE wavefront, whose LDS instructions execute in order, so the reads behind the writes of
// an exchange need neither a barrier nor an s_waitcnt -- only the compiler must keep their order.  This is the form for
// workgroups of SEVERAL wavefronts that each work on their own buffers (mlkem_decaps_chain_kernel): no s_barrier is issued.
template <bool NW = false> __device__ __forceinline__ void wave_sync() {
    if constexpr (NW) wave_lds_order();
    else __syncthreads();
}

// Re-distribute a polynomial between register layouts through `xch` (256 elements of T in LDS: 512 bytes for
// uint16_t, 1 KB for uint32_t -- values must fit T).
template <int FROM, int TO, class T, bool NW = false> __device__ __forceinline__ void relayout(int (&c)[4], void *xch_raw, int lane) {
    T *xch = reinterpret_cast<T *>(xch_raw);
    auto idx = [&](int which, int r) {
        return which == 1 ? idx_l1(lane, r) : which == 2 ? idx_l2(lane, r) : which == 3 ? idx_l3(lane, r) : idx_l4(lane, r);
    };
    wave_sync<NW>();  // earlier readers of xch are done
#pragma unroll
    for (int r = 0; r < 4; r++) xch[idx(FROM, r)] = (T)c[r];
    wave_sync<NW>();
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = (int)xch[idx(TO, r)];
}

// Poly.NTT (ntt.go:60-135).  In: layout L1, 0 <= c < IN_BOUND (any bound up to 2^16 - 7q).  Out: layout L4,
// 0 <= c < IN_BOUND + 7q, congruent to the reference's transform.
template <bool NW = false> __device__ __forceinline__ void ntt(int (&c)[4], const LaneZetas &z, void *xch, int lane) {
    const uint32_t z1 = zeta_c(1), z2 = zeta_c(2), z3 = zeta_c(3);
    ct(c[0], c[2], z1); ct(c[1], c[3], z1);
    ct(c[0], c[1], z2); ct(c[2], c[3], z3);
    relayout<1, 2, uint16_t, NW>(c, xch, lane);
    ct(c[0], c[2], z.f2); ct(c[1], c[3], z.f2);
    ct(c[0], c[1], z.f3a); ct(c[2], c[3], z.f3b);
    relayout<2, 3, uint16_t, NW>(c, xch, lane);
    ct(c[0], c[2], z.f4); ct(c[1], c[3], z.f4);
    ct(c[0], c[1], z.f5a); ct(c[2], c[3], z.f5b);
    relayout<3, 4, uint16_t, NW>(c, xch, lane);
    ct(c[0], c[2], z.f6); ct(c[1], c[3], z.f6);
}

// Poly.InvNTT (ntt.go:145-193) times a caller-chosen residue: out = SCALE * InvNTT_true(in), where InvNTT_true
// includes the 1/128.  (The reference's transform is the case SCALE = 2^16: its last step multiplies by 1441 =
// 128^-1 R^2 after Montgomery products that each carried R^-1.)  In: layout L4, 0 <= c < q.  Out: layout L1, 0 <= c < q.
// The sums double per layer (below 2^k q after k layers) and are never reduced on the way: 128 q is far below
// MULC_LIMIT.  The first two exchanges fit 16-bit elements (< 8q), the third (< 32q) uses 32-bit ones.
template <uint32_t SCALE, bool NW = false> __device__ __forceinline__ void invntt(int (&c)[4], const LaneZetas &z, void *xch, int lane) {
    static_assert(128u * Q < MULC_LIMIT, "lazy sums stay inside the mulc domain");
    gs<Q>(c[0], c[2], z.i6); gs<Q>(c[1], c[3], z.i6);
    relayout<4, 3, uint16_t, NW>(c, xch, lane);
    gs<2 * Q>(c[0], c[1], z.i5a); gs<2 * Q>(c[2], c[3], z.i5b);
    gs<4 * Q>(c[0], c[2], z.i4); gs<4 * Q>(c[1], c[3], z.i4);
    relayout<3, 2, uint16_t, NW>(c, xch, lane);
    gs<8 * Q>(c[0], c[1], z.i3a); gs<8 * Q>(c[2], c[3], z.i3b);
    gs<16 * Q>(c[0], c[2], z.i2); gs<16 * Q>(c[1], c[3], z.i2);
    relayout<2, 1, uint32_t, NW>(c, xch, lane);
    const uint32_t z1 = zeta_c(1), z2 = zeta_c(2), z3 = zeta_c(3);
    gs<32 * Q>(c[0], c[1], z3); gs<32 * Q>(c[2], c[3], z2);
    gs<64 * Q>(c[0], c[2], z1); gs<64 * Q>(c[1], c[3], z1);
    constexpr uint32_t fin = mulc_const((uint32_t)((uint64_t)(SCALE % Q) * invq(128) % Q));
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = (int)mulc((uint32_t)c[r], fin);
}
#endif  // device

// MulHat accumulation (poly.go:63-100 + vec.go:30-37 PolyDotHat), layout L4, on packed int16 pairs with
// V_DOT2_I32_I16 (two multiplies and the add per instruction), kept lazy:
//   acc0 += a0 b0 + a1 (zeta b1)     acc1 += a0 b1 + a1 b0     acc2 += a2 b2 + a3 (-zeta b3)     acc3 += a2 b3 + a3 b2
// The b side is prepared once per polynomial and reused for every row of the matrix:
//   p0 = (b0, zeta b1 mod q)   q0 = (b1, b0)     p1 = (b2, -zeta b3 mod q)   q1 = (b3, b2)
// The a side is the pair of dwords exactly as they lie in memory (int16 coefficients 4l..4l+3, in [0, q)).
// Everything is non-negative: with b < 2^15 a term is below 2 q 2^15, so K <= 4 terms stay below 2^31; ONE
// reduce32 per coefficient at the end (mulhat_finish) leaves -2^-32 times the true products, in [0, q).
struct HatOperand {
    uint32_t p0, q0, p1, q1;
};
CIRCL_HD uint32_t pack16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
CIRCL_HD HatOperand hat_prepare(const int (&b)[4], uint32_t c_zeta, uint32_t c_neg_zeta) {
    HatOperand h;
    h.p0 = pack16(b[0], (int)mulc((uint32_t)b[1], c_zeta));
    h.q0 = pack16(b[1], b[0]);
    h.p1 = pack16(b[2], (int)mulc((uint32_t)b[3], c_neg_zeta));
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
    for (int r = 0; r < 4; r++) acc[r] = (int)reduce32((uint32_t)acc[r]);
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

// poly.go:248-332 CompressTo arithmetic: round(x 2^d / q) mod 2^d for ANY representative 0 <= x < 4q (2q + 8 for
// d = 11) of the coefficient -- the rounding is periodic in q modulo 2^d, so callers need not bring the sum of an
// inverse transform's output and the noise back into [0, q).  floor(y / q) for y = x 2^d + q/2 < 2^24 as
// (y * ceil(2^35 / q)) >> 35: both factors fit the 24-bit multiplier (V_MUL_HI_U32_U24 gives bits 32..47) and the
// rounding error y * 2492 / (q 2^35) stays below the 1/q slack of a floor.  Same value as the reference's
// multiply-shift constants on [0, q) (poly.go:254-260); checked for every x of the domain in tests/test_hostsim.py.
template <int D> CIRCL_HD unsigned compress_coeff(int x) {
    static_assert(D == 4 || D == 5 || D == 10 || D == 11, "d");
    const unsigned y = ((unsigned)x << D) + Q / 2;
    const unsigned hi = (unsigned)(((uint64_t)(y & 0xffffffu) * (uint64_t)10321340u) >> 32);
    return (hi >> 3) & ((1u << D) - 1);
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
