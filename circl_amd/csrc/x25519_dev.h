// x25519_dev.h -- X25519 (RFC 7748) on gfx950, one scalar multiplication per lane.
//
// Replaces dh/x25519 (key.go, curve.go, curve_generic.go and the ADX/BMI2 assembler of math/fp25519) for the
// Diffie-Hellman half of the hybrid KEMs (kem/hybrid/xkem.go, kem/xwing/xwing.go: SURVEY.md 8(f) row f2).
//
// GF(2^255 - 19) in ten limbs of 26/25 bits (radix 2^25.5) held in 32-bit registers: gfx950 has a full-rate-for-a-
// multiplier V_MAD_U64_U32 (32 x 32 + 64 -> 64), so a field product is 100 multiply-accumulates into ten 64-bit
// column sums -- the wrap-around 2^255 = 19 and the doubling of odd x odd limb pairs are folded into pre-scaled copies of
// the operands -- followed by one carry chain.  The reference's representation (four saturated 64-bit words,
// fp_generic.go) would need 64 x 64 -> 128 products and carry flags, neither of which the vector ALU has.
// Everything is unsigned: a subtraction adds 2p limb-wise, and the bounds are
//     "carried" (output of mul / sqr / mul_small):  limbs <= 2^26 + 2^18 (even), 2^25 + 2^18 (odd)
//     add of two carried values:                    < 2^27.1
//     sub of two carried values:                    < 2^27.6      (a + 2p - b)
// A product of two values below 2^27.6 has column sums below 2^63.8 (tests/test_hostsim.py drives the host
// instantiation with every limb at its bound).  The ladder is RFC 7748's, which computes the same x/z as
// ladderStepGeneric (curve_generic.go:37-58); conditional swaps are per-lane selects on the scalar bit.
#pragma once
#include <stdint.h>

#ifndef CIRCL_HD
#if defined(__HIPCC__)
#define CIRCL_HD __host__ __device__ __forceinline__
#else
#define CIRCL_HD inline
#endif
#endif

namespace circl {
namespace x25519 {

struct Fe {
    uint32_t v[10];
};

constexpr uint32_t M26 = (1u << 26) - 1, M25 = (1u << 25) - 1;
CIRCL_HD constexpr uint32_t limb_mask(int i) { return (i & 1) ? M25 : M26; }
CIRCL_HD constexpr int limb_bits(int i) { return (i & 1) ? 25 : 26; }

CIRCL_HD Fe fe_const(uint32_t c) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = 0;
    r.v[0] = c;
    return r;
}

// 255-bit little-endian value in eight words (bit 255 ignored) -> limbs
CIRCL_HD Fe fe_from_words(const uint32_t w[8]) {
    Fe r;
    r.v[0] = w[0] & M26;
    r.v[1] = ((w[0] >> 26) | (w[1] << 6)) & M25;
    r.v[2] = ((w[1] >> 19) | (w[2] << 13)) & M26;
    r.v[3] = ((w[2] >> 13) | (w[3] << 19)) & M25;
    r.v[4] = (w[3] >> 6) & M26;
    r.v[5] = w[4] & M25;
    r.v[6] = ((w[4] >> 25) | (w[5] << 7)) & M26;
    r.v[7] = ((w[5] >> 19) | (w[6] << 13)) & M25;
    r.v[8] = ((w[6] >> 12) | (w[7] << 20)) & M26;
    r.v[9] = (w[7] >> 6) & M25;
    return r;
}

// one carry chain over 64-bit column sums: h0 -> h1 -> ... -> h9 -> (x19) h0 -> h1
CIRCL_HD Fe fe_carry64(uint64_t h[10]) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        h[i + 1] += h[i] >> limb_bits(i);
        r.v[i] = (uint32_t)h[i] & limb_mask(i);
    }
    uint64_t t = (uint64_t)r.v[0] + 19 * (h[9] >> 25);  // h9 >> 25 < 2^39
    r.v[9] = (uint32_t)h[9] & M25;
    r.v[0] = (uint32_t)t & M26;
    r.v[1] += (uint32_t)(t >> 26);  // < 2^18
    return r;
}

CIRCL_HD Fe fe_add(const Fe &a, const Fe &b) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}

// a - b + 2p, limb-wise; b must be carried (its limbs do not exceed those of 2p)
CIRCL_HD Fe fe_sub(const Fe &a, const Fe &b) {
    Fe r;
    r.v[0] = a.v[0] + (2 * M26 - 36) - b.v[0];  // 2 (2^26 - 19)
#pragma unroll
    for (int i = 1; i < 10; i++) r.v[i] = a.v[i] + 2 * limb_mask(i) - b.v[i];
    return r;
}

CIRCL_HD Fe fe_mul(const Fe &f, const Fe &g) {
    uint32_t g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        g19[i] = g.v[i] * 19u;
        f2[i] = f.v[i] << 1;
    }
    uint64_t h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) h[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const int k = i + j;
            const uint32_t a = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
            const uint32_t b = k >= 10 ? g19[j] : g.v[j];
            h[k % 10] += (uint64_t)a * b;
        }
    }
    return fe_carry64(h);
}

CIRCL_HD Fe fe_sqr(const Fe &f) {
    uint32_t f2[10], f4[10], f19[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        f2[i] = f.v[i] << 1;
        f4[i] = f.v[i] << 2;
        f19[i] = f.v[i] * 19u;
    }
    uint64_t h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) h[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
#pragma unroll
        for (int j = i; j < 10; j++) {
            const int k = i + j;
            const bool odd = (i & 1) && (j & 1);
            // multiplicity: 2 for i < j, times 2 for odd x odd
            const uint32_t a = (i < j) ? (odd ? f4[i] : f2[i]) : (odd ? f2[i] : f.v[i]);
            const uint32_t b = k >= 10 ? f19[j] : f.v[j];
            h[k % 10] += (uint64_t)a * b;
        }
    }
    return fe_carry64(h);
}

// f * c for a small constant (c < 2^20)
CIRCL_HD Fe fe_mul_small(const Fe &f, uint32_t c) {
    uint64_t h[10];
#pragma unroll
    for (int i = 0; i < 10; i++) h[i] = (uint64_t)f.v[i] * c;
    return fe_carry64(h);
}

// per-lane select on the bit (two V_CNDMASK per limb on the device; no branch)
CIRCL_HD void fe_cswap(Fe &a, Fe &b, uint32_t bit) {
    const bool c = bit != 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const uint32_t ta = a.v[i], tb = b.v[i];
        a.v[i] = c ? tb : ta;
        b.v[i] = c ? ta : tb;
    }
}

CIRCL_HD Fe fe_sqr_n(Fe t, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 0; i < n; i++) t = fe_sqr(t);
    return t;
}

// z^(p-2): 254 squarings + 11 products (the chain of fp.go:135-181)
CIRCL_HD Fe fe_inv(const Fe &z) {
    const Fe z2 = fe_sqr(z);
    const Fe z9 = fe_mul(fe_sqr_n(z2, 2), z);
    const Fe z11 = fe_mul(z9, z2);
    const Fe z2_5_0 = fe_mul(fe_sqr(z11), z9);
    const Fe z2_10_0 = fe_mul(fe_sqr_n(z2_5_0, 5), z2_5_0);
    const Fe z2_20_0 = fe_mul(fe_sqr_n(z2_10_0, 10), z2_10_0);
    const Fe z2_40_0 = fe_mul(fe_sqr_n(z2_20_0, 20), z2_20_0);
    const Fe z2_50_0 = fe_mul(fe_sqr_n(z2_40_0, 10), z2_10_0);
    const Fe z2_100_0 = fe_mul(fe_sqr_n(z2_50_0, 50), z2_50_0);
    const Fe z2_200_0 = fe_mul(fe_sqr_n(z2_100_0, 100), z2_100_0);
    const Fe z2_250_0 = fe_mul(fe_sqr_n(z2_200_0, 50), z2_50_0);
    return fe_mul(fe_sqr_n(z2_250_0, 5), z11);
}

// canonical 255-bit value of a carried element, as eight words (fp.go:30-38 ToBytes)
CIRCL_HD void fe_to_words(uint32_t w[8], const Fe &a) {
    uint32_t l[10];
#pragma unroll
    for (int i = 0; i < 10; i++) l[i] = a.v[i];
    // strict limbs except l0 < 2^26 + 19 * 2
#pragma unroll
    for (int i = 0; i < 9; i++) {
        l[i + 1] += l[i] >> limb_bits(i);
        l[i] &= limb_mask(i);
    }
    l[0] += 19 * (l[9] >> 25);
    l[9] &= M25;
    // q = 1 iff the value is >= p, i.e. iff value + 19 carries out of bit 255
    uint32_t q = (l[0] + 19) >> 26;
#pragma unroll
    for (int i = 1; i < 10; i++) q = (l[i] + q) >> limb_bits(i);
    l[0] += 19 * q;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        l[i + 1] += l[i] >> limb_bits(i);
        l[i] &= limb_mask(i);
    }
    l[9] &= M25;
    w[0] = l[0] | (l[1] << 26);
    w[1] = (l[1] >> 6) | (l[2] << 19);
    w[2] = (l[2] >> 13) | (l[3] << 13);
    w[3] = (l[3] >> 19) | (l[4] << 6);
    w[4] = l[5] | (l[6] << 25);
    w[5] = (l[6] >> 7) | (l[7] << 19);
    w[6] = (l[7] >> 13) | (l[8] << 12);
    w[7] = (l[8] >> 20) | (l[9] << 6);
}

// key.go:24-31 isValidPubKey on the masked public key words: reduce mod p, compare with the five u-coordinates of
// order 1, 2, 4, 8 (curve.go:71-96).  Returns 1 for a valid key.
CIRCL_HD uint32_t valid_public(const uint32_t u[8]) {
    uint32_t c[8];
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = u[i];
    bool ge_p = (c[7] == 0x7fffffffu) && (c[0] >= 0xffffffedu);
#pragma unroll
    for (int i = 1; i < 7; i++) ge_p = ge_p && (c[i] == 0xffffffffu);
    if (ge_p) {
        c[0] -= 0xffffffedu;
#pragma unroll
        for (int i = 1; i < 8; i++) c[i] = 0;
    }
    uint32_t hi_or = 0, hi_and = 0xffffffffu;
#pragma unroll
    for (int i = 1; i < 7; i++) {
        hi_or |= c[i];
        hi_and &= c[i];
    }
    const bool small = (hi_or | c[7]) == 0 && c[0] <= 1;                                           // 0, 1
    const bool minus1 = hi_and == 0xffffffffu && c[7] == 0x7fffffffu && c[0] == 0xffffffecu;      // p - 1
    constexpr uint32_t O8A[8] = {0x7c7aebe0u, 0xaeb8413bu, 0xfae35616u, 0x6ac49ff1u, 0xeb8d09dau, 0xfdb1329cu, 0x16056286u, 0x00b8495fu};
    constexpr uint32_t O8B[8] = {0xbc959c5fu, 0x248c50a3u, 0x55b1d0b1u, 0x5bef839cu, 0xc45c4404u, 0x868e1c58u, 0xdd4e22d8u, 0x57119fd0u};
    bool a = true, b = true;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a = a && c[i] == O8A[i];
        b = b && c[i] == O8B[i];
    }
    return (small || minus1 || a || b) ? 0u : 1u;
}

// X25519(k, u): k = the 32 scalar bytes as words (clamped here, key.go:16-21), u = the 32 point bytes as words (bit 255
// masked here, key.go:43).  BASE: u = 9 (KeyGen, key.go:34-36; the product by x1 becomes a product by a constant).
template <bool BASE>
CIRCL_HD void scalar_mult(uint32_t out[8], const uint32_t k_in[8], const uint32_t u_in[8]) {
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = k_in[i];
    k[0] &= ~7u;
    k[7] = (k[7] & 0x7fffffffu) | 0x40000000u;
    Fe x1 = fe_const(9);
    if (!BASE) {
        uint32_t u[8];
#pragma unroll
        for (int i = 0; i < 8; i++) u[i] = u_in[i];
        u[7] &= 0x7fffffffu;
        x1 = fe_from_words(u);
    }
    Fe x2 = fe_const(1), z2 = fe_const(0), x3 = x1, z3 = fe_const(1);
    uint32_t swap = 0;
    // bit 254 first: the scalar is shifted left by one so that the current bit is always the top bit of k[7]
#pragma unroll
    for (int i = 7; i > 0; i--) k[i] = (k[i] << 1) | (k[i - 1] >> 31);
    k[0] <<= 1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int t = 254; t >= 0; t--) {
        const uint32_t bit = k[7] >> 31;
#pragma unroll
        for (int i = 7; i > 0; i--) k[i] = (k[i] << 1) | (k[i - 1] >> 31);
        k[0] <<= 1;
        swap ^= bit;
        fe_cswap(x2, x3, swap);
        fe_cswap(z2, z3, swap);
        swap = bit;
        const Fe A = fe_add(x2, z2), B = fe_sub(x2, z2);
        const Fe C = fe_add(x3, z3), D = fe_sub(x3, z3);
        const Fe DA = fe_mul(D, A), CB = fe_mul(C, B);
        const Fe AA = fe_sqr(A), BB = fe_sqr(B);
        x3 = fe_sqr(fe_add(DA, CB));
        const Fe t1 = fe_sqr(fe_sub(DA, CB));
        z3 = BASE ? fe_mul_small(t1, 9) : fe_mul(x1, t1);
        x2 = fe_mul(AA, BB);
        const Fe E = fe_sub(AA, BB);
        z2 = fe_mul(E, fe_add(BB, fe_mul_small(E, 121666)));  // curve_generic.go:52-55: E (BB + 121666 E)
    }
    fe_cswap(x2, x3, swap);
    fe_cswap(z2, z3, swap);
    fe_to_words(out, fe_mul(x2, fe_inv(z2)));
}

}  // namespace x25519
}  // namespace circl

#include "x25519_base_table.h"

namespace circl {
namespace x25519 {

// KeyGen (key.go:34-36): X25519(k, 9) through the fixed-base comb of x25519_base_table.h instead of a ladder -- 64 mixed
// additions on the twisted Edwards form (7 products each) against 255 ladder steps (4 products + 4 squarings + 2 small):
// a third of the instructions.  The reference makes the same kind of trade with its own precomputation (a right-to-left
// Montgomery ladder over a table of multiples, curve.go:9-45, table.go); the result is the same u-coordinate.
// k B = sum_j e_j 16^j B with signed digits e_j in [-8, 8] (the clamped scalar has bit 255 clear, so the top digit needs no
// carry out).  The eight candidates of a digit position are the same for every lane: they are read with wave-uniform
// addresses and selected per lane by compares, so neither addresses nor control flow depend on the scalar.
// P + Q for Q = ((y+x)/2, (y-x)/2, dxy):  A = (Y+X) q0, B = (Y-X) q1, C = T q2;  E = A-B, H = A+B, G = Z+C, F = Z-C;
// (X, Y, Z, T) <- (E F, G H, F G, E H) -- the extended coordinates of the sum scaled by 1/4.  For -Q: q0 <-> q1, G <-> F.
CIRCL_HD void base_mult(uint32_t out[8], const uint32_t k_in[8]) {
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = k_in[i];
    k[0] &= ~7u;
    k[7] = (k[7] & 0x7fffffffu) | 0x40000000u;
    Fe X = fe_const(0), Y = fe_const(1), Z = fe_const(1), T = fe_const(0);
    uint32_t carry = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int j = 0; j < 64; j++) {
        const uint32_t d = (k[0] & 15u) + carry;  // 0..16
#pragma unroll
        for (int i = 0; i < 7; i++) k[i] = (k[i] >> 4) | (k[i + 1] << 28);
        k[7] >>= 4;
        carry = (j == 63) ? 0u : ((d + 8u) >> 4);
        const int32_t e = (int32_t)d - (int32_t)(carry << 4);  // -8..8
        const bool neg = e < 0;
        const uint32_t m = (uint32_t)(neg ? -e : e);
        Fe q0, q1, q2 = fe_const(0);  // the identity: ((1+0)/2, (1-0)/2, 0), 1/2 = 2^254 - 9
#pragma unroll
        for (int i = 0; i < 10; i++) q0.v[i] = q1.v[i] = (i == 0) ? (M26 - 8) : (i == 9) ? 0xffffffu : limb_mask(i);
#pragma unroll
        for (int mm = 0; mm < 8; mm++) {
            const uint32_t *t = base_comb_entry(j, mm);
            const uint32_t mask = 0u - (((m ^ (uint32_t)(mm + 1)) - 1u) >> 31);  // all ones iff m == mm + 1; no branch
#pragma unroll
            for (int i = 0; i < 10; i++) {
                q0.v[i] ^= mask & (q0.v[i] ^ t[i]);
                q1.v[i] ^= mask & (q1.v[i] ^ t[10 + i]);
                q2.v[i] ^= mask & (q2.v[i] ^ t[20 + i]);
            }
        }
        fe_cswap(q0, q1, neg ? 1u : 0u);
        const Fe A = fe_mul(fe_add(Y, X), q0), B = fe_mul(fe_sub(Y, X), q1), C = fe_mul(T, q2);
        const Fe E = fe_sub(A, B), H = fe_add(A, B);
        Fe G = fe_add(Z, C), F = fe_sub(Z, C);
        fe_cswap(G, F, neg ? 1u : 0u);
        X = fe_mul(E, F);
        Y = fe_mul(G, H);
        Z = fe_mul(F, G);
        T = fe_mul(E, H);
    }
    fe_to_words(out, fe_mul(fe_add(Z, Y), fe_inv(fe_sub(Z, Y))));  // u = (1 + y) / (1 - y)
}

}  // namespace x25519
}  // namespace circl
