/* oracle/x25519.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's X25519 (dh/x25519), the Diffie-Hellman half of the hybrid KEMs of SURVEY.md
 * 8(f) row f2 (kem/hybrid/xkem.go, kem/xwing/xwing.go).  Function by function:
 *   clamp                 dh/x25519/key.go:16-21
 *   isValidPubKey         dh/x25519/key.go:24-31   (reduce mod p, compare with the five low-order u-coordinates,
 *                                                   curve.go:71-96)
 *   KeyGen                dh/x25519/key.go:34-36   (the reference runs a table-driven Joye ladder, curve.go:9-45; its
 *                                                   result is Shared(secret, 9) -- key_test.go:100-112 TestBase --
 *                                                   which is what is computed here)
 *   Shared                dh/x25519/key.go:41-47   (top bit of the public key masked, ladder, validity flag)
 *   ladderMontgomery      dh/x25519/curve.go:47-63
 *   ladderStepGeneric     dh/x25519/curve_generic.go:37-58 (same sequence of field operations)
 *   toAffine              dh/x25519/curve.go:65-69
 * The field GF(2^255-19) (math/fp25519/fp_generic.go: four saturated 64-bit words, 2^256 = 38) is held here in
 * five 51-bit limbs with 128-bit products; every value is the same residue, and ToBytes (fp.go:30-38) emits the
 * canonical one.  Pinned by the reference's own vectors: dh/x25519/testdata/{rfc7748_kat_test, rfc7748_times_test,
 * wycheproof_kat}.json.gz through tests/test_oracle_x25519.py. */
#include <string.h>

#include "oracle.h"

typedef unsigned __int128 u128;
typedef struct { uint64_t v[5]; } fe;
#define MASK51 ((1ull << 51) - 1)

static void fe_frombytes(fe *r, const uint8_t s[32]) {
    uint64_t w[4];
    for (int i = 0; i < 4; i++) {
        w[i] = 0;
        for (int j = 0; j < 8; j++) w[i] |= (uint64_t)s[8 * i + j] << (8 * j);
    }
    r->v[0] = w[0] & MASK51;
    r->v[1] = ((w[0] >> 51) | (w[1] << 13)) & MASK51;
    r->v[2] = ((w[1] >> 38) | (w[2] << 26)) & MASK51;
    r->v[3] = ((w[2] >> 25) | (w[3] << 39)) & MASK51;
    r->v[4] = (w[3] >> 12) & MASK51; /* bit 255 is dropped: callers mask it first (key.go:43) */
}

static void fe_carry(fe *r) {
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < 4; i++) {
            r->v[i + 1] += r->v[i] >> 51;
            r->v[i] &= MASK51;
        }
        r->v[0] += 19 * (r->v[4] >> 51);
        r->v[4] &= MASK51;
    }
}

/* canonical little-endian bytes (fp.go:30-38 ToBytes = Modp + copy) */
static void fe_tobytes(uint8_t s[32], const fe *a) {
    fe t = *a;
    fe_carry(&t);
    /* q = 1 iff t >= p */
    uint64_t q = (t.v[0] + 19) >> 51;
    for (int i = 1; i < 5; i++) q = (t.v[i] + q) >> 51;
    t.v[0] += 19 * q;
    for (int i = 0; i < 4; i++) {
        t.v[i + 1] += t.v[i] >> 51;
        t.v[i] &= MASK51;
    }
    t.v[4] &= MASK51;
    uint64_t w[4];
    w[0] = t.v[0] | (t.v[1] << 51);
    w[1] = (t.v[1] >> 13) | (t.v[2] << 38);
    w[2] = (t.v[2] >> 26) | (t.v[3] << 25);
    w[3] = (t.v[3] >> 39) | (t.v[4] << 12);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) s[8 * i + j] = (uint8_t)(w[i] >> (8 * j));
}

static void fe_add(fe *r, const fe *a, const fe *b) {
    for (int i = 0; i < 5; i++) r->v[i] = a->v[i] + b->v[i];
}
/* a - b with a bias of 4p so that limbs stay non-negative for operands below 2^53 */
static void fe_sub(fe *r, const fe *a, const fe *b) {
    r->v[0] = a->v[0] + 4 * (MASK51 - 18) - b->v[0];
    for (int i = 1; i < 5; i++) r->v[i] = a->v[i] + 4 * MASK51 - b->v[i];
    fe_carry(r);
}
static void fe_mul(fe *r, const fe *a, const fe *b) {
    u128 t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) {
            u128 p = (u128)a->v[i] * b->v[j];
            if (i + j >= 5) p *= 19;
            t[(i + j) % 5] += p;
        }
    u128 c = 0;
    for (int i = 0; i < 5; i++) {
        t[i] += c;
        r->v[i] = (uint64_t)t[i] & MASK51;
        c = t[i] >> 51;
    }
    c *= 19;
    c += r->v[0];
    r->v[0] = (uint64_t)c & MASK51;
    r->v[1] += (uint64_t)(c >> 51);
}
static void fe_sqr(fe *r, const fe *a) { fe_mul(r, a, a); }
/* fp.AddSub (fp_generic.go:123-128): (x, y) <- (x + y, x - y) */
static void fe_addsub(fe *x, fe *y) {
    fe s, d;
    fe_add(&s, x, y);
    fe_carry(&s);
    fe_sub(&d, x, y);
    *x = s;
    *y = d;
}
static void fe_cmov(fe *x, const fe *y, unsigned b) {
    const uint64_t m = 0 - (uint64_t)(b & 1);
    for (int i = 0; i < 5; i++) x->v[i] ^= m & (x->v[i] ^ y->v[i]);
}
static void fe_mul_a24(fe *r, const fe *a) { /* curve_generic.go:60-85 mulA24Generic, A24 = 121666 */
    fe k = {{121666, 0, 0, 0, 0}};
    fe_mul(r, a, &k);
}
/* z^(p-2) (fp.go:135-181 Inv: the usual 254 squarings + 11 products) */
static void fe_inv(fe *r, const fe *z) {
    fe z2, z9, z11, z2_5_0, z2_10_0, z2_20_0, z2_50_0, z2_100_0, t;
    fe_sqr(&z2, z);
    fe_sqr(&t, &z2);
    fe_sqr(&t, &t);
    fe_mul(&z9, &t, z);
    fe_mul(&z11, &z9, &z2);
    fe_sqr(&t, &z11);
    fe_mul(&z2_5_0, &t, &z9);
    t = z2_5_0;
    for (int i = 0; i < 5; i++) fe_sqr(&t, &t);
    fe_mul(&z2_10_0, &t, &z2_5_0);
    t = z2_10_0;
    for (int i = 0; i < 10; i++) fe_sqr(&t, &t);
    fe_mul(&z2_20_0, &t, &z2_10_0);
    t = z2_20_0;
    for (int i = 0; i < 20; i++) fe_sqr(&t, &t);
    fe_mul(&t, &t, &z2_20_0);
    for (int i = 0; i < 10; i++) fe_sqr(&t, &t);
    fe_mul(&z2_50_0, &t, &z2_10_0);
    t = z2_50_0;
    for (int i = 0; i < 50; i++) fe_sqr(&t, &t);
    fe_mul(&z2_100_0, &t, &z2_50_0);
    t = z2_100_0;
    for (int i = 0; i < 100; i++) fe_sqr(&t, &t);
    fe_mul(&t, &t, &z2_100_0);
    for (int i = 0; i < 50; i++) fe_sqr(&t, &t);
    fe_mul(&t, &t, &z2_50_0);
    for (int i = 0; i < 5; i++) fe_sqr(&t, &t);
    fe_mul(r, &t, &z11);
}

/* curve_generic.go:37-58; w = [x1, x2, z2, x3, z3] */
static void ladder_step(fe w[5], unsigned b) {
    fe *x1 = &w[0], *x2 = &w[1], *z2 = &w[2], *x3 = &w[3], *z3 = &w[4];
    fe t0, t1;
    fe_addsub(x2, z2);
    fe_addsub(x3, z3);
    fe_mul(&t0, x2, z3);
    fe_mul(&t1, x3, z2);
    fe_addsub(&t0, &t1);
    fe_cmov(x2, x3, b);
    fe_cmov(z2, z3, b);
    fe_sqr(x3, &t0);
    fe_sqr(z3, &t1);
    fe_mul(z3, x1, z3);
    fe_sqr(x2, x2);
    fe_sqr(z2, z2);
    fe_sub(&t0, x2, z2);
    fe_mul_a24(&t1, &t0);
    fe_add(&t1, &t1, z2);
    fe_mul(x2, x2, z2);
    fe_mul(z2, &t0, &t1);
}

/* curve.go:47-63 + toAffine :65-69; k already clamped, u already masked */
static void ladder_montgomery(uint8_t out[32], const uint8_t k[32], const uint8_t u[32]) {
    fe w[5];
    memset(w, 0, sizeof w);
    fe_frombytes(&w[0], u);
    w[1].v[0] = 1;
    w[3] = w[0];
    w[4].v[0] = 1;
    unsigned move = 0;
    for (int s = 254; s >= 0; s--) {
        const unsigned bit = (k[s / 8] >> (s % 8)) & 1;
        ladder_step(w, move ^ bit);
        move = bit;
    }
    fe zi, x;
    fe_inv(&zi, &w[2]);
    fe_mul(&x, &w[1], &zi);
    fe_tobytes(out, &x);
}

static const uint8_t kLowOrder[5][32] = {
    /* curve.go:71-96: u = 0 (order 2), 1 (order 4), the two order-8 u-coordinates, p - 1 (order 4 on the twist) */
    {0},
    {1},
    {0xe0, 0xeb, 0x7a, 0x7c, 0x3b, 0x41, 0xb8, 0xae, 0x16, 0x56, 0xe3, 0xfa, 0xf1, 0x9f, 0xc4, 0x6a,
     0xda, 0x09, 0x8d, 0xeb, 0x9c, 0x32, 0xb1, 0xfd, 0x86, 0x62, 0x05, 0x16, 0x5f, 0x49, 0xb8, 0x00},
    {0x5f, 0x9c, 0x95, 0xbc, 0xa3, 0x50, 0x8c, 0x24, 0xb1, 0xd0, 0xb1, 0x55, 0x9c, 0x83, 0xef, 0x5b,
     0x04, 0x44, 0x5c, 0xc4, 0x58, 0x1c, 0x8e, 0x86, 0xd8, 0x22, 0x4e, 0xdd, 0xd0, 0x9f, 0x11, 0x57},
    {0xec, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff,
     0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0x7f},
};

/* key.go:41-47.  Returns 1 when the public key is valid (not of low order), 0 otherwise; shared is written either way. */
int orc_x25519_shared(uint8_t shared[32], const uint8_t secret[32], const uint8_t public_[32]) {
    uint8_t k[32], u[32], canon[32];
    memcpy(k, secret, 32);
    k[0] &= 248;
    k[31] = (uint8_t)((k[31] & 127) | 64);
    memcpy(u, public_, 32);
    u[31] &= 127;
    fe t;
    fe_frombytes(&t, u);
    fe_tobytes(canon, &t); /* Modp */
    int low = 0;
    for (int i = 0; i < 5; i++) low |= memcmp(canon, kLowOrder[i], 32) == 0;
    ladder_montgomery(shared, k, u);
    return !low;
}

/* key.go:34-36 */
void orc_x25519_keygen(uint8_t public_[32], const uint8_t secret[32]) {
    static const uint8_t base[32] = {9};
    (void)orc_x25519_shared(public_, secret, base);
}

/* batch forms: point == NULL means the base point (KeyGen) */
int orc_x25519_batch(const uint8_t *scalar, const uint8_t *point, uint8_t *out, uint8_t *ok, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (point) {
            const int v = orc_x25519_shared(out + 32 * i, scalar + 32 * i, point + 32 * i);
            if (ok) ok[i] = (uint8_t)v;
        } else {
            orc_x25519_keygen(out + 32 * i, scalar + 32 * i);
            if (ok) ok[i] = 1;
        }
    }
    return 0;
}
