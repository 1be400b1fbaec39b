/* oracle/dilithium.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Scalar C restatement of the reference's *generic Go* ML-DSA path:
 *   sign/internal/dilithium/{field,ntt,poly,pack}.go, params/params.go
 *   sign/mldsa/mldsa{44,65,87}/internal/{dilithium,sample,mat,vec,rounding,pack,params}.go
 *   sign/mldsa/mldsa65/dilithium.go:56-132 (context framing)
 * Parity is PINNED by tests/test_oracle_mldsa.py: NIST ACVP keyGen / sigGen / sigVer
 * vectors, the Wycheproof verify sets and the fixed sampler vector of
 * sign/mldsa/mldsa65/internal/sample_test.go:12-63.
 *
 * Nothing under circl_amd/ may include, link or call this file.
 */
#include "oracle.h"
#include "keccak.h"
#include <string.h>

#define DQ 8380417u
#define DN 256
#define DD 13
#define DQINV 4236238847u /* -(q^-1) mod 2^32, params.go:13 */
#define ROVER256 41978u   /* 256^-1 R^2 mod q, params.go:14 */

typedef struct { uint32_t c[DN]; } dpoly;

typedef struct {
    int k, l, eta, deta_bits, omega, tau, gamma1_bits;
    uint32_t gamma2;
    int ctilde;
    int tr;   /* TRSize: 64 for ML-DSA, 32 for round-3 Dilithium */
    int nist; /* params.go NIST flag: ML-DSA domain separation in keygen, rnd in signing */
} dparams;

static int dil_params(int param, dparams *p) {
    /* sign/mldsa/mldsa{44,65,87}/internal/params.go:5-18 */
    switch (param) {
    case 44: *p = (dparams){4, 4, 2, 3, 80, 39, 17, 95232, 32, 64, 1}; return 0;
    case 65: *p = (dparams){6, 5, 4, 4, 55, 49, 19, 261888, 48, 64, 1}; return 0;
    case 87: *p = (dparams){8, 7, 2, 3, 75, 60, 19, 261888, 64, 64, 1}; return 0;
    /* round-3 Dilithium2/3/5: sign/dilithium/mode{2,3,5}/internal/params.go:5-18 */
    case 2: *p = (dparams){4, 4, 2, 3, 80, 39, 17, 95232, 32, 32, 0}; return 0;
    case 3: *p = (dparams){6, 5, 4, 4, 55, 49, 19, 261888, 32, 32, 0}; return 0;
    case 5: *p = (dparams){8, 7, 2, 3, 75, 60, 19, 261888, 32, 32, 0}; return 0;
    }
    return -1;
}
#define P_BETA(P) ((uint32_t)((P)->tau * (P)->eta))
#define P_GAMMA1(P) (1u << (P)->gamma1_bits)
#define P_LEQETA_SZ(P) (DN * (P)->deta_bits / 8)
#define P_LEGAMMA1_SZ(P) (((P)->gamma1_bits + 1) * DN / 8)
#define P_W1_SZ(P) (DN * (23 - (P)->gamma1_bits) / 8)
#define P_PK_SZ(P) (32 + 320 * (P)->k)
#define P_SK_SZ(P) (32 + 32 + (P)->tr + P_LEQETA_SZ(P) * ((P)->l + (P)->k) + 416 * (P)->k)
#define P_SIG_SZ(P) ((P)->l * P_LEGAMMA1_SZ(P) + (P)->omega + (P)->k + (P)->ctilde)

/* ---- field.go ---------------------------------------------------------- */
static inline uint32_t reduce_le2q(uint32_t x) { /* field.go:5-13 */
    uint32_t x1 = x >> 23, x2 = x & 0x7FFFFF;
    return x2 + (x1 << 13) - x1;
}
static inline uint32_t le2q_modq(uint32_t x) { /* field.go:27-31 */
    x -= DQ;
    uint32_t mask = (uint32_t)((int32_t)x >> 31);
    return x + (mask & DQ);
}
static inline uint32_t mod_q(uint32_t x) { return le2q_modq(reduce_le2q(x)); } /* field.go:15-17 */
static inline uint32_t mont_reduce_le2q(uint64_t x) { /* field.go:20-24 */
    uint64_t m = (x * DQINV) & 0xffffffffu;
    return (uint32_t)((x + m * (uint64_t)DQ) >> 32);
}
static inline void power2round(uint32_t a, uint32_t *a0plusq, uint32_t *a1) { /* field.go:35-52 */
    uint32_t a0 = a & ((1u << DD) - 1);
    a0 -= (1u << (DD - 1)) + 1;
    a0 += (uint32_t)((int32_t)a0 >> 31) & (1u << DD);
    a0 -= (1u << (DD - 1)) - 1;
    *a0plusq = DQ + a0;
    *a1 = (a - a0) >> DD;
}

/* ---- ntt.go ------------------------------------------------------------ */
/* ntt.go:19-57 Zetas[i] = 1753^brv8(i) * 2^32 mod q; ntt.go:66-104 InvZetas[i] =
 * (1753^-1)^(256 - brv8(255-i)) * 2^32 = -Zetas[255-i] mod q.  Generated here. */
static uint32_t DZETAS[DN], DINVZETAS[DN];
static int dz_ready;
__attribute__((constructor)) static void dz_init(void) {
    uint64_t R = (1ull << 32) % DQ;
    for (int i = 0; i < DN; i++) {
        int brv = 0;
        for (int b = 0; b < 8; b++) brv |= ((i >> b) & 1) << (7 - b);
        uint64_t z = 1;
        for (int e = 0; e < brv; e++) z = z * 1753 % DQ;
        DZETAS[i] = (uint32_t)(z * R % DQ);
    }
    for (int i = 0; i < DN; i++) DINVZETAS[i] = DQ - DZETAS[255 - i];
    dz_ready = 1;
}

static void dpoly_ntt(dpoly *p) { /* ntt.go:166-183 nttGeneric */
    int k = 0;
    for (unsigned l = DN / 2; l > 0; l >>= 1)
        for (unsigned off = 0; off < DN - l; off += 2 * l) {
            uint64_t zeta = DZETAS[++k];
            for (unsigned j = off; j < off + l; j++) {
                uint32_t t = mont_reduce_le2q(zeta * p->c[j + l]);
                p->c[j + l] = p->c[j] + (2 * DQ - t);
                p->c[j] += t;
            }
        }
}
static void dpoly_invntt(dpoly *p) { /* ntt.go:191-217 invNttGeneric */
    int k = 0;
    for (unsigned l = 1; l < DN; l <<= 1)
        for (unsigned off = 0; off < DN - l; off += 2 * l) {
            uint64_t zeta = DINVZETAS[k++];
            for (unsigned j = off; j < off + l; j++) {
                uint32_t t = p->c[j];
                p->c[j] = t + p->c[j + l];
                t += 256 * DQ - p->c[j + l];
                p->c[j + l] = mont_reduce_le2q(zeta * t);
            }
        }
    for (unsigned j = 0; j < DN; j++) p->c[j] = mont_reduce_le2q((uint64_t)ROVER256 * p->c[j]);
}

/* ---- poly.go ----------------------------------------------------------- */
static void dpoly_reduce_le2q(dpoly *p) { for (int i = 0; i < DN; i++) p->c[i] = reduce_le2q(p->c[i]); }
static void dpoly_normalize(dpoly *p) { for (int i = 0; i < DN; i++) p->c[i] = mod_q(p->c[i]); }
static void dpoly_normalize_le2q(dpoly *p) { for (int i = 0; i < DN; i++) p->c[i] = le2q_modq(p->c[i]); }
static void dpoly_add(dpoly *r, const dpoly *a, const dpoly *b) { for (int i = 0; i < DN; i++) r->c[i] = a->c[i] + b->c[i]; }
static void dpoly_sub(dpoly *r, const dpoly *a, const dpoly *b) { /* poly.go:40-45 */
    for (int i = 0; i < DN; i++) r->c[i] = a->c[i] + (2 * DQ - b->c[i]);
}
static int dpoly_exceeds(const dpoly *p, uint32_t bound) { /* poly.go:51-71 */
    for (int i = 0; i < DN; i++) {
        int32_t x = (int32_t)((DQ - 1) / 2) - (int32_t)p->c[i];
        x ^= (x >> 31);
        x = (int32_t)((DQ - 1) / 2) - x;
        if ((uint32_t)x >= bound) return 1;
    }
    return 0;
}
static void dpoly_mulhat(dpoly *r, const dpoly *a, const dpoly *b) { /* poly.go:88-92 */
    for (int i = 0; i < DN; i++) r->c[i] = mont_reduce_le2q((uint64_t)a->c[i] * b->c[i]);
}

/* ---- bit codecs (sign/internal/dilithium/pack.go, mldsa65/internal/pack.go) ---- */
static void bits_put(uint8_t *buf, size_t bitpos, unsigned d, uint32_t v) {
    for (unsigned b = 0; b < d; b++, bitpos++)
        if ((v >> b) & 1) buf[bitpos >> 3] |= (uint8_t)(1u << (bitpos & 7));
}
static uint32_t bits_get(const uint8_t *buf, size_t bitpos, unsigned d) {
    uint32_t v = 0;
    for (unsigned b = 0; b < d; b++, bitpos++) v |= (uint32_t)((buf[bitpos >> 3] >> (bitpos & 7)) & 1) << b;
    return v;
}
/* pack.go:7-16 UnpackT1 / pack.go:88-98 PackT1: 10 bits */
static void unpack_t1(dpoly *p, const uint8_t *buf) { for (int i = 0; i < DN; i++) p->c[i] = bits_get(buf, 10u * i, 10); }
static void pack_t1(uint8_t *buf, const dpoly *p) { memset(buf, 0, 320); for (int i = 0; i < DN; i++) bits_put(buf, 10u * i, 10, p->c[i]); }
/* pack.go:23-82 PackT0/UnpackT0: 13 bits of q + 2^12 - x */
static void pack_t0(uint8_t *buf, const dpoly *p) {
    memset(buf, 0, 416);
    for (int i = 0; i < DN; i++) bits_put(buf, 13u * i, 13, DQ + (1u << (DD - 1)) - p->c[i]);
}
static void unpack_t0(dpoly *p, const uint8_t *buf) {
    for (int i = 0; i < DN; i++) p->c[i] = DQ + (1u << (DD - 1)) - bits_get(buf, 13u * i, 13);
}
/* mldsa65/internal/pack.go:9-64: 3 or 4 bits of q + eta - x */
static void pack_leqeta(uint8_t *buf, const dpoly *p, const dparams *P) {
    memset(buf, 0, (size_t)P_LEQETA_SZ(P));
    for (int i = 0; i < DN; i++)
        bits_put(buf, (size_t)P->deta_bits * i, (unsigned)P->deta_bits, (DQ + (uint32_t)P->eta - p->c[i]) & 0xff);
}
static void unpack_leqeta(dpoly *p, const uint8_t *buf, const dparams *P) {
    for (int i = 0; i < DN; i++)
        p->c[i] = DQ + (uint32_t)P->eta - bits_get(buf, (size_t)P->deta_bits * i, (unsigned)P->deta_bits);
}
/* pack.go:146-199 PolyUnpackLeGamma1: gamma1 - field, normalised; pack.go:202-254 pack */
static void unpack_legamma1(dpoly *p, const uint8_t *buf, const dparams *P) {
    unsigned w = (unsigned)P->gamma1_bits + 1;
    for (int i = 0; i < DN; i++) {
        uint32_t x = P_GAMMA1(P) - bits_get(buf, (size_t)w * i, w);
        x += (uint32_t)((int32_t)x >> 31) & DQ;
        p->c[i] = x;
    }
}
static void pack_legamma1(uint8_t *buf, const dpoly *p, const dparams *P) {
    unsigned w = (unsigned)P->gamma1_bits + 1;
    memset(buf, 0, (size_t)P_LEGAMMA1_SZ(P));
    for (int i = 0; i < DN; i++) {
        uint32_t x = P_GAMMA1(P) - p->c[i];
        x += (uint32_t)((int32_t)x >> 31) & DQ;
        bits_put(buf, (size_t)w * i, w, x);
    }
}
/* pack.go:256-270 PolyPackW1: 4 bits (gamma1 = 2^19 sets) or 6 bits (ML-DSA-44) */
static void pack_w1(uint8_t *buf, const dpoly *p, const dparams *P) {
    unsigned w = (unsigned)(23 - P->gamma1_bits);
    memset(buf, 0, (size_t)P_W1_SZ(P));
    for (int i = 0; i < DN; i++) bits_put(buf, (size_t)w * i, w, p->c[i]);
}
/* pack.go:94-108 PackHint */
static void pack_hint(uint8_t *buf, const dpoly *h, const dparams *P) {
    unsigned off = 0;
    for (int i = 0; i < P->k; i++) {
        for (int j = 0; j < DN; j++)
            if (h[i].c[j] != 0) buf[off++] = (uint8_t)j;
        buf[P->omega + i] = (uint8_t)off;
    }
    for (; off < (unsigned)P->omega; off++) buf[off] = 0;
}
/* pack.go:113-141 UnpackHint: strict decoding */
static int unpack_hint(dpoly *h, const uint8_t *buf, const dparams *P) {
    unsigned prev = 0;
    for (int i = 0; i < P->k; i++) memset(&h[i], 0, sizeof(dpoly));
    for (int i = 0; i < P->k; i++) {
        unsigned sop = buf[P->omega + i];
        if (sop < prev || sop > (unsigned)P->omega) return 0;
        for (unsigned j = prev; j < sop; j++) {
            if (j > prev && buf[j] <= buf[j - 1]) return 0;
            h[i].c[buf[j]] = 1;
        }
        prev = sop;
    }
    for (unsigned j = prev; j < (unsigned)P->omega; j++)
        if (buf[j] != 0) return 0;
    return 1;
}

/* ---- sample.go --------------------------------------------------------- */
/* sample.go:92-123 PolyDeriveUniform: SHAKE128(seed || LE16(nonce)), 23-bit rejection */
static void dpoly_uniform(dpoly *p, const uint8_t seed[32], uint16_t nonce) {
    uint8_t iv[34], buf[168];
    orc_sponge h;
    memcpy(iv, seed, 32);
    iv[32] = (uint8_t)nonce;
    iv[33] = (uint8_t)(nonce >> 8);
    orc_sponge_init(&h, ORC_SHAKE128_RATE, ORC_DS_SHAKE);
    orc_sponge_absorb(&h, iv, 34);
    int i = 0;
    while (i < DN) {
        orc_sponge_squeeze(&h, buf, 168);
        for (int j = 0; j < 168 && i < DN; j += 3) {
            uint32_t t = (buf[j] | ((uint32_t)buf[j + 1] << 8) | ((uint32_t)buf[j + 2] << 16)) & 0x7fffff;
            if (t < DQ) p->c[i++] = t;
        }
    }
}
/* sample.go:125-175 PolyDeriveUniformLeqEta: SHAKE256(seed64 || LE16(nonce)), nibble rejection */
static void dpoly_leqeta(dpoly *p, const uint8_t seed[64], uint16_t nonce, const dparams *P) {
    uint8_t iv[66], buf[136];
    orc_sponge h;
    memcpy(iv, seed, 64);
    iv[64] = (uint8_t)nonce;
    iv[65] = (uint8_t)(nonce >> 8);
    orc_sponge_init(&h, ORC_SHAKE256_RATE, ORC_DS_SHAKE);
    orc_sponge_absorb(&h, iv, 66);
    int i = 0;
    while (i < DN) {
        orc_sponge_squeeze(&h, buf, 136);
        for (int j = 0; j < 136 && i < DN; j++) {
            uint32_t t1 = buf[j] & 15, t2 = buf[j] >> 4;
            if (P->eta == 2) {
                if (t1 <= 14) { t1 -= ((205 * t1) >> 10) * 5; p->c[i++] = DQ + 2 - t1; }
                if (t2 <= 14 && i < DN) { t2 -= ((205 * t2) >> 10) * 5; p->c[i++] = DQ + 2 - t2; }
            } else {
                if (t1 <= 8) p->c[i++] = DQ + 4 - t1;
                if (t2 <= 8 && i < DN) p->c[i++] = DQ + 4 - t2;
            }
        }
    }
}
/* sample.go:184-196 PolyDeriveUniformLeGamma1: SHAKE256(seed64 || LE16(nonce)) */
static void dpoly_legamma1(dpoly *p, const uint8_t seed[64], uint16_t nonce, const dparams *P) {
    uint8_t iv[66], buf[640];
    memcpy(iv, seed, 64);
    iv[64] = (uint8_t)nonce;
    iv[65] = (uint8_t)(nonce >> 8);
    orc_shake256(buf, (size_t)P_LEGAMMA1_SZ(P), iv, 66);
    unpack_legamma1(p, buf, P);
}
/* sample.go:299-339 PolyDeriveUniformBall: SHAKE256(ctilde), 8 sign bytes, Fisher-Yates */
static void dpoly_ball(dpoly *p, const uint8_t *seed, const dparams *P) {
    uint8_t buf[136];
    orc_sponge h;
    orc_sponge_init(&h, ORC_SHAKE256_RATE, ORC_DS_SHAKE);
    orc_sponge_absorb(&h, seed, (size_t)P->ctilde);
    orc_sponge_squeeze(&h, buf, 136);
    uint64_t signs = 0;
    for (int i = 0; i < 8; i++) signs |= (uint64_t)buf[i] << (8 * i);
    int off = 8;
    memset(p, 0, sizeof *p);
    for (unsigned i = (unsigned)(DN - P->tau); i < DN; i++) {
        unsigned b;
        for (;;) {
            if (off >= 136) { orc_sponge_squeeze(&h, buf, 136); off = 0; }
            b = buf[off++];
            if (b <= i) break;
        }
        p->c[i] = p->c[b];
        p->c[b] = 1;
        p->c[b] ^= (uint32_t)(-(int64_t)(signs & 1)) & (1 | (DQ - 1));
        signs >>= 1;
    }
}

/* ---- rounding.go ------------------------------------------------------- */
static inline void decompose(uint32_t a, uint32_t *a0plusq, uint32_t *a1out, const dparams *P) { /* rounding.go:13-43 */
    uint32_t alpha = 2 * P->gamma2;
    uint32_t a1 = (a + 127) >> 7;
    if (alpha == 523776) {
        a1 = (a1 * 1025 + (1u << 21)) >> 22;
        a1 &= 15;
    } else {
        a1 = (a1 * 11275 + (1u << 23)) >> 24;
        a1 ^= (uint32_t)((int32_t)(43 - a1) >> 31) & a1;
    }
    uint32_t a0 = a - a1 * alpha;
    a0 += (uint32_t)((int32_t)(a0 - (DQ - 1) / 2) >> 31) & DQ;
    *a0plusq = a0;
    *a1out = a1;
}
static inline uint32_t make_hint(uint32_t z0, uint32_t r1, const dparams *P) { /* rounding.go:55-62 */
    if (z0 <= P->gamma2 || z0 > DQ - P->gamma2 || (z0 == DQ - P->gamma2 && r1 == 0)) return 0;
    return 1;
}
static void dpoly_use_hint(dpoly *p, const dpoly *q, const dpoly *hint, const dparams *P) { /* rounding.go:98-135 */
    for (int i = 0; i < DN; i++) {
        uint32_t q0, q1;
        decompose(q->c[i], &q0, &q1, P);
        if (hint->c[i] != 0) {
            if (P->gamma2 == 261888) {
                q1 = (q0 > DQ) ? ((q1 + 1) & 15) : ((q1 - 1) & 15);
            } else {
                if (q0 > DQ) q1 = (q1 == 43) ? 0 : q1 + 1;
                else q1 = (q1 == 0) ? 43 : q1 - 1;
            }
        }
        p->c[i] = q1;
    }
}

/* ---- keys ---------------------------------------------------------------- */
typedef struct {
    uint8_t rho[32], key[32], tr[64];
    dpoly s1[7], s2[8], t0[8];
    dpoly A[8][7];
    dpoly s1h[7], s2h[8], t0h[8];
} dsk;
typedef struct {
    uint8_t rho[32], tr[64];
    dpoly t1[8];
    dpoly A[8][7];
} dpk;

static void mat_derive(dpoly A[8][7], const uint8_t rho[32], const dparams *P) { /* mat.go:15-49 */
    for (int i = 0; i < P->k; i++)
        for (int j = 0; j < P->l; j++) dpoly_uniform(&A[i][j], rho, (uint16_t)((i << 8) + j));
}
static void dot_hat(dpoly *p, const dpoly *a, const dpoly *b, int L) { /* mat.go:52-59 */
    dpoly t;
    memset(p, 0, sizeof *p);
    for (int i = 0; i < L; i++) {
        dpoly_mulhat(&t, &a[i], &b[i]);
        dpoly_add(p, &t, p);
    }
}

size_t orc_mldsa_pk_size(int param) { dparams P; return dil_params(param, &P) ? 0 : (size_t)P_PK_SZ(&P); }
size_t orc_mldsa_sk_size(int param) { dparams P; return dil_params(param, &P) ? 0 : (size_t)P_SK_SZ(&P); }
size_t orc_mldsa_sig_size(int param) { dparams P; return dil_params(param, &P) ? 0 : (size_t)P_SIG_SZ(&P); }

/* dilithium.go:181-267 NewKeyFromSeed + computeT0andT1 (:282-296); packing :131-179 */
int orc_mldsa_keygen(int param, const uint8_t seed[32], uint8_t *pk, uint8_t *skbuf) {
    dparams P;
    if (dil_params(param, &P)) return -1;
    if (!dz_ready) dz_init();
    static __thread dsk sk;
    uint8_t eseed[128], in[34];
    memcpy(in, seed, 32);
    in[32] = (uint8_t)P.k;
    in[33] = (uint8_t)P.l;
    orc_shake256(eseed, 128, in, P.nist ? 34 : 32); /* dilithium.go:189-195 */
    const uint8_t *rho = eseed, *sseed = eseed + 32, *key = eseed + 96;
    mat_derive(sk.A, rho, &P);
    for (int i = 0; i < P.l; i++) dpoly_leqeta(&sk.s1[i], sseed, (uint16_t)i, &P);
    for (int i = 0; i < P.k; i++) dpoly_leqeta(&sk.s2[i], sseed, (uint16_t)(i + P.l), &P);
    for (int i = 0; i < P.l; i++) { sk.s1h[i] = sk.s1[i]; dpoly_ntt(&sk.s1h[i]); }
    dpoly t, t0[8], t1[8];
    for (int i = 0; i < P.k; i++) {
        dot_hat(&t, sk.A[i], sk.s1h, P.l);
        dpoly_reduce_le2q(&t);
        dpoly_invntt(&t);
        dpoly_add(&t, &t, &sk.s2[i]);
        dpoly_normalize(&t);
        for (int j = 0; j < DN; j++) power2round(t.c[j], &t0[i].c[j], &t1[i].c[j]);
    }
    memcpy(pk, rho, 32);
    for (int i = 0; i < P.k; i++) pack_t1(pk + 32 + 320 * i, &t1[i]);
    uint8_t tr[64];
    orc_shake256(tr, (size_t)P.tr, pk, (size_t)P_PK_SZ(&P));
    uint8_t *o = skbuf;
    memcpy(o, rho, 32); o += 32;
    memcpy(o, key, 32); o += 32;
    memcpy(o, tr, (size_t)P.tr); o += P.tr;
    for (int i = 0; i < P.l; i++, o += P_LEQETA_SZ(&P)) pack_leqeta(o, &sk.s1[i], &P);
    for (int i = 0; i < P.k; i++, o += P_LEQETA_SZ(&P)) pack_leqeta(o, &sk.s2[i], &P);
    for (int i = 0; i < P.k; i++, o += 416) pack_t0(o, &t0[i]);
    return 0;
}

/* dilithium.go:149-179 PrivateKey.Unpack */
static void sk_unpack(dsk *sk, const uint8_t *buf, const dparams *P) {
    memcpy(sk->rho, buf, 32);
    memcpy(sk->key, buf + 32, 32);
    memcpy(sk->tr, buf + 64, (size_t)P->tr);
    const uint8_t *o = buf + 64 + P->tr;
    for (int i = 0; i < P->l; i++, o += P_LEQETA_SZ(P)) unpack_leqeta(&sk->s1[i], o, P);
    for (int i = 0; i < P->k; i++, o += P_LEQETA_SZ(P)) unpack_leqeta(&sk->s2[i], o, P);
    for (int i = 0; i < P->k; i++, o += 416) unpack_t0(&sk->t0[i], o);
    mat_derive(sk->A, sk->rho, P);
    for (int i = 0; i < P->k; i++) { sk->t0h[i] = sk->t0[i]; dpoly_ntt(&sk->t0h[i]); }
    for (int i = 0; i < P->l; i++) { sk->s1h[i] = sk->s1[i]; dpoly_ntt(&sk->s1h[i]); }
    for (int i = 0; i < P->k; i++) { sk->s2h[i] = sk->s2[i]; dpoly_ntt(&sk->s2h[i]); }
}

static void absorb_msg(orc_sponge *h, const uint8_t *msg, size_t msglen, const uint8_t *ctx,
                       size_t ctxlen, int internal) {
    if (!internal) { /* mldsa65/dilithium.go:121-129 */
        uint8_t pre[2] = {0, (uint8_t)ctxlen};
        orc_sponge_absorb(h, pre, 2);
        orc_sponge_absorb(h, ctx, ctxlen);
    }
    orc_sponge_absorb(h, msg, msglen);
}

/* dilithium.go:340-470 SignTo */
int orc_mldsa_sign(int param, const uint8_t *skbuf, const uint8_t *msg, size_t msglen,
                   const uint8_t *ctx, size_t ctxlen, const uint8_t rnd[32], int internal, uint8_t *sig) {
    dparams P;
    if (dil_params(param, &P)) return -1;
    if (ctxlen > 255) return -2;
    if (!dz_ready) dz_init();
    static __thread dsk sk;
    sk_unpack(&sk, skbuf, &P);
    uint8_t mu[64], rhop[64], w1p[192 * 8], ct[64];
    orc_sponge h;
    orc_sponge_init(&h, ORC_SHAKE256_RATE, ORC_DS_SHAKE);
    orc_sponge_absorb(&h, sk.tr, (size_t)P.tr);
    absorb_msg(&h, msg, msglen, ctx, ctxlen, internal || !P.nist); /* round 3: mu = CRH(tr || msg), mode3/dilithium.go:54-66 */
    orc_sponge_squeeze(&h, mu, 64);
    orc_sponge_init(&h, ORC_SHAKE256_RATE, ORC_DS_SHAKE);
    orc_sponge_absorb(&h, sk.key, 32);
    if (P.nist) orc_sponge_absorb(&h, rnd, 32); /* dilithium.go:360-362 */
    orc_sponge_absorb(&h, mu, 64);
    orc_sponge_squeeze(&h, rhop, 64);

    dpoly y[7], yh[7], z[7], w[8], w0[8], w1[8], w0mcs2[8], ct0[8], hint[8], ch, tmp;
    uint16_t ynonce = 0;
    for (int attempt = 1;; attempt++) {
        if (attempt >= 576) return -3;
        for (int i = 0; i < P.l; i++) dpoly_legamma1(&y[i], rhop, (uint16_t)(ynonce + i), &P);
        ynonce = (uint16_t)(ynonce + P.l);
        for (int i = 0; i < P.l; i++) { yh[i] = y[i]; dpoly_ntt(&yh[i]); }
        for (int i = 0; i < P.k; i++) {
            dot_hat(&w[i], sk.A[i], yh, P.l);
            dpoly_reduce_le2q(&w[i]);
            dpoly_invntt(&w[i]);
            dpoly_normalize_le2q(&w[i]);
            for (int j = 0; j < DN; j++) decompose(w[i].c[j], &w0[i].c[j], &w1[i].c[j], &P);
            pack_w1(w1p + P_W1_SZ(&P) * i, &w1[i], &P);
        }
        orc_sponge_init(&h, ORC_SHAKE256_RATE, ORC_DS_SHAKE);
        orc_sponge_absorb(&h, mu, 64);
        orc_sponge_absorb(&h, w1p, (size_t)(P_W1_SZ(&P) * P.k));
        orc_sponge_squeeze(&h, ct, (size_t)P.ctilde);
        dpoly_ball(&ch, ct, &P);
        dpoly_ntt(&ch);

        int bad = 0;
        for (int i = 0; i < P.k; i++) {
            dpoly_mulhat(&tmp, &ch, &sk.s2h[i]);
            dpoly_invntt(&tmp);
            dpoly_sub(&w0mcs2[i], &w0[i], &tmp);
            dpoly_normalize(&w0mcs2[i]);
            bad |= dpoly_exceeds(&w0mcs2[i], P.gamma2 - P_BETA(&P));
        }
        if (bad) continue;
        for (int i = 0; i < P.l; i++) {
            dpoly_mulhat(&z[i], &ch, &sk.s1h[i]);
            dpoly_invntt(&z[i]);
            dpoly_add(&z[i], &z[i], &y[i]);
            dpoly_normalize(&z[i]);
            bad |= dpoly_exceeds(&z[i], P_GAMMA1(&P) - P_BETA(&P));
        }
        if (bad) continue;
        for (int i = 0; i < P.k; i++) {
            dpoly_mulhat(&ct0[i], &ch, &sk.t0h[i]);
            dpoly_invntt(&ct0[i]);
            dpoly_normalize_le2q(&ct0[i]);
            bad |= dpoly_exceeds(&ct0[i], P.gamma2);
        }
        if (bad) continue;
        uint32_t pop = 0;
        for (int i = 0; i < P.k; i++) {
            dpoly_add(&tmp, &w0mcs2[i], &ct0[i]);
            dpoly_normalize_le2q(&tmp);
            for (int j = 0; j < DN; j++) { /* rounding.go:83-90 PolyMakeHint */
                uint32_t hb = make_hint(tmp.c[j], w1[i].c[j], &P);
                hint[i].c[j] = hb;
                pop += hb;
            }
        }
        if (pop > (uint32_t)P.omega) continue;
        break;
    }
    /* dilithium.go:84-88 unpackedSignature.Pack */
    memcpy(sig, ct, (size_t)P.ctilde);
    uint8_t *o = sig + P.ctilde;
    for (int i = 0; i < P.l; i++, o += P_LEGAMMA1_SZ(&P)) pack_legamma1(o, &z[i], &P);
    pack_hint(o, hint, &P);
    return 0;
}

/* dilithium.go:114-126 PublicKey.Unpack + :273-332 Verify (+ :90-105 sig.Unpack) */
int orc_mldsa_verify(int param, const uint8_t *pkbuf, const uint8_t *msg, size_t msglen,
                     const uint8_t *ctx, size_t ctxlen, int internal, const uint8_t *sig, size_t siglen) {
    dparams P;
    if (dil_params(param, &P)) return -1;
    if (!dz_ready) dz_init();
    if (!P.nist) internal = 1; /* round 3 hashes the bare message: mode3/dilithium.go:68-75 */
    if (!internal && ctxlen > 255) return 0; /* mldsa65/dilithium.go:116-118 */
    if (siglen != (size_t)P_SIG_SZ(&P)) return 0;
    static __thread dpk pk;
    memcpy(pk.rho, pkbuf, 32);
    for (int i = 0; i < P.k; i++) unpack_t1(&pk.t1[i], pkbuf + 32 + 320 * i);
    mat_derive(pk.A, pk.rho, &P);
    orc_shake256(pk.tr, (size_t)P.tr, pkbuf, (size_t)P_PK_SZ(&P));

    dpoly z[7], hint[8], ch, az, t, w1;
    const uint8_t *o = sig + P.ctilde;
    for (int i = 0; i < P.l; i++, o += P_LEGAMMA1_SZ(&P)) {
        unpack_legamma1(&z[i], o, &P);
        if (dpoly_exceeds(&z[i], P_GAMMA1(&P) - P_BETA(&P))) return 0;
    }
    if (!unpack_hint(hint, o, &P)) return 0;

    uint8_t mu[64], w1p[192 * 8], cp[64];
    orc_sponge h;
    orc_sponge_init(&h, ORC_SHAKE256_RATE, ORC_DS_SHAKE);
    orc_sponge_absorb(&h, pk.tr, (size_t)P.tr);
    absorb_msg(&h, msg, msglen, ctx, ctxlen, internal);
    orc_sponge_squeeze(&h, mu, 64);

    for (int i = 0; i < P.l; i++) dpoly_ntt(&z[i]);
    dpoly_ball(&ch, sig, &P);
    dpoly_ntt(&ch);
    for (int i = 0; i < P.k; i++) {
        dot_hat(&az, pk.A[i], z, P.l);
        for (int j = 0; j < DN; j++) t.c[j] = pk.t1[i].c[j] << DD; /* poly.go:97-101 mulBy2toD */
        dpoly_ntt(&t);
        dpoly_mulhat(&t, &t, &ch);
        dpoly_sub(&t, &az, &t);
        dpoly_reduce_le2q(&t);
        dpoly_invntt(&t);
        dpoly_normalize_le2q(&t);
        dpoly_use_hint(&w1, &t, &hint[i], &P);
        pack_w1(w1p + P_W1_SZ(&P) * i, &w1, &P);
    }
    orc_sponge_init(&h, ORC_SHAKE256_RATE, ORC_DS_SHAKE);
    orc_sponge_absorb(&h, mu, 64);
    orc_sponge_absorb(&h, w1p, (size_t)(P_W1_SZ(&P) * P.k));
    orc_sponge_squeeze(&h, cp, (size_t)P.ctilde);
    return memcmp(cp, sig, (size_t)P.ctilde) == 0;
}

/* ---- primitives for unit-level parity tests ------------------------------ */
void orc_dilithium_ntt(uint32_t p[256]) { if (!dz_ready) dz_init(); dpoly_ntt((dpoly *)p); }
void orc_dilithium_invntt(uint32_t p[256]) { if (!dz_ready) dz_init(); dpoly_invntt((dpoly *)p); }
void orc_dilithium_normalize(uint32_t p[256]) { dpoly_normalize((dpoly *)p); }
void orc_dilithium_uniform(uint32_t p[256], const uint8_t seed[32], uint16_t nonce) { dpoly_uniform((dpoly *)p, seed, nonce); }
int orc_dilithium_ball(int param, uint32_t p[256], const uint8_t *ctilde) {
    dparams P;
    if (dil_params(param, &P)) return -1;
    dpoly_ball((dpoly *)p, ctilde, &P);
    return 0;
}
const uint32_t *orc_dilithium_zetas(void) { if (!dz_ready) dz_init(); return DZETAS; }
