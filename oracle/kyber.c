/* oracle/kyber.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Scalar C restatement of the reference's *generic Go* ML-KEM path (never the
 * AVX2 assembler).  Each function cites the reference file:line it follows
 * (paths relative to the cloudflare/circl tree).  Parity is PINNED: see
 * tests/test_oracle_mlkem.py (NIST ACVP keyGen/encapDecap vectors, the three
 * ML-KEM KAT transcript hashes of kem/kyber/kat_test.go:31-33, and the fixed
 * sampler vectors of pke/kyber/internal/common/sample_test.go).
 *
 * Nothing under circl_amd/ may include, link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as checker.
 */
#include "oracle.h"
#include "keccak.h"
#include <string.h>

#define KQ 3329
#define KN 256

typedef struct { int16_t c[KN]; } poly;

typedef struct {
    int k, eta1, du, dv;
} kparams;

static int kyber_params(int param, kparams *p) {
    /* pke/kyber/kyber{512,768,1024}/internal/params.go:9-21 */
    switch (param) {
    case 512:  *p = (kparams){2, 3, 10, 4}; return 0;
    case 768:  *p = (kparams){3, 2, 10, 4}; return 0;
    case 1024: *p = (kparams){4, 2, 11, 5}; return 0;
    }
    return -1;
}

/* ---- field.go ---------------------------------------------------------- */

/* field.go:4-32: x*R^-1 mod q, R = 2^16, q^-1 = 62209 mod 2^16, result in (-q, q). */
static inline int16_t mont_reduce(int32_t x) {
    int16_t m = (int16_t)(uint16_t)((uint32_t)x * 62209u);
    return (int16_t)((uint32_t)(x - (int32_t)m * KQ) >> 16);
}
/* field.go:35-39: R^2 mod q = 1353 */
static inline int16_t to_mont(int16_t x) { return mont_reduce((int32_t)x * 1353); }
/* field.go:45-64: x - floor(x*20159 / 2^26)*q, result in [0, q] */
static inline int16_t barrett_reduce(int16_t x) {
    return (int16_t)(x - (int16_t)(((int32_t)x * 20159) >> 26) * KQ);
}
/* field.go:67-74 */
static inline int16_t csubq(int16_t x) {
    x = (int16_t)(x - KQ);
    x = (int16_t)(x + ((x >> 15) & KQ));
    return x;
}

/* ---- ntt.go ------------------------------------------------------------ */

/* ntt.go:16-28 tabulates Zetas[i] = 17^brv7(i) * 2^16 mod q; generated here. */
static int16_t ZETAS[128];
static int zetas_ready;

__attribute__((constructor)) static void zetas_init(void) {
    for (int i = 0; i < 128; i++) {
        int brv = 0;
        for (int b = 0; b < 7; b++) brv |= ((i >> b) & 1) << (6 - b);
        uint32_t z = 1;
        for (int e = 0; e < brv; e++) z = z * 17 % KQ;
        ZETAS[i] = (int16_t)((z << 16) % KQ);
    }
    zetas_ready = 1;
}

/* ntt.go:117-134 nttGeneric: 7 Cooley-Tukey layers, l = 128..2 */
static void poly_ntt(poly *p) {
    int k = 0;
    for (int l = KN / 2; l > 1; l >>= 1) {
        for (int off = 0; off < KN - l; off += 2 * l) {
            k++;
            int32_t zeta = ZETAS[k];
            for (int j = off; j < off + l; j++) {
                int16_t t = mont_reduce(zeta * (int32_t)p->c[j + l]);
                p->c[j + l] = (int16_t)(p->c[j] - t);
                p->c[j] = (int16_t)(p->c[j] + t);
            }
        }
    }
}

/* ntt.go:145-193 invNTTGeneric: 7 Gentleman-Sande layers with the same zetas
 * backwards, then multiply by 1441 = 128^-1 R^2.  The reference Barrett-reduces
 * only a lazy subset of coefficients after layers 3..6 (InvNTTReductions,
 * ntt.go:38-50); reducing *every* coefficient after every layer is congruent
 * mod q and equally overflow-free, and the reference itself compares inverse
 * transforms only after Normalize (ntt_test.go:64-81), as do we. */
static void poly_invntt(poly *p) {
    int k = 127;
    for (int l = 2; l < KN; l <<= 1) {
        for (int off = 0; off < KN - l; off += 2 * l) {
            int32_t min_zeta = ZETAS[k--];
            for (int j = off; j < off + l; j++) {
                int16_t t = (int16_t)(p->c[j + l] - p->c[j]);
                p->c[j] = (int16_t)(p->c[j] + p->c[j + l]);
                p->c[j + l] = mont_reduce(min_zeta * (int32_t)t);
            }
        }
        for (int i = 0; i < KN; i++) p->c[i] = barrett_reduce(p->c[i]);
    }
    for (int j = 0; j < KN; j++) p->c[j] = mont_reduce(1441 * (int32_t)p->c[j]);
}

/* ---- poly.go ----------------------------------------------------------- */

static void poly_add(poly *r, const poly *a, const poly *b) { /* poly.go:13-17 */
    for (int i = 0; i < KN; i++) r->c[i] = (int16_t)(a->c[i] + b->c[i]);
}
static void poly_sub(poly *r, const poly *a, const poly *b) { /* poly.go:20-24 */
    for (int i = 0; i < KN; i++) r->c[i] = (int16_t)(a->c[i] - b->c[i]);
}
static void poly_barrett(poly *p) { /* poly.go:28-32 */
    for (int i = 0; i < KN; i++) p->c[i] = barrett_reduce(p->c[i]);
}
static void poly_normalize(poly *p) { /* poly.go:35-39 */
    for (int i = 0; i < KN; i++) p->c[i] = csubq(barrett_reduce(p->c[i]));
}
static void poly_tomont(poly *p) { /* poly.go:48-52 */
    for (int i = 0; i < KN; i++) p->c[i] = to_mont(p->c[i]);
}

/* poly.go:63-100 mulHatGeneric: 128 products mod (x^2 -/+ zeta), zeta = Zetas[64+i/4] */
static void poly_mulhat(poly *p, const poly *a, const poly *b) {
    int k = 64;
    for (int i = 0; i < KN; i += 4) {
        int32_t zeta = ZETAS[k++];
        int16_t p0 = mont_reduce((int32_t)a->c[i + 1] * b->c[i + 1]);
        p0 = mont_reduce((int32_t)p0 * zeta);
        p0 = (int16_t)(p0 + mont_reduce((int32_t)a->c[i] * b->c[i]));
        int16_t p1 = mont_reduce((int32_t)a->c[i] * b->c[i + 1]);
        p1 = (int16_t)(p1 + mont_reduce((int32_t)a->c[i + 1] * b->c[i]));
        p->c[i] = p0;
        p->c[i + 1] = p1;
        int16_t p2 = mont_reduce((int32_t)a->c[i + 3] * b->c[i + 3]);
        p2 = (int16_t)(-mont_reduce((int32_t)p2 * zeta));
        p2 = (int16_t)(p2 + mont_reduce((int32_t)a->c[i + 2] * b->c[i + 2]));
        int16_t p3 = mont_reduce((int32_t)a->c[i + 2] * b->c[i + 3]);
        p3 = (int16_t)(p3 + mont_reduce((int32_t)a->c[i + 3] * b->c[i + 2]));
        p->c[i + 2] = p2;
        p->c[i + 3] = p3;
    }
}

/* poly.go:106-117 Pack (12-bit little-endian, two coefficients per 3 bytes) */
static void poly_pack(uint8_t *buf, const poly *p) {
    for (int i = 0; i < 128; i++) {
        uint16_t t0 = (uint16_t)p->c[2 * i], t1 = (uint16_t)p->c[2 * i + 1];
        buf[3 * i] = (uint8_t)t0;
        buf[3 * i + 1] = (uint8_t)((t0 >> 8) | (t1 << 4));
        buf[3 * i + 2] = (uint8_t)(t1 >> 4);
    }
}
/* poly.go:123-129 Unpack */
static void poly_unpack(poly *p, const uint8_t *buf) {
    for (int i = 0; i < 128; i++) {
        p->c[2 * i] = (int16_t)(buf[3 * i] | ((buf[3 * i + 1] << 8) & 0xfff));
        p->c[2 * i + 1] = (int16_t)((buf[3 * i + 1] >> 4) | (buf[3 * i + 2] << 4));
    }
}

/* poly.go:134-145 DecompressMessage */
static void poly_from_msg(poly *p, const uint8_t m[32]) {
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 8; j++) {
            int bit = (m[i] >> j) & 1;
            p->c[8 * i + j] = (int16_t)(-bit & ((KQ + 1) / 2));
        }
}
/* poly.go:150-165 CompressMessageTo */
static void poly_to_msg(uint8_t m[32], const poly *p) {
    for (int i = 0; i < 32; i++) {
        m[i] = 0;
        for (int j = 0; j < 8; j++) {
            int16_t x = (int16_t)(1664 - p->c[8 * i + j]);
            x = (int16_t)((x >> 15) ^ x);
            x = (int16_t)(x - 832);
            m[i] |= (uint8_t)((((uint16_t)x >> 15) & 1) << j);
        }
    }
}

/* Little-endian bit-stream writer/reader: the byte formulas at poly.go:170-332 are
 * exactly "d bits per coefficient, least-significant bit first". */
static void bits_put(uint8_t *buf, size_t bitpos, unsigned d, uint32_t v) {
    for (unsigned b = 0; b < d; b++, bitpos++) {
        if ((v >> b) & 1) buf[bitpos >> 3] |= (uint8_t)(1u << (bitpos & 7));
    }
}
static uint32_t bits_get(const uint8_t *buf, size_t bitpos, unsigned d) {
    uint32_t v = 0;
    for (unsigned b = 0; b < d; b++, bitpos++) v |= (uint32_t)((buf[bitpos >> 3] >> (bitpos & 7)) & 1) << b;
    return v;
}

/* poly.go:248-332 CompressTo: round(x * 2^d / q) mod 2^d with the reference's
 * multiply-shift constants (315 / 2^20 for d<=5, 20642679 / 2^36 for d>=10). */
static void poly_compress(uint8_t *m, const poly *p, int d) {
    memset(m, 0, (size_t)(32 * d));
    for (int i = 0; i < KN; i++) {
        uint32_t x = (uint32_t)(uint16_t)p->c[i], t;
        if (d == 4 || d == 5)
            t = ((((x << d) + KQ / 2) * 315) >> 20) & ((1u << d) - 1);
        else
            t = (uint32_t)(((uint64_t)((x << d) + KQ / 2) * 20642679ull) >> 36) & ((1u << d) - 1);
        bits_put(m, (size_t)i * (size_t)d, (unsigned)d, t);
    }
}
/* poly.go:170-243 Decompress: (2^(d-1) + t*q) >> d */
static void poly_decompress(poly *p, const uint8_t *m, int d) {
    for (int i = 0; i < KN; i++) {
        uint32_t t = bits_get(m, (size_t)i * (size_t)d, (unsigned)d);
        p->c[i] = (int16_t)(((1u << (d - 1)) + t * KQ) >> d);
    }
}

/* ---- sample.go --------------------------------------------------------- */

/* sample.go:31-62 DeriveNoise3 / 67-95 DeriveNoise2: SHAKE256(seed || nonce), CBD */
static void poly_noise(poly *p, const uint8_t seed[32], uint8_t nonce, int eta) {
    uint8_t in[33], buf[192];
    memcpy(in, seed, 32);
    in[32] = nonce;
    orc_shake256(buf, (size_t)(64 * eta), in, 33);
    for (int i = 0; i < KN; i++) {
        int a = 0, b = 0;
        for (int j = 0; j < eta; j++) {
            int bit = 2 * eta * i + j;
            a += (buf[bit >> 3] >> (bit & 7)) & 1;
            bit += eta;
            b += (buf[bit >> 3] >> (bit & 7)) & 1;
        }
        p->c[i] = (int16_t)(a - b);
    }
}

/* sample.go:192-236 DeriveUniform: SHAKE128(seed || x || y), 12-bit rejection,
 * candidates t1 then t2 from each 3-byte group. */
static void poly_uniform(poly *p, const uint8_t seed[32], uint8_t x, uint8_t y) {
    uint8_t in[34], buf[168];
    orc_sponge h;
    memcpy(in, seed, 32);
    in[32] = x;
    in[33] = y;
    orc_sponge_init(&h, ORC_SHAKE128_RATE, ORC_DS_SHAKE);
    orc_sponge_absorb(&h, in, 34);
    int i = 0;
    while (i < KN) {
        orc_sponge_squeeze(&h, buf, 168);
        for (int j = 0; j < 168 && i < KN; j += 3) {
            uint16_t t1 = (uint16_t)((buf[j] | (buf[j + 1] << 8)) & 0xfff);
            uint16_t t2 = (uint16_t)(((buf[j + 1] >> 4) | (buf[j + 2] << 4)) & 0xfff);
            if (t1 < KQ) p->c[i++] = (int16_t)t1;
            if (t2 < KQ && i < KN) p->c[i++] = (int16_t)t2;
        }
    }
}

/* ---- kyber768/internal/{mat,vec,cpapke}.go ------------------------------ */

typedef struct { poly v[4]; } pvec;
typedef struct { pvec r[4]; } pmat;

/* mat.go:13-74 Derive: transpose=true samples m[i][j] from (x=i, y=j), else (x=j, y=i) */
static void mat_derive(pmat *m, const uint8_t rho[32], int transpose, int K) {
    for (int i = 0; i < K; i++)
        for (int j = 0; j < K; j++) {
            if (transpose) poly_uniform(&m->r[i].v[j], rho, (uint8_t)i, (uint8_t)j);
            else poly_uniform(&m->r[i].v[j], rho, (uint8_t)j, (uint8_t)i);
        }
}
/* vec.go:30-37 PolyDotHat */
static void vec_dothat(poly *p, const pvec *a, const pvec *b, int K) {
    poly t;
    memset(p, 0, sizeof *p);
    for (int i = 0; i < K; i++) {
        poly_mulhat(&t, &a->v[i], &b->v[i]);
        poly_add(p, &t, p);
    }
}

/* cpapke.go:58-63 Unpack (th normalised, rho copied); the matrix is derived by callers. */
static void pk_unpack_th(pvec *th, const uint8_t *buf, int K) {
    for (int i = 0; i < K; i++) {
        poly_unpack(&th->v[i], buf + 384 * i);
        poly_normalize(&th->v[i]);
    }
}

/* cpapke.go:66-110 NewKeyFromSeed (seed already carries the FIPS 203 domain byte K:
 * pke/kyber/kyber768/kyber.go:77-86 NewKeyFromSeedMLKEM). */
static void kpke_keygen(uint8_t *ek, uint8_t *sk_packed, const uint8_t *seed, size_t seedlen,
                        const kparams *P) {
    int K = P->k;
    uint8_t es[64];
    orc_sha3_512(es, seed, seedlen);
    const uint8_t *rho = es, *sigma = es + 32;
    static __thread pmat A;
    pvec sh, eh, th;
    mat_derive(&A, rho, 0, K);
    for (int i = 0; i < K; i++) {
        poly_noise(&sh.v[i], sigma, (uint8_t)i, P->eta1);
        poly_ntt(&sh.v[i]);
        poly_normalize(&sh.v[i]);
    }
    for (int i = 0; i < K; i++) {
        poly_noise(&eh.v[i], sigma, (uint8_t)(K + i), P->eta1);
        poly_ntt(&eh.v[i]);
    }
    for (int i = 0; i < K; i++) {
        vec_dothat(&th.v[i], &A.r[i], &sh, K);
        poly_tomont(&th.v[i]);
        poly_add(&th.v[i], &th.v[i], &eh.v[i]);
        poly_normalize(&th.v[i]);
    }
    for (int i = 0; i < K; i++) {
        poly_pack(ek + 384 * i, &th.v[i]);
        poly_pack(sk_packed + 384 * i, &sh.v[i]);
    }
    memcpy(ek + 384 * K, rho, 32);
}

/* cpapke.go:137-181 EncryptTo */
static void kpke_encrypt_parsed(uint8_t *ct, const pvec *th, const pmat *AT, const uint8_t pt[32],
                                const uint8_t seed[32], const kparams *P);
static void kpke_encrypt(uint8_t *ct, const uint8_t *ekbuf, const uint8_t pt[32],
                         const uint8_t seed[32], const kparams *P) {
    int K = P->k;
    static __thread pmat AT;
    pvec th;
    pk_unpack_th(&th, ekbuf, K);
    mat_derive(&AT, ekbuf + 384 * K, 1, K);
    kpke_encrypt_parsed(ct, &th, &AT, pt, seed, P);
}
/* EncryptTo on a parsed key: cpapke.go:19-25 PublicKey{rho, th, aT} caches th and A^T; :137-181 is the work left per message */
static void kpke_encrypt_parsed(uint8_t *ct, const pvec *thp, const pmat *ATp, const uint8_t pt[32],
                                const uint8_t seed[32], const kparams *P) {
    int K = P->k;
    pvec rh, e1, u;
    poly e2, v, m;
#define th (*thp)
#define AT (*ATp)
    for (int i = 0; i < K; i++) {
        poly_noise(&rh.v[i], seed, (uint8_t)i, P->eta1);
        poly_ntt(&rh.v[i]);
        poly_barrett(&rh.v[i]);
    }
    for (int i = 0; i < K; i++) poly_noise(&e1.v[i], seed, (uint8_t)(K + i), 2); /* Eta2 = 2 */
    poly_noise(&e2, seed, (uint8_t)(2 * K), 2);
    for (int i = 0; i < K; i++) {
        vec_dothat(&u.v[i], &AT.r[i], &rh, K);
        poly_barrett(&u.v[i]);
        poly_invntt(&u.v[i]);
        poly_add(&u.v[i], &u.v[i], &e1.v[i]);
    }
    vec_dothat(&v, &th, &rh, K);
    poly_barrett(&v);
    poly_invntt(&v);
    poly_from_msg(&m, pt);
    poly_add(&v, &v, &m);
    poly_add(&v, &v, &e2);
    for (int i = 0; i < K; i++) {
        poly_normalize(&u.v[i]);
        poly_compress(ct + 32 * P->du * i, &u.v[i], P->du);
    }
    poly_normalize(&v);
    poly_compress(ct + 32 * P->du * K, &v, P->dv);
#undef th
#undef AT
}

/* cpapke.go:113-130 DecryptTo (sh = Unpack + Normalize, cpapke.go:33-36) */
static void kpke_decrypt(uint8_t pt[32], const uint8_t *skbuf, const uint8_t *ct, const kparams *P) {
    int K = P->k;
    pvec sh, u;
    poly v, m;
    for (int i = 0; i < K; i++) {
        poly_unpack(&sh.v[i], skbuf + 384 * i);
        poly_normalize(&sh.v[i]);
        poly_decompress(&u.v[i], ct + 32 * P->du * i, P->du);
        poly_ntt(&u.v[i]);
    }
    poly_decompress(&v, ct + 32 * P->du * K, P->dv);
    vec_dothat(&m, &sh, &u, K);
    poly_barrett(&m);
    poly_invntt(&m);
    poly_sub(&m, &v, &m);
    poly_normalize(&m);
    poly_to_msg(pt, &m);
}

/* ---- kem/mlkem/mlkem768/kyber.go ---------------------------------------- */

size_t orc_mlkem_ek_size(int param) { kparams P; return kyber_params(param, &P) ? 0 : (size_t)(384 * P.k + 32); }
size_t orc_mlkem_dk_size(int param) { kparams P; return kyber_params(param, &P) ? 0 : (size_t)(768 * P.k + 96); }
size_t orc_mlkem_ct_size(int param) { kparams P; return kyber_params(param, &P) ? 0 : (size_t)(32 * (P.du * P.k + P.dv)); }

/* kyber.go:57-78 NewKeyFromSeed (seed = d || z); dk layout kyber.go:189-201 */
int orc_mlkem_keygen(int param, const uint8_t seed[64], uint8_t *ek, uint8_t *dk) {
    kparams P;
    if (kyber_params(param, &P)) return -1;
    if (!zetas_ready) zetas_init();
    int K = P.k;
    uint8_t seed2[33];
    memcpy(seed2, seed, 32);
    seed2[32] = (uint8_t)K;
    kpke_keygen(ek, dk, seed2, 33, &P);
    memcpy(dk + 384 * K, ek, (size_t)(384 * K + 32));
    orc_sha3_256(dk + 768 * K + 32, ek, (size_t)(384 * K + 32));
    memcpy(dk + 768 * K + 64, seed + 32, 32);
    return 0;
}

/* kyber.go:247-263 PublicKey.Unpack -> cpapke.go:45-55 UnpackMLKEM (re-pack and
 * compare: any 12-bit coefficient >= q is rejected with kem.ErrPubKey), then
 * kyber.go:103-137 EncapsulateTo.  Returns 0 or ORC_ERR_PUBKEY. */
int orc_mlkem_encaps(int param, const uint8_t *ek, const uint8_t m[32], uint8_t *ct, uint8_t ss[32]) {
    kparams P;
    if (kyber_params(param, &P)) return -1;
    if (!zetas_ready) zetas_init();
    int K = P.k;
    {
        pvec th;
        uint8_t buf2[384 * 4];
        pk_unpack_th(&th, ek, K);
        for (int i = 0; i < K; i++) poly_pack(buf2 + 384 * i, &th.v[i]);
        if (memcmp(buf2, ek, (size_t)(384 * K)) != 0) return ORC_ERR_PUBKEY;
    }
    uint8_t g_in[64], kr[64];
    memcpy(g_in, m, 32);
    orc_sha3_256(g_in + 32, ek, (size_t)(384 * K + 32));
    orc_sha3_512(kr, g_in, 64);
    kpke_encrypt(ct, ek, m, kr + 32, &P);
    memcpy(ss, kr, 32);
    return 0;
}

/* n encapsulations to ONE parsed key: kyber.go:247-263 Unpack once (th, A^T, H(ek): kyber.go:39-43), then
 * kyber.go:103-137 EncapsulateTo per message -- the shape of the reference's BenchmarkEncapsulate
 * (kem/schemes/schemes_test.go:28-38).  Returns 0 or ORC_ERR_PUBKEY (then nothing is written). */
int orc_mlkem_encaps_cached(int param, const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss, size_t n) {
    kparams P;
    if (kyber_params(param, &P)) return -1;
    if (!zetas_ready) zetas_init();
    int K = P.k;
    size_t ctsz = (size_t)(32 * (P.du * K + P.dv));
    static __thread pmat AT;
    pvec th;
    uint8_t buf2[384 * 4], hpk[32];
    pk_unpack_th(&th, ek, K);
    for (int i = 0; i < K; i++) poly_pack(buf2 + 384 * i, &th.v[i]);
    if (memcmp(buf2, ek, (size_t)(384 * K)) != 0) return ORC_ERR_PUBKEY;
    mat_derive(&AT, ek + 384 * K, 1, K);
    orc_sha3_256(hpk, ek, (size_t)(384 * K + 32));
    for (size_t i = 0; i < n; i++) {
        uint8_t g_in[64], kr[64];
        memcpy(g_in, m + 32 * i, 32);
        memcpy(g_in + 32, hpk, 32);
        orc_sha3_512(kr, g_in, 64);
        kpke_encrypt_parsed(ct + ctsz * i, &th, &AT, m + 32 * i, kr + 32, &P);
        memcpy(ss + 32 * i, kr, 32);
    }
    return 0;
}

/* kyber.go:209-230 PrivateKey.Unpack (H(ek) must equal the stored hash, else
 * kem.ErrPrivKey) then kyber.go:144-184 DecapsulateTo (implicit rejection). */
int orc_mlkem_decaps(int param, const uint8_t *dk, const uint8_t *ct, uint8_t ss[32]) {
    kparams P;
    if (kyber_params(param, &P)) return -1;
    if (!zetas_ready) zetas_init();
    int K = P.k;
    size_t eksz = (size_t)(384 * K + 32), ctsz = (size_t)(32 * (P.du * K + P.dv));
    const uint8_t *ek = dk + 384 * K, *hpk = dk + 768 * K + 32, *z = dk + 768 * K + 64;
    uint8_t h[32];
    orc_sha3_256(h, ek, eksz);
    if (memcmp(h, hpk, 32) != 0) return ORC_ERR_PRIVKEY;
    uint8_t g_in[64], kr2[64], ct2[1568], ss2[32];
    kpke_decrypt(g_in, dk, ct, &P);
    memcpy(g_in + 32, hpk, 32);
    orc_sha3_512(kr2, g_in, 64);
    kpke_encrypt(ct2, ek, g_in, kr2 + 32, &P);
    orc_sponge prf;
    orc_sponge_init(&prf, ORC_SHAKE256_RATE, ORC_DS_SHAKE);
    orc_sponge_absorb(&prf, z, 32);
    orc_sponge_absorb(&prf, ct, ctsz);
    orc_sponge_squeeze(&prf, ss2, 32);
    /* subtle.ConstantTimeCopy(ConstantTimeCompare(ct, ct2), ss2, kr2[:32]) */
    uint8_t diff = 0;
    for (size_t i = 0; i < ctsz; i++) diff |= (uint8_t)(ct[i] ^ ct2[i]);
    uint8_t mask = (uint8_t)(((uint32_t)diff - 1) >> 8); /* 0xff when equal */
    for (int i = 0; i < 32; i++) ss[i] = (uint8_t)((kr2[i] & mask) | (ss2[i] & ~mask));
    return 0;
}

/* ---- kem/kyber/kyber768/kyber.go: round-3 Kyber ("Kyber512/768/1024"), SURVEY 8f row f3 ---- */

/* kyber.go:60-82 NewKeyFromSeed: the K-PKE seed carries no domain byte (cpapke.go:66-76); same dk layout */
int orc_kyber_r3_keygen(int param, const uint8_t seed[64], uint8_t *ek, uint8_t *dk) {
    kparams P;
    if (kyber_params(param, &P)) return -1;
    if (!zetas_ready) zetas_init();
    int K = P.k;
    kpke_keygen(ek, dk, seed, 32, &P);
    memcpy(dk + 384 * K, ek, (size_t)(384 * K + 32));
    orc_sha3_256(dk + 768 * K + 32, ek, (size_t)(384 * K + 32));
    memcpy(dk + 768 * K + 64, seed + 32, 32);
    return 0;
}

/* kyber.go:248-262 PublicKey.Unpack (no canonical check: coefficients are reduced, H(pk) binds the raw
 * bytes) then kyber.go:105-154 EncapsulateTo: m = H(seed); (K', r) = G(m || H(pk)); c = Enc(pk, m, r);
 * K = KDF(K' || H(c)). */
int orc_kyber_r3_encaps(int param, const uint8_t *ek, const uint8_t seed[32], uint8_t *ct, uint8_t ss[32]) {
    kparams P;
    if (kyber_params(param, &P)) return -1;
    if (!zetas_ready) zetas_init();
    int K = P.k;
    size_t ctsz = (size_t)(32 * (P.du * K + P.dv));
    uint8_t g_in[64], kr[64];
    orc_sha3_256(g_in, seed, 32);
    orc_sha3_256(g_in + 32, ek, (size_t)(384 * K + 32));
    orc_sha3_512(kr, g_in, 64);
    kpke_encrypt(ct, ek, g_in, kr + 32, &P);
    orc_sha3_256(kr + 32, ct, ctsz);
    orc_shake256(ss, 32, kr, 64);
    return 0;
}

/* kyber.go:215-232 PrivateKey.Unpack (no hash check) then kyber.go:156-197 DecapsulateTo */
int orc_kyber_r3_decaps(int param, const uint8_t *dk, const uint8_t *ct, uint8_t ss[32]) {
    kparams P;
    if (kyber_params(param, &P)) return -1;
    if (!zetas_ready) zetas_init();
    int K = P.k;
    size_t ctsz = (size_t)(32 * (P.du * K + P.dv));
    const uint8_t *ek = dk + 384 * K, *hpk = dk + 768 * K + 32, *z = dk + 768 * K + 64;
    uint8_t g_in[64], kr2[64], ct2[1568];
    kpke_decrypt(g_in, dk, ct, &P);
    memcpy(g_in + 32, hpk, 32);
    orc_sha3_512(kr2, g_in, 64);
    kpke_encrypt(ct2, ek, g_in, kr2 + 32, &P);
    orc_sha3_256(kr2 + 32, ct, ctsz);
    /* subtle.ConstantTimeCopy(1 - ConstantTimeCompare(ct, ct2), kr2[:32], z) */
    uint8_t diff = 0;
    for (size_t i = 0; i < ctsz; i++) diff |= (uint8_t)(ct[i] ^ ct2[i]);
    uint8_t mask = (uint8_t)(((uint32_t)diff - 1) >> 8); /* 0xff when equal */
    for (int i = 0; i < 32; i++) kr2[i] = (uint8_t)((kr2[i] & mask) | (z[i] & ~mask));
    orc_shake256(ss, 32, kr2, 64);
    return 0;
}

/* ---- primitives exposed for the unit-level parity tests ------------------ */

void orc_kyber_ntt(int16_t p[256]) { if (!zetas_ready) zetas_init(); poly_ntt((poly *)p); }
void orc_kyber_invntt(int16_t p[256]) { if (!zetas_ready) zetas_init(); poly_invntt((poly *)p); }
void orc_kyber_normalize(int16_t p[256]) { poly_normalize((poly *)p); }
void orc_kyber_mulhat(int16_t r[256], const int16_t a[256], const int16_t b[256]) {
    if (!zetas_ready) zetas_init();
    poly_mulhat((poly *)r, (const poly *)a, (const poly *)b);
}
void orc_kyber_noise(int16_t p[256], const uint8_t seed[32], uint8_t nonce, int eta) {
    poly_noise((poly *)p, seed, nonce, eta);
}
void orc_kyber_uniform(int16_t p[256], const uint8_t seed[32], uint8_t x, uint8_t y) {
    poly_uniform((poly *)p, seed, x, y);
}
void orc_kyber_compress(uint8_t *m, const int16_t p[256], int d) { poly_compress(m, (const poly *)p, d); }
void orc_kyber_decompress(int16_t p[256], const uint8_t *m, int d) { poly_decompress((poly *)p, m, d); }
const int16_t *orc_kyber_zetas(void) { if (!zetas_ready) zetas_init(); return ZETAS; }
