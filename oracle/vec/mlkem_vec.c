/* oracle/vec/mlkem_vec.c -- TEST INFRASTRUCTURE ONLY: the "vectorized" leg of bench.py's cpu_baseline.
 *
 * What the scalar oracle (oracle/kyber.c: a line-by-line restatement of the reference's GENERIC Go) cannot say is how fast a
 * CPU is at this workload when it is programmed the way the GPU is: a batch, data-parallel across items.  The reference's own
 * fast path (pke/kyber/internal/common/amd64.go:219-225 -> AVX2 assembler for NTT / InvNTT / MulHat, simd/keccakf1600
 * f1600x4 for the matrix, sample.go:101-190 DeriveX4) vectorises INSIDE one operation; no Go toolchain exists on any box this
 * repository can use, so that path cannot be timed.  This file is the CPU counterpart of the batch engine instead:
 *   - W items side by side in the 16-bit lanes of a vector (W = 16 with AVX2, 32 with AVX-512): one polynomial of W items is
 *     int16[256][W], every ring operation (ntt.go:117-193, poly.go:63-100, field.go:4-74) is a loop of full-width vector
 *     instructions with no shuffles;
 *   - Keccak-f[1600] on 4 / 8 states per vector (simd/keccakf1600's shape; AVX-512: VPROLQ and VPTERNLOGQ), the states of one
 *     batch being the same sponge of 4 / 8 different items;
 *   - byte-level stages (12-bit unpack, rejection sampling sample.go:192-236, CBD sample.go:67-95, compression poly.go:248-332
 *     through a table of the reference's multiply-shift results) per item, 8x8 transposes between the two layouts.
 * ML-KEM-768 and -1024 encapsulation (kem/mlkem/mlkem768/kyber.go:103-137 EncapsulateTo after :247-263 Unpack), bytes equal
 * to oracle/kyber.c's (tests/test_oracle_vec.py) -- the oracle stays the checker, this is only a second, faster CPU number.
 * Compiled twice (-DSUF=_avx2 / -DSUF=_avx512, oracle/Makefile); oracle/vec/dispatch.c picks at run time.
 * Nothing under circl_amd/ may include, link or call this file.
 */
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

#define KQ 3329
#define QINV 62209 /* q^-1 mod 2^16 (field.go:4-32) */

#ifdef ORCV_AVX512
#define W16 32
#define N64 8
typedef __m512i vec;
#define V_LOAD(p) _mm512_loadu_si512((const void *)(p))
#define V_STORE(p, v) _mm512_storeu_si512((void *)(p), v)
#define V_SET1_16(x) _mm512_set1_epi16((short)(x))
#define V_SET1_64(x) _mm512_set1_epi64((long long)(x))
#define V_ADD16 _mm512_add_epi16
#define V_SUB16 _mm512_sub_epi16
#define V_MULLO16 _mm512_mullo_epi16
#define V_MULHI16 _mm512_mulhi_epi16
#define V_SRAI16(a, n) _mm512_srai_epi16(a, n)
#define V_AND _mm512_and_si512
#define V_XOR _mm512_xor_si512
#define V_ROL64(a, n) _mm512_rol_epi64(a, n)
#define V_XOR3(a, b, c) _mm512_ternarylogic_epi64(a, b, c, 0x96)
#define V_CHI(a, b, c) _mm512_ternarylogic_epi64(a, b, c, 0xD2) /* a ^ (~b & c) */
#define V_ZERO() _mm512_setzero_si512()
#else
#define W16 16
#define N64 4
typedef __m256i vec;
#define V_LOAD(p) _mm256_loadu_si256((const __m256i *)(p))
#define V_STORE(p, v) _mm256_storeu_si256((__m256i *)(p), v)
#define V_SET1_16(x) _mm256_set1_epi16((short)(x))
#define V_SET1_64(x) _mm256_set1_epi64x((long long)(x))
#define V_ADD16 _mm256_add_epi16
#define V_SUB16 _mm256_sub_epi16
#define V_MULLO16 _mm256_mullo_epi16
#define V_MULHI16 _mm256_mulhi_epi16
#define V_SRAI16(a, n) _mm256_srai_epi16(a, n)
#define V_AND _mm256_and_si256
#define V_XOR _mm256_xor_si256
#define V_ROL64(a, n) _mm256_or_si256(_mm256_slli_epi64(a, n), _mm256_srli_epi64(a, 64 - (n)))
#define V_XOR3(a, b, c) _mm256_xor_si256(_mm256_xor_si256(a, b), c)
#define V_CHI(a, b, c) _mm256_xor_si256(a, _mm256_andnot_si256(b, c))
#define V_ZERO() _mm256_setzero_si256()
#endif

/* ---- Keccak-f[1600] on N64 states (internal/sha3/keccakf.go:12-391; the lane-interleaved layout of simd/keccakf1600) ---- */
static const uint64_t RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
    0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
    0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

/* st: 25 words x N64 states, word-major (state l's word w at st[w * N64 + l]).  A round writes a second set of 25 words row by row
 * (theta's column words, then per output row: five rho-pi moves and chi), so that an input word dies where an output word is born:
 * 25 + 5 + 5 vectors live instead of 60 -- with 32 registers that is a handful of spills per round instead of several dozen. */
#define ROL0(a, n) ((n) == 0 ? (a) : V_ROL64(a, (n) == 0 ? 1 : (n)))
#define ROW(A, E, y, i0, i1, i2, i3, i4, r0, r1, r2, r3, r4)                                                                     \
    do {                                                                                                                         \
        const vec b0 = ROL0(V_XOR(A[i0], d[(i0) % 5]), r0), b1 = ROL0(V_XOR(A[i1], d[(i1) % 5]), r1), b2 = ROL0(V_XOR(A[i2], d[(i2) % 5]), r2), \
                  b3 = ROL0(V_XOR(A[i3], d[(i3) % 5]), r3), b4 = ROL0(V_XOR(A[i4], d[(i4) % 5]), r4);                           \
        E[5 * (y)] = V_CHI(b0, b1, b2);                                                                                          \
        E[5 * (y) + 1] = V_CHI(b1, b2, b3);                                                                                      \
        E[5 * (y) + 2] = V_CHI(b2, b3, b4);                                                                                      \
        E[5 * (y) + 3] = V_CHI(b3, b4, b0);                                                                                      \
        E[5 * (y) + 4] = V_CHI(b4, b0, b1);                                                                                      \
    } while (0)
#define ROUND(A, E, rc)                                                                                                          \
    do {                                                                                                                         \
        vec c[5], d[5];                                                                                                          \
        c[0] = V_XOR3(V_XOR3(A[0], A[5], A[10]), A[15], A[20]);                                                                  \
        c[1] = V_XOR3(V_XOR3(A[1], A[6], A[11]), A[16], A[21]);                                                                  \
        c[2] = V_XOR3(V_XOR3(A[2], A[7], A[12]), A[17], A[22]);                                                                  \
        c[3] = V_XOR3(V_XOR3(A[3], A[8], A[13]), A[18], A[23]);                                                                  \
        c[4] = V_XOR3(V_XOR3(A[4], A[9], A[14]), A[19], A[24]);                                                                  \
        d[0] = V_XOR(c[4], V_ROL64(c[1], 1));                                                                                    \
        d[1] = V_XOR(c[0], V_ROL64(c[2], 1));                                                                                    \
        d[2] = V_XOR(c[1], V_ROL64(c[3], 1));                                                                                    \
        d[3] = V_XOR(c[2], V_ROL64(c[4], 1));                                                                                    \
        d[4] = V_XOR(c[3], V_ROL64(c[0], 1));                                                                                    \
        ROW(A, E, 0, 0, 6, 12, 18, 24, 0, 44, 43, 21, 14);                                                                       \
        E[0] = V_XOR(E[0], V_SET1_64(rc));                                                                                       \
        ROW(A, E, 1, 3, 9, 10, 16, 22, 28, 20, 3, 45, 61);                                                                       \
        ROW(A, E, 2, 1, 7, 13, 19, 20, 1, 6, 25, 8, 18);                                                                         \
        ROW(A, E, 3, 4, 5, 11, 17, 23, 27, 36, 10, 15, 56);                                                                      \
        ROW(A, E, 4, 2, 8, 14, 15, 21, 62, 55, 39, 41, 2);                                                                       \
    } while (0)
static void FN(f1600)(uint64_t *st) {
    vec s[25], e[25];
    for (int i = 0; i < 25; i++) s[i] = V_LOAD(st + i * N64);
    for (int r = 0; r < 24; r += 2) {
        ROUND(s, e, RC[r]);
        ROUND(e, s, RC[r + 1]);
    }
    for (int i = 0; i < 25; i++) V_STORE(st + i * N64, s[i]);
}

static inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* ---- field.go / ntt.go / poly.go on int16[256][W16] ---- */
typedef struct { vec c[256]; } spoly;

static int16_t ZETAS[128], ZETAS_QINV[128];
static uint16_t LUT10[KQ], LUT11[KQ];
static uint8_t LUT4[KQ], LUT5[KQ];

/* once per process and instruction set (oracle/vec/dispatch.c: pthread_once) */
static void FN(tables)(void) {
    for (int i = 0; i < 128; i++) { /* ntt.go:16-28: Zetas[i] = 17^brv7(i) * 2^16 mod q */
        int brv = 0;
        for (int b = 0; b < 7; b++) brv |= ((i >> b) & 1) << (6 - b);
        uint32_t z = 1;
        for (int e = 0; e < brv; e++) z = z * 17 % KQ;
        ZETAS[i] = (int16_t)((z << 16) % KQ);
        ZETAS_QINV[i] = (int16_t)(uint16_t)((uint32_t)(uint16_t)ZETAS[i] * QINV);
    }
    for (uint32_t x = 0; x < KQ; x++) { /* poly.go:248-332 CompressTo's multiply-shift constants, tabulated over [0, q) */
        LUT10[x] = (uint16_t)((((uint64_t)((x << 10) + KQ / 2) * 20642679ull) >> 36) & 1023);
        LUT11[x] = (uint16_t)((((uint64_t)((x << 11) + KQ / 2) * 20642679ull) >> 36) & 2047);
        LUT4[x] = (uint8_t)(((((x << 4) + KQ / 2) * 315) >> 20) & 15);
        LUT5[x] = (uint8_t)(((((x << 5) + KQ / 2) * 315) >> 20) & 31);
    }
}

/* field.go:4-32 montReduce(a * b) for a constant b given as (b, b * q^-1): hi16(a b) - hi16(lo16(a b q^-1) q), the same value */
static inline vec fqmul_c(vec a, vec z, vec zq) {
    const vec q = V_SET1_16(KQ);
    return V_SUB16(V_MULHI16(a, z), V_MULHI16(V_MULLO16(a, zq), q));
}
static inline vec fqmul_v(vec a, vec b) {
    const vec q = V_SET1_16(KQ), qi = V_SET1_16(QINV);
    return V_SUB16(V_MULHI16(a, b), V_MULHI16(V_MULLO16(V_MULLO16(a, b), qi), q));
}
/* field.go:45-64 barrettReduce: x - ((x * 20159) >> 26) q, in [0, q] */
static inline vec barrett(vec x) {
    return V_SUB16(x, V_MULLO16(V_SRAI16(V_MULHI16(x, V_SET1_16(20159)), 10), V_SET1_16(KQ)));
}
/* field.go:67-74 csubq */
static inline vec csubq(vec x) {
    x = V_SUB16(x, V_SET1_16(KQ));
    return V_ADD16(x, V_AND(V_SRAI16(x, 15), V_SET1_16(KQ)));
}

static void sp_ntt(spoly *p) { /* ntt.go:117-134 */
    int k = 0;
    for (int l = 128; l > 1; l >>= 1)
        for (int off = 0; off < 256 - l; off += 2 * l) {
            k++;
            const vec z = V_SET1_16(ZETAS[k]), zq = V_SET1_16(ZETAS_QINV[k]);
            for (int j = off; j < off + l; j++) {
                const vec t = fqmul_c(p->c[j + l], z, zq);
                p->c[j + l] = V_SUB16(p->c[j], t);
                p->c[j] = V_ADD16(p->c[j], t);
            }
        }
}
static void sp_invntt(spoly *p) { /* ntt.go:145-193; the sums are Barrett-reduced after every layer (congruent, never overflows) */
    int k = 127;
    for (int l = 2; l < 256; l <<= 1)
        for (int off = 0; off < 256 - l; off += 2 * l) {
            const vec z = V_SET1_16(ZETAS[k]), zq = V_SET1_16(ZETAS_QINV[k]);
            k--;
            for (int j = off; j < off + l; j++) {
                const vec t = V_SUB16(p->c[j + l], p->c[j]);
                p->c[j] = barrett(V_ADD16(p->c[j], p->c[j + l]));
                p->c[j + l] = fqmul_c(t, z, zq);
            }
        }
    const vec f = V_SET1_16(1441), fq = V_SET1_16((int16_t)(uint16_t)(1441u * QINV));
    for (int j = 0; j < 256; j++) p->c[j] = fqmul_c(p->c[j], f, fq);
}
/* poly.go:63-100 mulHat, accumulated: acc += a o b (acc = a o b when first) */
static void sp_mulhat_acc(spoly *acc, const spoly *a, const spoly *b, int first) {
    for (int i = 0; i < 256; i += 4) {
        const vec z = V_SET1_16(ZETAS[64 + i / 4]), zq = V_SET1_16(ZETAS_QINV[64 + i / 4]);
        const vec a0 = a->c[i], a1 = a->c[i + 1], a2 = a->c[i + 2], a3 = a->c[i + 3];
        const vec b0 = b->c[i], b1 = b->c[i + 1], b2 = b->c[i + 2], b3 = b->c[i + 3];
        vec p0 = V_ADD16(fqmul_c(fqmul_v(a1, b1), z, zq), fqmul_v(a0, b0));
        vec p1 = V_ADD16(fqmul_v(a0, b1), fqmul_v(a1, b0));
        vec p2 = V_SUB16(fqmul_v(a2, b2), fqmul_c(fqmul_v(a3, b3), z, zq));
        vec p3 = V_ADD16(fqmul_v(a2, b3), fqmul_v(a3, b2));
        if (!first) {
            p0 = V_ADD16(p0, acc->c[i]);
            p1 = V_ADD16(p1, acc->c[i + 1]);
            p2 = V_ADD16(p2, acc->c[i + 2]);
            p3 = V_ADD16(p3, acc->c[i + 3]);
        }
        acc->c[i] = p0; acc->c[i + 1] = p1; acc->c[i + 2] = p2; acc->c[i + 3] = p3;
    }
}

/* ---- 8x8 transposes between int16[W16][256] (one item's polynomial contiguous) and int16[256][W16] ---- */
static inline void tr8x8(const int16_t *src, size_t ss, int16_t *dst, size_t ds) {
    __m128i r0 = _mm_loadu_si128((const __m128i *)(src + 0 * ss)), r1 = _mm_loadu_si128((const __m128i *)(src + 1 * ss));
    __m128i r2 = _mm_loadu_si128((const __m128i *)(src + 2 * ss)), r3 = _mm_loadu_si128((const __m128i *)(src + 3 * ss));
    __m128i r4 = _mm_loadu_si128((const __m128i *)(src + 4 * ss)), r5 = _mm_loadu_si128((const __m128i *)(src + 5 * ss));
    __m128i r6 = _mm_loadu_si128((const __m128i *)(src + 6 * ss)), r7 = _mm_loadu_si128((const __m128i *)(src + 7 * ss));
    __m128i a0 = _mm_unpacklo_epi16(r0, r1), a1 = _mm_unpackhi_epi16(r0, r1), a2 = _mm_unpacklo_epi16(r2, r3), a3 = _mm_unpackhi_epi16(r2, r3);
    __m128i a4 = _mm_unpacklo_epi16(r4, r5), a5 = _mm_unpackhi_epi16(r4, r5), a6 = _mm_unpacklo_epi16(r6, r7), a7 = _mm_unpackhi_epi16(r6, r7);
    __m128i b0 = _mm_unpacklo_epi32(a0, a2), b1 = _mm_unpackhi_epi32(a0, a2), b2 = _mm_unpacklo_epi32(a1, a3), b3 = _mm_unpackhi_epi32(a1, a3);
    __m128i b4 = _mm_unpacklo_epi32(a4, a6), b5 = _mm_unpackhi_epi32(a4, a6), b6 = _mm_unpacklo_epi32(a5, a7), b7 = _mm_unpackhi_epi32(a5, a7);
    _mm_storeu_si128((__m128i *)(dst + 0 * ds), _mm_unpacklo_epi64(b0, b4));
    _mm_storeu_si128((__m128i *)(dst + 1 * ds), _mm_unpackhi_epi64(b0, b4));
    _mm_storeu_si128((__m128i *)(dst + 2 * ds), _mm_unpacklo_epi64(b1, b5));
    _mm_storeu_si128((__m128i *)(dst + 3 * ds), _mm_unpackhi_epi64(b1, b5));
    _mm_storeu_si128((__m128i *)(dst + 4 * ds), _mm_unpacklo_epi64(b2, b6));
    _mm_storeu_si128((__m128i *)(dst + 5 * ds), _mm_unpackhi_epi64(b2, b6));
    _mm_storeu_si128((__m128i *)(dst + 6 * ds), _mm_unpacklo_epi64(b3, b7));
    _mm_storeu_si128((__m128i *)(dst + 7 * ds), _mm_unpackhi_epi64(b3, b7));
}
static void to_soa(spoly *dst, const int16_t *aos /* [W16][256] */) {
    int16_t *d = (int16_t *)dst;
    for (int ib = 0; ib < W16; ib += 8)
        for (int cb = 0; cb < 256; cb += 8) tr8x8(aos + ib * 256 + cb, 256, d + cb * W16 + ib, W16);
}
static void to_aos(int16_t *aos, const spoly *src) {
    const int16_t *s = (const int16_t *)src;
    for (int ib = 0; ib < W16; ib += 8)
        for (int cb = 0; cb < 256; cb += 8) tr8x8(s + cb * W16 + ib, W16, aos + ib * 256 + cb, 256);
}

/* ---- per-item byte stages ---- */
/* poly.go:123-129 Unpack; returns non-zero if a coefficient is >= q (cpapke.go:45-55 UnpackMLKEM's re-pack-and-compare) */
static inline unsigned unpack12(int16_t *c, const uint8_t *b) {
    unsigned bad = 0;
    for (int i = 0; i < 128; i++) {
        const unsigned t0 = b[3 * i] | ((unsigned)(b[3 * i + 1] & 0xf) << 8), t1 = (b[3 * i + 1] >> 4) | ((unsigned)b[3 * i + 2] << 4);
        c[2 * i] = (int16_t)t0;
        c[2 * i + 1] = (int16_t)t1;
        bad |= (t0 >= KQ) | (t1 >= KQ);
    }
    return bad;
}
/* sample.go:192-236: one SHAKE128 block (168 bytes in a buffer of 192), candidates in stream order (t1 then t2 of every 3 bytes);
 * dst has room for 255 + 128 entries */
#ifdef ORCV_AVX512
/* 32 candidates per step: VPERMB brings the two bytes of candidate k (they start at byte 3k / 2) into 16-bit lane k, odd lanes shift
 * by 4, the accepted ones are packed to the front by VPCOMPRESSW (register form) and stored */
static inline unsigned rej_block(int16_t *dst, unsigned ctr, const uint8_t *buf) {
    static const uint8_t IDX[64] __attribute__((aligned(64))) = {
        0, 1, 1, 2, 3, 4, 4, 5, 6, 7, 7, 8, 9, 10, 10, 11, 12, 13, 13, 14, 15, 16, 16, 17, 18, 19, 19, 20, 21, 22, 22, 23,
        24, 25, 25, 26, 27, 28, 28, 29, 30, 31, 31, 32, 33, 34, 34, 35, 36, 37, 37, 38, 39, 40, 40, 41, 42, 43, 43, 44, 45, 46, 46, 47};
    const __m512i idx = _mm512_load_si512((const void *)IDX), sh = _mm512_set1_epi32(0x00040000), mk = _mm512_set1_epi16(0xfff), q = _mm512_set1_epi16(KQ);
    for (int j = 0; j < 4; j++) { /* 48 + 48 + 48 + 24 bytes */
        __m512i v = _mm512_permutexvar_epi8(idx, _mm512_loadu_si512((const void *)(buf + 48 * j)));
        v = _mm512_and_si512(_mm512_srlv_epi16(v, sh), mk);
        const __mmask32 ok = _mm512_cmplt_epu16_mask(v, q) & (j == 3 ? 0x0000ffffu : 0xffffffffu);
        _mm512_storeu_si512((void *)(dst + ctr), _mm512_maskz_compress_epi16(ok, v));
        ctr += (unsigned)__builtin_popcount(ok);
    }
    return ctr;
}
#else
static inline unsigned rej_block(int16_t *dst, unsigned ctr, const uint8_t *buf) {
    for (int j = 0; j < 168; j += 6) { /* four candidates per 48-bit load */
        const uint64_t t = ld64(buf + j);
        const unsigned c0 = (unsigned)t & 0xfff, c1 = (unsigned)(t >> 12) & 0xfff, c2 = (unsigned)(t >> 24) & 0xfff, c3 = (unsigned)(t >> 36) & 0xfff;
        dst[ctr] = (int16_t)c0; ctr += c0 < KQ;
        dst[ctr] = (int16_t)c1; ctr += c1 < KQ;
        dst[ctr] = (int16_t)c2; ctr += c2 < KQ;
        dst[ctr] = (int16_t)c3; ctr += c3 < KQ;
    }
    return ctr;
}
#endif
/* sample.go:67-95 DeriveNoise2's CBD on 128 bytes held as 16 words of one interleaved state: per nibble (a in bits 0-1, b in bits 2-3
 * after the pairwise bit sums) the coefficient a - b; 16 nibbles -> 16 coefficients per step */
static inline void cbd2(int16_t *c, const uint64_t *st, int lane) {
    const __m128i m3 = _mm_set1_epi16(3), mf = _mm_set1_epi16(0xf);
    for (int w = 0; w < 16; w++) {
        const uint64_t t = st[w * N64 + lane];
        const uint64_t d = (t & 0x5555555555555555ull) + ((t >> 1) & 0x5555555555555555ull);
        const __m128i v = _mm_cvtepu8_epi16(_mm_cvtsi64_si128((long long)d)); /* byte k of d in lane k */
        const __m128i lo = _mm_and_si128(v, mf), hi = _mm_srli_epi16(v, 4);
        const __m128i rl = _mm_sub_epi16(_mm_and_si128(lo, m3), _mm_srli_epi16(lo, 2)), rh = _mm_sub_epi16(_mm_and_si128(hi, m3), _mm_srli_epi16(hi, 2));
        _mm_storeu_si128((__m128i *)(c + 16 * w), _mm_unpacklo_epi16(rl, rh));
        _mm_storeu_si128((__m128i *)(c + 16 * w + 8), _mm_unpackhi_epi16(rl, rh));
    }
}
/* poly.go:134-145 DecompressMessage */
static inline void from_msg(int16_t *c, const uint8_t *m) {
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 8; j++) c[8 * i + j] = (int16_t)(-((m[i] >> j) & 1) & ((KQ + 1) / 2));
}
/* poly.go:248-332 CompressTo of a normalised polynomial: d bits per coefficient, least-significant bit first; eight coefficients
 * (sixteen for d = 4) per step */
static inline void pack10(uint8_t *o, const int16_t *c) {
    for (int i = 0; i < 256; i += 8, o += 10) {
        const uint64_t t6 = LUT10[(uint16_t)c[i + 6]];
        const uint64_t lo = (uint64_t)LUT10[(uint16_t)c[i]] | (uint64_t)LUT10[(uint16_t)c[i + 1]] << 10 | (uint64_t)LUT10[(uint16_t)c[i + 2]] << 20 |
                            (uint64_t)LUT10[(uint16_t)c[i + 3]] << 30 | (uint64_t)LUT10[(uint16_t)c[i + 4]] << 40 | (uint64_t)LUT10[(uint16_t)c[i + 5]] << 50 | t6 << 60;
        const uint16_t hi = (uint16_t)(t6 >> 4 | (uint64_t)LUT10[(uint16_t)c[i + 7]] << 6);
        memcpy(o, &lo, 8);
        memcpy(o + 8, &hi, 2);
    }
}
static inline void pack11(uint8_t *o, const int16_t *c) {
    for (int i = 0; i < 256; i += 8, o += 11) {
        const uint64_t t5 = LUT11[(uint16_t)c[i + 5]];
        const uint64_t lo = (uint64_t)LUT11[(uint16_t)c[i]] | (uint64_t)LUT11[(uint16_t)c[i + 1]] << 11 | (uint64_t)LUT11[(uint16_t)c[i + 2]] << 22 |
                            (uint64_t)LUT11[(uint16_t)c[i + 3]] << 33 | (uint64_t)LUT11[(uint16_t)c[i + 4]] << 44 | t5 << 55;
        const uint32_t hi = (uint32_t)(t5 >> 9 | (uint64_t)LUT11[(uint16_t)c[i + 6]] << 2 | (uint64_t)LUT11[(uint16_t)c[i + 7]] << 13);
        memcpy(o, &lo, 8);
        memcpy(o + 8, &hi, 3);
    }
}
static inline void pack4(uint8_t *o, const int16_t *c) {
    for (int i = 0; i < 256; i += 16, o += 8) {
        uint64_t w = 0;
        for (int j = 0; j < 16; j++) w |= (uint64_t)LUT4[(uint16_t)c[i + j]] << (4 * j);
        memcpy(o, &w, 8);
    }
}
static inline void pack5(uint8_t *o, const int16_t *c) {
    for (int i = 0; i < 256; i += 8, o += 5) {
        uint64_t w = 0;
        for (int j = 0; j < 8; j++) w |= (uint64_t)LUT5[(uint16_t)c[i + j]] << (5 * j);
        memcpy(o, &w, 5);
    }
}

typedef struct {
    spoly at[16], th[4], rh[4], e1[4], u[4], e2, m, v;
    int16_t aos[W16][256];
    int16_t rej[N64][416];
    uint64_t st[25 * N64];
    uint8_t kr[W16][64]; /* K-bar || r per item (G = SHA3-512: kyber.go:118-121) */
    uint64_t h[W16][4];  /* H(ek) per item */
    uint8_t bad[W16];
    int have_key;        /* one key for the whole batch (ekstride 0): th, aT, H(ek) and the verdict are those of the first group */
} scratch;

/* W16 encapsulations: item of lane l is it[l] (lanes >= cnt repeat it[0] and write nothing).  ekstride: bytes between the items' keys, or 0 =
 * ONE key for the batch -- the parsed-key shape of the reference's BenchmarkEncapsulate (kem/schemes/schemes_test.go:28-38; kyber.go:39-43
 * caches H(ek), cpapke.go:19-25 th and aT): the key's stages run with the first group only */
static void FN(group)(scratch *S, int K, int du, int dv, const uint8_t *ek, size_t ekstride, const uint8_t *m, uint8_t *ct, uint8_t *ss, uint8_t *status,
                      const size_t *it, int cnt) {
    const size_t eksz = (size_t)(384 * K + 32), ctsz = (size_t)(32 * (du * K + dv));
    uint64_t *st = S->st;
    const int skip = ekstride == 0 && S->have_key;
    /* kyber.go:247-263 Unpack: t-hat (12-bit), the >= q check */
    if (!skip) {
        memset(S->bad, 0, sizeof S->bad);
        for (int j = 0; j < K; j++) {
            for (int l = 0; l < W16; l++) S->bad[l] |= (uint8_t)unpack12(S->aos[l], ek + ekstride * it[l] + 384 * j);
            to_soa(&S->th[j], &S->aos[0][0]);
        }
    }
    /* H(ek) = SHA3-256 (kyber.go:39-43), then (K-bar, r) = G(m || H(ek)) = SHA3-512 (kyber.go:118-121); N64 items per permutation */
    for (int b0 = 0; b0 < W16; b0 += N64) {
        if (skip) goto have_h;
        memset(st, 0, sizeof S->st);
        size_t off = 0;
        for (; off + 136 <= eksz; off += 136) {
            for (int l = 0; l < N64; l++) {
                const uint8_t *p = ek + ekstride * it[b0 + l] + off;
                for (int w = 0; w < 17; w++) st[w * N64 + l] ^= ld64(p + 8 * w);
            }
            FN(f1600)(st);
        }
        const int remw = (int)((eksz - off) / 8); /* 96 (K = 3) / 72 (K = 4) bytes: whole words */
        for (int l = 0; l < N64; l++) {
            const uint8_t *p = ek + ekstride * it[b0 + l] + off;
            for (int w = 0; w < remw; w++) st[w * N64 + l] ^= ld64(p + 8 * w);
            st[remw * N64 + l] ^= 0x06;
            st[16 * N64 + l] ^= 0x8000000000000000ull;
        }
        FN(f1600)(st);
        for (int l = 0; l < N64; l++)
            for (int w = 0; w < 4; w++) S->h[b0 + l][w] = st[w * N64 + l];
    have_h:
        memset(st, 0, sizeof S->st);
        for (int l = 0; l < N64; l++) {
            const uint8_t *p = m + 32 * it[b0 + l];
            for (int w = 0; w < 4; w++) {
                st[w * N64 + l] = ld64(p + 8 * w);
                st[(4 + w) * N64 + l] = S->h[b0 + l][w];
            }
            st[8 * N64 + l] = 0x8000000000000006ull; /* 0x06 at byte 64, 0x80 at byte 71 (rate 72) */
        }
        FN(f1600)(st);
        for (int l = 0; l < N64; l++)
            for (int w = 0; w < 8; w++) memcpy(&S->kr[b0 + l][8 * w], &st[w * N64 + l], 8);
    }
    /* mat.go:13-74 Derive(rho, transpose = true): aT[i][j] = SHAKE128(rho || i || j), rejection-sampled */
    for (int i = 0; i < K && !skip; i++)
        for (int j = 0; j < K; j++) {
            for (int b0 = 0; b0 < W16; b0 += N64) {
                memset(st, 0, sizeof S->st);
                unsigned ctr[N64];
                for (int l = 0; l < N64; l++) {
                    const uint8_t *rho = ek + ekstride * it[b0 + l] + 384 * K;
                    for (int w = 0; w < 4; w++) st[w * N64 + l] = ld64(rho + 8 * w);
                    st[4 * N64 + l] = (uint64_t)i | ((uint64_t)j << 8) | (0x1full << 16);
                    st[20 * N64 + l] = 0x8000000000000000ull;
                    ctr[l] = 0;
                }
                for (;;) {
                    FN(f1600)(st);
                    int open = 0;
                    for (int l = 0; l < N64; l++) {
                        if (ctr[l] >= 256) continue;
                        uint64_t blk[24];
                        for (int w = 0; w < 21; w++) blk[w] = st[w * N64 + l];
                        blk[21] = blk[22] = blk[23] = 0;
                        ctr[l] = rej_block(S->rej[l], ctr[l], (const uint8_t *)blk);
                        open |= ctr[l] < 256;
                    }
                    if (!open) break;
                }
                for (int l = 0; l < N64; l++) memcpy(S->aos[b0 + l], S->rej[l], 512);
            }
            to_soa(&S->at[i * K + j], &S->aos[0][0]);
        }
    if (ekstride == 0) S->have_key = 1;
    /* cpapke.go:137-181 EncryptTo: r-hat, e1, e2 = CBD(PRF(r, nonce)) with nonces 0 .. 2K (eta1 = eta2 = 2) */
    for (int nonce = 0; nonce <= 2 * K; nonce++) {
        for (int b0 = 0; b0 < W16; b0 += N64) {
            memset(st, 0, sizeof S->st);
            for (int l = 0; l < N64; l++) {
                for (int w = 0; w < 4; w++) st[w * N64 + l] = ld64(&S->kr[b0 + l][32 + 8 * w]);
                st[4 * N64 + l] = (uint64_t)nonce | (0x1full << 8);
                st[16 * N64 + l] = 0x8000000000000000ull;
            }
            FN(f1600)(st);
            for (int l = 0; l < N64; l++) cbd2(S->aos[b0 + l], st, l);
        }
        to_soa(nonce < K ? &S->rh[nonce] : nonce < 2 * K ? &S->e1[nonce - K] : &S->e2, &S->aos[0][0]);
    }
    for (int l = 0; l < W16; l++) from_msg(S->aos[l], m + 32 * it[l]);
    to_soa(&S->m, &S->aos[0][0]);
    /* the ring phase, W16 items per instruction */
    for (int i = 0; i < K; i++) {
        sp_ntt(&S->rh[i]);
        for (int c = 0; c < 256; c++) S->rh[i].c[c] = barrett(S->rh[i].c[c]);
    }
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < K; j++) sp_mulhat_acc(&S->u[i], &S->at[i * K + j], &S->rh[j], j == 0);
        for (int c = 0; c < 256; c++) S->u[i].c[c] = barrett(S->u[i].c[c]);
        sp_invntt(&S->u[i]);
        for (int c = 0; c < 256; c++) S->u[i].c[c] = csubq(barrett(V_ADD16(S->u[i].c[c], S->e1[i].c[c])));
    }
    for (int j = 0; j < K; j++) sp_mulhat_acc(&S->v, &S->th[j], &S->rh[j], j == 0);
    for (int c = 0; c < 256; c++) S->v.c[c] = barrett(S->v.c[c]);
    sp_invntt(&S->v);
    for (int c = 0; c < 256; c++) S->v.c[c] = csubq(barrett(V_ADD16(V_ADD16(S->v.c[c], S->m.c[c]), S->e2.c[c])));
    /* compress + pack (poly.go:248-332), K-bar out */
    for (int i = 0; i <= K; i++) {
        to_aos(&S->aos[0][0], i < K ? &S->u[i] : &S->v);
        for (int l = 0; l < cnt; l++) {
            uint8_t *o = ct + ctsz * it[l] + (size_t)(32 * du * i);
            if (i < K) (du == 10 ? pack10 : pack11)(o, S->aos[l]);
            else (dv == 4 ? pack4 : pack5)(o, S->aos[l]);
        }
    }
    for (int l = 0; l < cnt; l++) {
        if (S->bad[l]) { /* kem.ErrPubKey: nothing but the status */
            memset(ct + ctsz * it[l], 0, ctsz);
            memset(ss + 32 * it[l], 0, 32);
        } else {
            memcpy(ss + 32 * it[l], S->kr[l], 32);
        }
        if (status) status[it[l]] = S->bad[l] ? 1 : 0;
    }
}

/* items [lo, hi) of a batch; returns 0, -1 (parameter set), -2 (memory) */
int FN(orcv_mlkem_encaps)(int param, const uint8_t *ek, int shared, const uint8_t *m, uint8_t *ct, uint8_t *ss, uint8_t *status, size_t lo, size_t hi) {
    int K, du, dv;
    if (param == 768) { K = 3; du = 10; dv = 4; }
    else if (param == 1024) { K = 4; du = 11; dv = 5; }
    else return -1; /* ML-KEM-512 has eta1 = 3: not needed by any BASELINE config */
    scratch *S = 0;
    if (posix_memalign((void **)&S, 64, sizeof *S)) return -2;
    S->have_key = 0;
    for (size_t g = lo; g < hi; g += W16) {
        size_t it[W16];
        const int cnt = (int)(hi - g < W16 ? hi - g : W16);
        for (int l = 0; l < W16; l++) it[l] = l < cnt ? g + (size_t)l : g;
        FN(group)(S, K, du, dv, ek, shared ? 0 : (size_t)(384 * K + 32), m, ct, ss, status, it, cnt);
    }
    free(S);
    return 0;
}
void FN(orcv_tables)(void) { FN(tables)(); }
int FN(orcv_width)(void) { return W16; }
/* the permutation alone, for the test that pins it: 25 x N64 words, word-major */
void FN(orcv_f1600)(uint64_t *st) { FN(f1600)(st); }
int FN(orcv_states)(void) { return N64; }
