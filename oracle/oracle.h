/* oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU oracle for the batch ML-KEM / ML-DSA hot path: a scalar C restatement of
 * the reference's generic Go (cloudflare/circl), pinned by the reference's own
 * NIST ACVP / KAT / Wycheproof vectors (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker or as the reported CPU baseline.  The
 * product (circl_amd/) never links, imports or falls back to it.
 */
#ifndef ORC_ORACLE_H
#define ORC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-item status values (mirroring kem.ErrPubKey / kem.ErrPrivKey) */
#define ORC_OK 0
#define ORC_ERR_PUBKEY 1
#define ORC_ERR_PRIVKEY 2

/* ---- ML-KEM (param = 512 | 768 | 1024) ---- */
size_t orc_mlkem_ek_size(int param);
size_t orc_mlkem_dk_size(int param);
size_t orc_mlkem_ct_size(int param);
int orc_mlkem_keygen(int param, const uint8_t seed[64], uint8_t *ek, uint8_t *dk);
int orc_mlkem_encaps(int param, const uint8_t *ek, const uint8_t m[32], uint8_t *ct, uint8_t ss[32]);
int orc_mlkem_decaps(int param, const uint8_t *dk, const uint8_t *ct, uint8_t ss[32]);
int orc_mlkem_encaps_cached(int param, const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss, size_t n);
int orc_mlkem_encaps_shared_batch(int param, const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss, size_t n, int threads);

/* batch forms, `threads` pthreads over contiguous slices; status[n] per item.
 * On a per-item error the item's outputs are zero-filled. */
/* round-3 Kyber (kem/kyber/kyber{512,768,1024}): same sizes as ML-KEM, different hashing; never fails */
int orc_kyber_r3_keygen(int param, const uint8_t seed[64], uint8_t *ek, uint8_t *dk);
int orc_kyber_r3_encaps(int param, const uint8_t *ek, const uint8_t seed[32], uint8_t *ct, uint8_t ss[32]);
int orc_kyber_r3_decaps(int param, const uint8_t *dk, const uint8_t *ct, uint8_t ss[32]);
int orc_kyber_r3_keygen_batch(int param, const uint8_t *seed, uint8_t *ek, uint8_t *dk, size_t n, int threads);
int orc_kyber_r3_encaps_batch(int param, const uint8_t *ek, const uint8_t *seed, uint8_t *ct, uint8_t *ss, size_t n, int threads);
int orc_kyber_r3_decaps_batch(int param, const uint8_t *dk, const uint8_t *ct, uint8_t *ss, size_t n, int threads);
int orc_mlkem_keygen_batch(int param, const uint8_t *seed, uint8_t *ek, uint8_t *dk, size_t n, int threads);
int orc_mlkem_encaps_batch(int param, const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss,
                           uint8_t *status, size_t n, int threads);
int orc_mlkem_decaps_batch(int param, const uint8_t *dk, const uint8_t *ct, uint8_t *ss,
                           uint8_t *status, size_t n, int threads);

/* ---- X25519 (dh/x25519), the DH half of the hybrid KEMs; see x25519.c ---- */
int orc_x25519_shared(uint8_t shared[32], const uint8_t secret[32], const uint8_t public_[32]); /* 1 = valid public key */
void orc_x25519_keygen(uint8_t public_[32], const uint8_t secret[32]);
int orc_x25519_batch(const uint8_t *scalar, const uint8_t *point /* NULL = base point */, uint8_t *out, uint8_t *ok, size_t n);

/* Kyber ring primitives */
void orc_kyber_ntt(int16_t p[256]);
void orc_kyber_invntt(int16_t p[256]);
void orc_kyber_normalize(int16_t p[256]);
void orc_kyber_mulhat(int16_t r[256], const int16_t a[256], const int16_t b[256]);
void orc_kyber_noise(int16_t p[256], const uint8_t seed[32], uint8_t nonce, int eta);
void orc_kyber_uniform(int16_t p[256], const uint8_t seed[32], uint8_t x, uint8_t y);
void orc_kyber_compress(uint8_t *m, const int16_t p[256], int d);
void orc_kyber_decompress(int16_t p[256], const uint8_t *m, int d);
const int16_t *orc_kyber_zetas(void);

/* ---- ML-DSA (param = 44 | 65 | 87) ---- */
size_t orc_mldsa_pk_size(int param);
size_t orc_mldsa_sk_size(int param);
size_t orc_mldsa_sig_size(int param);
int orc_mldsa_keygen(int param, const uint8_t seed[32], uint8_t *pk, uint8_t *sk);
/* internal = 1: ML-DSA.Sign_internal / Verify_internal (message hashed bare, as the
 * ACVP vectors use); internal = 0: M' = 0 || len(ctx) || ctx || msg. */
int orc_mldsa_sign(int param, const uint8_t *sk, const uint8_t *msg, size_t msglen,
                   const uint8_t *ctx, size_t ctxlen, const uint8_t rnd[32], int internal, uint8_t *sig);
int orc_mldsa_verify(int param, const uint8_t *pk, const uint8_t *msg, size_t msglen,
                     const uint8_t *ctx, size_t ctxlen, int internal, const uint8_t *sig, size_t siglen);
int orc_mldsa_keygen_batch(int param, const uint8_t *seed, uint8_t *pk, uint8_t *sk, size_t n, int threads);
int orc_mldsa_sign_batch(int param, const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off,
                         const uint8_t *ctx_blob, const uint64_t *ctx_off, const uint8_t *rnd,
                         uint8_t *sig, size_t n, int threads);
int orc_mldsa_verify_batch(int param, const uint8_t *pk, const uint8_t *sig, const uint8_t *msg_blob,
                           const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off,
                           uint8_t *ok, size_t n, int threads);

/* Dilithium ring primitives */
void orc_dilithium_ntt(uint32_t p[256]);
void orc_dilithium_invntt(uint32_t p[256]);
void orc_dilithium_normalize(uint32_t p[256]);
void orc_dilithium_uniform(uint32_t p[256], const uint8_t seed[32], uint16_t nonce);
int orc_dilithium_ball(int param, uint32_t p[256], const uint8_t *ctilde);
const uint32_t *orc_dilithium_zetas(void);

#ifdef __cplusplus
}
#endif
#endif
