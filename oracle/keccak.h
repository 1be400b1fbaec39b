/* oracle/keccak.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Scalar restatement of the reference's Keccak-f[1600] and sponge:
 *   internal/sha3/keccakf.go:12-391  (permutation, 24 or 12 rounds)
 *   internal/sha3/rc.go:4-29         (round constants)
 *   internal/sha3/sha3.go:82-185     (State.Write/Read/permute/padAndPermute)
 *   internal/sha3/shake.go:42-76, hashes.go:21-37 (rates / domain bytes)
 * Nothing under circl_amd/ may include or link this file.
 */
#ifndef ORC_KECCAK_H
#define ORC_KECCAK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

void orc_keccak_f1600(uint64_t a[25], int rounds);

typedef struct {
    uint64_t a[25];
    unsigned rate;     /* bytes */
    unsigned pos;      /* absorb: bytes xored into the current block; squeeze: bytes already read */
    uint8_t ds;        /* domain-separation byte: 0x06 (SHA3) or 0x1f (SHAKE) */
    int squeezing;
} orc_sponge;

void orc_sponge_init(orc_sponge *s, unsigned rate, uint8_t ds);
void orc_sponge_absorb(orc_sponge *s, const uint8_t *in, size_t len);
void orc_sponge_squeeze(orc_sponge *s, uint8_t *out, size_t len);

/* rate / ds pairs (shake.go:42-46, hashes.go:21-37) */
#define ORC_SHAKE128_RATE 168
#define ORC_SHAKE256_RATE 136
#define ORC_SHA3_256_RATE 136
#define ORC_SHA3_512_RATE 72
#define ORC_DS_SHAKE 0x1f
#define ORC_DS_SHA3 0x06

void orc_sha3_256(uint8_t out[32], const uint8_t *in, size_t len);
void orc_sha3_512(uint8_t out[64], const uint8_t *in, size_t len);
void orc_shake128(uint8_t *out, size_t outlen, const uint8_t *in, size_t len);
void orc_shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t len);
/* generic one-shot used by the KAT tests: rate in bytes, ds byte */
void orc_sponge_oneshot(uint8_t *out, size_t outlen, const uint8_t *in, size_t len,
                        unsigned rate, uint8_t ds);
void orc_sponge_oneshot_rounds(uint8_t *out, size_t outlen, const uint8_t *in, size_t len, unsigned rate,
                               uint8_t ds, int rounds);

#ifdef __cplusplus
}
#endif
#endif
