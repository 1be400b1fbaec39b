/* examples/resident_keys.c -- parsed key objects from plain C: a TLS-front-end shape.  One static ML-KEM-768 key pair and one ML-DSA-65
 * signing key are parsed ONCE into resident key tables (what kem.Scheme.UnmarshalBinaryPrivateKey / sign.Scheme.UnmarshalBinaryPrivateKey
 * keep in the key object: kem/mlkem/mlkem768/kyber.go:39-43, sign/mldsa/mldsa65/internal/dilithium.go:149-179); every later call then
 * moves only ciphertexts, messages and results.  Small calls are one kernel launch each.  Build (after `make lib`):
 *   gcc -O2 -Iinclude examples/resident_keys.c -Lcircl_amd -lcirclhip -Wl,-rpath,$PWD/circl_amd -Wl,-rpath,/opt/rocm/lib -o build/resident_keys */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "circl_hip.h"

static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

int main(void) {
    if (circl_hip_init() <= 0) { fprintf(stderr, "no HIP device: %s\n", circl_hip_last_error()); return 2; }
    enum { EK = 1184, DK = 2400, CT = 1088, PK = 1952, SK = 4032, SIG = 3309, N = 64 };
    static uint8_t seed[64], ek[EK], dk[DK], dseed[32], pk[PK], sk[SK], m[32 * N], ct[CT * N], ss[32 * N], ss2[32 * N], st[N], sig[SIG * N], ok[N];
    for (size_t i = 0; i < sizeof seed; i++) seed[i] = (uint8_t)(7 * i + 1);  /* a real caller draws these from its CSPRNG */
    for (size_t i = 0; i < sizeof dseed; i++) dseed[i] = (uint8_t)(11 * i + 3);
    for (size_t i = 0; i < sizeof m; i++) m[i] = (uint8_t)(i * 29 + (i >> 7));
    int rc = circl_hip_mlkem_keygen(768, seed, ek, dk, 1, 0);
    if (!rc) rc = circl_hip_mldsa_keygen(65, dseed, pk, sk, 1, 0);
    /* parse once: device 0 here; CIRCL_HIP_ALL_DEVICES would replicate the tables and shard every call's batch */
    circl_hip_keytable *kpub = NULL, *kprv = NULL, *signer = NULL, *verifier = NULL;
    uint8_t verdict = 0;
    if (!rc) rc = circl_hip_mlkem_keytable_new(768, 0, ek, 1, 0, NULL, &kpub);
    if (!rc) rc = circl_hip_mlkem_keytable_new(768, 1, dk, 1, 0, &verdict, &kprv); /* verdict 2 = kem.ErrPrivKey (stored hash mismatch) */
    if (!rc) rc = circl_hip_mldsa_privkeys_new(65, sk, 1, 0, &signer);
    if (!rc) rc = circl_hip_mldsa_keytable_new(65, pk, 1, 0, &verifier);
    if (rc || verdict) { fprintf(stderr, "setup failed: %d / %d: %s\n", rc, verdict, circl_hip_last_error()); return 1; }
    /* messages for the signer: N transcripts of 100 bytes, empty contexts */
    static uint8_t blob[100 * N + 1];
    uint64_t off[N + 1];
    for (int i = 0; i <= N; i++) off[i] = 100u * (uint64_t)i;
    for (size_t i = 0; i < sizeof blob; i++) blob[i] = (uint8_t)(i * 13 + 5);
    for (int round = 0; round < 3; round++) { /* call after call on the same tables; key_idx == NULL: entry 0 */
        const size_t n = round == 0 ? 1 : N;
        const double t0 = now_us();
        rc = circl_hip_mlkem_encaps_table(kpub, NULL, m, ct, ss, st, n);
        const double t1 = now_us();
        if (!rc) rc = circl_hip_mlkem_decaps_table(kprv, NULL, ct, ss2, st, n);
        const double t2 = now_us();
        if (!rc) rc = circl_hip_mldsa_sign_table(signer, blob, off, NULL, NULL, NULL /* deterministic */, sig, n);
        const double t3 = now_us();
        if (!rc) rc = circl_hip_mldsa_verify_table(verifier, NULL, sig, blob, off, NULL, NULL, ok, n);
        const double t4 = now_us();
        if (rc) { fprintf(stderr, "circl-hip error %d: %s\n", rc, circl_hip_last_error()); return 1; }
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) bad += st[i] != 0 || ok[i] != 1 || memcmp(ss + 32 * i, ss2 + 32 * i, 32) != 0;
        printf("n=%-3zu encaps %6.0f us  decaps %6.0f us  sign %6.0f us  verify %6.0f us  mismatches %zu\n", n, t1 - t0, t2 - t1, t3 - t2, t4 - t3, bad);
        if (bad) return 1;
    }
    circl_hip_keytable_free(kpub);
    circl_hip_keytable_free(kprv);     /* private tables are wiped before their memory is released */
    circl_hip_keytable_free(signer);
    circl_hip_keytable_free(verifier);
    return 0;
}
