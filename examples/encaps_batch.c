/* examples/encaps_batch.c -- the C ABI from plain C: derive N ML-KEM-768 key pairs, encapsulate to them,
 * decapsulate, compare.  Build (after `make -C .. lib`):
 *   gcc -O2 -I../include encaps_batch.c -L../circl_amd -lcirclhip -Wl,-rpath,'$ORIGIN/../circl_amd' -Wl,-rpath,/opt/rocm/lib -o encaps_batch
 * This is what the cgo bridge under go/ does, minus Go. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "circl_hip.h"

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 10000;
    const int param = 768;
    if (circl_hip_init() <= 0) { fprintf(stderr, "no HIP device: %s\n", circl_hip_last_error()); return 2; }
    const size_t EK = circl_hip_mlkem_ek_size(param), DK = circl_hip_mlkem_dk_size(param), CT = circl_hip_mlkem_ct_size(param);
    uint8_t *seed = malloc(64 * n), *m = malloc(32 * n), *ek = malloc(EK * n), *dk = malloc(DK * n), *ct = malloc(CT * n);
    uint8_t *ss = malloc(32 * n), *ss2 = malloc(32 * n), *st = malloc(n);
    if (!seed || !m || !ek || !dk || !ct || !ss || !ss2 || !st) return 1;
    for (size_t i = 0; i < 64 * n; i++) seed[i] = (uint8_t)(i * 131 + (i >> 8));  /* a real caller draws these from its CSPRNG */
    for (size_t i = 0; i < 32 * n; i++) m[i] = (uint8_t)(i * 29 + (i >> 7));
    int rc = circl_hip_mlkem_keygen(param, seed, ek, dk, n, 0);
    if (!rc) rc = circl_hip_mlkem_encaps(param, ek, m, ct, ss, st, n, 0);
    if (!rc) rc = circl_hip_mlkem_decaps(param, dk, ct, ss2, st, n, CIRCL_HIP_ALL_DEVICES);
    if (rc) { fprintf(stderr, "circl-hip error %d: %s\n", rc, circl_hip_last_error()); return 1; }
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) bad += st[i] != 0 || memcmp(ss + 32 * i, ss2 + 32 * i, 32) != 0;
    printf("%zu ML-KEM-768 round trips, %zu mismatches (%s)\n", n, bad, circl_hip_version());
    return bad != 0;
}
