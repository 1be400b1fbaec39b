/* tests/cgo_shape_test.c -- drives the C ABI exactly the way cgo's generated stubs for go/.../hipbatch do, so that the only
 * part of the Go bridge this repository cannot test (there is no Go toolchain in the image) is Go syntax:
 *   - every buffer is a byte-aligned SUB-SLICE of a larger allocation (Go slices carry no alignment promise),
 *   - zero-length batches pass NULL pointers (ptr() in hipbatch.go returns nil for an empty slice),
 *   - status == NULL where the bridge has no use for it,
 *   - blobs carry one spare byte and n + 1 uint64 offsets built by appending (VerifyBatch / SignBatch),
 *   - key tables + uint32 index vectors (EncapsulateKeyedBatch, VerifyKeyedBatch),
 *   - the hybrid KEMs' packed keys and []x25519.Key arrays (kem/hybrid/hipbatch, dh/x25519/hipbatch),
 *   - the same call from several threads at once (goroutines on different OS threads).
 * Results are checked against the ABI's own other paths (keygen -> encaps -> decaps round trips; sign -> verify; keyed
 * == per-item); bit-exactness against the oracle is the job of the Python tests.  Prints OK.
 *   gcc -O1 -pthread -Iinclude tests/cgo_shape_test.c -Lcircl_amd -lcirclhip -Wl,-rpath,$PWD/circl_amd -Wl,-rpath,/opt/rocm/lib */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "circl_hip.h"

#define CHECK(c)                                                                                    \
    do {                                                                                            \
        if (!(c)) { fprintf(stderr, "%s:%d: check failed: %s (%s)\n", __FILE__, __LINE__, #c, circl_hip_last_error()); exit(1); } \
    } while (0)

/* a "Go slice": n bytes starting `skew` bytes into a fresh allocation */
static uint8_t *slice(size_t n, size_t skew) {
    uint8_t *base = malloc(n + 64);
    CHECK(base);
    memset(base, 0, n + 64);
    return base + skew; /* (never freed: test process) */
}
static void fill(uint8_t *p, size_t n, unsigned seed) {
    for (size_t i = 0; i < n; i++) p[i] = (uint8_t)((i * 2654435761u + seed * 40503u) >> 11);
}

static void kem_round_trip(int param, size_t n, int device) {
    const size_t EK = circl_hip_mlkem_ek_size(param), DK = circl_hip_mlkem_dk_size(param), CT = circl_hip_mlkem_ct_size(param);
    uint8_t *seed = slice(64 * n, 1), *m = slice(32 * n, 3), *ek = slice(EK * n, 5), *dk = slice(DK * n, 7), *ct = slice(CT * n, 9);
    uint8_t *ss = slice(32 * n, 11), *ss2 = slice(32 * n, 13), *st = slice(n, 15);
    fill(seed, 64 * n, (unsigned)param);
    fill(m, 32 * n, (unsigned)param + 1);
    CHECK(circl_hip_mlkem_keygen(param, seed, ek, dk, n, device) == 0);
    CHECK(circl_hip_mlkem_encaps(param, ek, m, ct, ss, st, n, device) == 0);
    for (size_t i = 0; i < n; i++) CHECK(st[i] == 0);
    CHECK(circl_hip_mlkem_decaps(param, dk, ct, ss2, NULL /* the bridge may not want the status */, n, device) == 0);
    CHECK(memcmp(ss, ss2, 32 * n) == 0);
    /* key table: the first 5 keys, index i mod 5 */
    const size_t nk = n < 5 ? n : 5;
    uint32_t *idx = (uint32_t *)slice(4 * n + 8, 4); /* a []uint32 is 4-byte aligned in Go */
    uint8_t *ekg = slice(EK * n, 1), *ct2 = slice(CT * n, 3), *ss3 = slice(32 * n, 5), *ct3 = slice(CT * n, 7), *ss4 = slice(32 * n, 9);
    for (size_t i = 0; i < n; i++) { idx[i] = (uint32_t)(i % nk); memcpy(ekg + EK * i, ek + EK * idx[i], EK); }
    CHECK(circl_hip_mlkem_encaps_keyed(param, ek, nk, idx, m, ct2, ss3, st, n, device) == 0);
    CHECK(circl_hip_mlkem_encaps(param, ekg, m, ct3, ss4, st, n, device) == 0);
    CHECK(memcmp(ct2, ct3, CT * n) == 0 && memcmp(ss3, ss4, 32 * n) == 0);
    CHECK(circl_hip_mlkem_decaps_keyed(param, dk, nk, idx, ct2, ss2, st, n, device) == 0);
    CHECK(memcmp(ss2, ss3, 32 * n) == 0);
    uint32_t bad_idx[1] = {(uint32_t)nk};
    CHECK(circl_hip_mlkem_encaps_keyed(param, ek, nk, bad_idx, m, ct2, ss3, st, 1, device) == CIRCL_HIP_EPARAM);
    /* go/kem/mlkem/hipbatch/keytable.go: the same table kept resident across calls (one device: logical device 0) */
    circl_hip_keytable *pub = NULL, *prv = NULL;
    uint8_t *kst = slice(nk, 3);
    CHECK(circl_hip_mlkem_keytable_new(param, 0, ek, nk, 0, NULL, &pub) == 0 && pub != NULL);
    CHECK(circl_hip_mlkem_keytable_new(param, 1, dk, nk, 0, kst, &prv) == 0 && prv != NULL);
    for (size_t i = 0; i < nk; i++) CHECK(kst[i] == 0);
    for (int rep = 0; rep < 2; rep++) { /* call after call on one table */
        memset(ct3, 0, CT * n);
        CHECK(circl_hip_mlkem_encaps_table(pub, idx, m, ct3, ss4, st, n) == 0);
        CHECK(memcmp(ct2, ct3, CT * n) == 0 && memcmp(ss3, ss4, 32 * n) == 0);
        CHECK(circl_hip_mlkem_decaps_table(prv, idx, ct3, ss2, NULL, n) == 0);
        CHECK(memcmp(ss2, ss3, 32 * n) == 0);
    }
    CHECK(circl_hip_mlkem_encaps_table(pub, NULL /* nil index slice: entry 0 */, m, ct3, ss4, st, n) == 0);
    CHECK(circl_hip_mlkem_decaps_table(prv, NULL, ct3, ss2, st, n) == 0);
    CHECK(memcmp(ss2, ss4, 32 * n) == 0);
    CHECK(circl_hip_mlkem_encaps_table(pub, bad_idx, m, ct2, ss3, st, 1) == CIRCL_HIP_EPARAM);
    CHECK(circl_hip_mlkem_encaps_table(prv, NULL, m, ct2, ss3, st, 1) == CIRCL_HIP_EPARAM); /* a private table does not encapsulate */
    circl_hip_keytable_free(pub);
    circl_hip_keytable_free(prv);
    circl_hip_keytable_free(NULL);
    /* round 4: a table replicated on every device (calls shard the batch) and PrivateKey.Public() over the batch */
    CHECK(circl_hip_mlkem_keytable_new(param, 0, ek, nk, CIRCL_HIP_ALL_DEVICES, NULL, &pub) == 0 && circl_hip_keytable_device(pub) == CIRCL_HIP_ALL_DEVICES);
    CHECK(circl_hip_mlkem_encaps_table(pub, idx, m, ct2, ss3, st, n) == 0);
    CHECK(circl_hip_mlkem_encaps_keyed(param, ek, nk, idx, m, ct3, ss4, st, n, device) == 0);
    CHECK(memcmp(ct2, ct3, CT * n) == 0 && memcmp(ss3, ss4, 32 * n) == 0);
    circl_hip_keytable_free(pub);
    uint8_t *ek2 = slice(circl_hip_mlkem_ek_size(param) * n, 9);
    CHECK(circl_hip_mlkem_public_from_private(param, dk, ek2, n) == 0);
    CHECK(memcmp(ek, ek2, circl_hip_mlkem_ek_size(param) * n) == 0);
}

static void dsa_round_trip(int param, size_t n, int device) {
    const size_t PK = circl_hip_mldsa_pk_size(param), SK = circl_hip_mldsa_sk_size(param), SIG = circl_hip_mldsa_sig_size(param);
    uint8_t *seed = slice(32 * n, 1), *pk = slice(PK * n, 3), *sk = slice(SK * n, 5), *sig = slice(SIG * n, 7), *ok = slice(n, 9);
    fill(seed, 32 * n, (unsigned)param);
    CHECK(circl_hip_mldsa_keygen(param, seed, pk, sk, n, device) == 0);
    /* blobs as VerifyBatch builds them: appended rows, one spare byte, n + 1 offsets */
    uint64_t *moff = (uint64_t *)slice(8 * (n + 1), 8), *coff = (uint64_t *)slice(8 * (n + 1), 8);
    size_t mt = 0, ctot = 0;
    for (size_t i = 0; i < n; i++) { moff[i] = mt; coff[i] = ctot; mt += 1 + i % 70; ctot += i % 9; }
    moff[n] = mt; coff[n] = ctot;
    uint8_t *mblob = slice(mt + 1, 3), *cblob = slice(ctot + 1, 5);
    fill(mblob, mt, 77);
    fill(cblob, ctot, 78);
    CHECK(circl_hip_mldsa_sign(param, sk, mblob, moff, cblob, coff, NULL /* deterministic */, sig, n, device) == 0);
    CHECK(circl_hip_mldsa_verify(param, pk, sig, mblob, moff, cblob, coff, ok, n, device) == 0);
    for (size_t i = 0; i < n; i++) CHECK(ok[i] == 1);
    sig[SIG * (n / 2) + 40] ^= 1;
    uint32_t *idx = (uint32_t *)slice(4 * n + 8, 4);
    for (size_t i = 0; i < n; i++) idx[i] = (uint32_t)i; /* the table is the whole key array here */
    CHECK(circl_hip_mldsa_verify_keyed(param, pk, n, idx, sig, mblob, moff, cblob, coff, ok, n, device) == 0);
    for (size_t i = 0; i < n; i++) CHECK(ok[i] == (i == n / 2 ? 0 : 1));
    /* go/sign/mldsa/hipbatch/keytable.go: resident public keys, one prepared private key */
    circl_hip_keytable *vt = NULL, *st1 = NULL;
    CHECK(circl_hip_mldsa_keytable_new(param, pk, n, 0, &vt) == 0 && vt != NULL);
    memset(ok, 7, n);
    CHECK(circl_hip_mldsa_verify_table(vt, idx, sig, mblob, moff, cblob, coff, ok, n) == 0);
    for (size_t i = 0; i < n; i++) CHECK(ok[i] == (i == n / 2 ? 0 : 1));
    CHECK(circl_hip_mldsa_privkey_new(param, sk, 0, &st1) == 0 && st1 != NULL);
    uint8_t *sig1 = slice(SIG * n, 11), *sig2 = slice(SIG * n, 13);
    CHECK(circl_hip_mldsa_sign_table(st1, mblob, moff, cblob, coff, NULL, sig1, n) == 0);
    CHECK(circl_hip_mldsa_sign_shared(param, sk, mblob, moff, cblob, coff, NULL, sig2, n, 0) == 0);
    CHECK(memcmp(sig1, sig2, SIG * n) == 0);
    CHECK(circl_hip_mldsa_verify_table(vt, NULL /* entry 0 */, sig1, mblob, moff, cblob, coff, ok, n) == 0);
    for (size_t i = 0; i < n; i++) CHECK(ok[i] == 1);
    CHECK(circl_hip_mldsa_sign_table(vt, mblob, moff, cblob, coff, NULL, sig1, n) == CIRCL_HIP_EPARAM); /* a public table does not sign */
    /* round 4: a table of SEVERAL prepared private keys with an index slice, replicated on every device (device = -1), and
     * PrivateKey.Public() over the batch (go/sign/mldsa/hipbatch/keytable.go: NewResidentPrivateKeys, Sign(idx, ...), PublicKeys) */
    circl_hip_keytable *stn = NULL;
    CHECK(circl_hip_mldsa_privkeys_new(param, sk, n, CIRCL_HIP_ALL_DEVICES, &stn) == 0 && stn != NULL);
    CHECK(circl_hip_keytable_device(stn) == CIRCL_HIP_ALL_DEVICES && circl_hip_keytable_nkeys(stn) == n && circl_hip_keytable_on_device(stn, 0) != NULL);
    CHECK(circl_hip_mldsa_sign_table_keyed(stn, idx, mblob, moff, cblob, coff, NULL, sig1, n) == 0);
    CHECK(circl_hip_mldsa_sign(param, sk, mblob, moff, cblob, coff, NULL, sig2, n, device) == 0);
    CHECK(memcmp(sig1, sig2, SIG * n) == 0);
    uint32_t bad_idx[1] = {(uint32_t)n};
    CHECK(circl_hip_mldsa_sign_table_keyed(stn, bad_idx, mblob, moff, cblob, coff, NULL, sig1, 1) == CIRCL_HIP_EPARAM);
    uint8_t *pk2 = slice(PK * n, 15);
    CHECK(circl_hip_mldsa_public_from_private(param, sk, pk2, n, device) == 0);
    CHECK(memcmp(pk, pk2, PK * n) == 0);
    circl_hip_keytable_free(stn);
    circl_hip_keytable_free(vt);
    circl_hip_keytable_free(st1);
}

/* go/kem/hybrid/hipbatch + go/dh/x25519/hipbatch: packed keys / ciphertexts as byte-aligned sub-slices, NULL status */
static void hybrid_round_trip(int scheme, size_t n, int device) {
    const size_t SEED = circl_hip_hybrid_seed_size(scheme), ES = circl_hip_hybrid_eseed_size(scheme), PK = circl_hip_hybrid_pk_size(scheme),
                 SK = circl_hip_hybrid_sk_size(scheme), CT = circl_hip_hybrid_ct_size(scheme), SS = circl_hip_hybrid_ss_size(scheme);
    CHECK(SEED && PK == 1216 && CT == 1120);
    uint8_t *seed = slice(SEED * n, 1), *es = slice(ES * n, 3), *pk = slice(PK * n, 5), *sk = slice(SK * n, 7), *ct = slice(CT * n, 9);
    uint8_t *ss = slice(SS * n, 11), *ss2 = slice(SS * n, 13), *st = slice(n, 15);
    fill(seed, SEED * n, (unsigned)scheme + 50);
    fill(es, ES * n, (unsigned)scheme + 51);
    CHECK(circl_hip_hybrid_keygen(scheme, seed, pk, sk, n, device) == 0);
    CHECK(circl_hip_hybrid_encaps(scheme, pk, es, ct, ss, st, n, device) == 0);
    CHECK(circl_hip_hybrid_decaps(scheme, sk, ct, ss2, NULL, n, device) == 0);
    for (size_t i = 0; i < n; i++) CHECK(st[i] == 0);
    CHECK(memcmp(ss, ss2, SS * n) == 0);
    if (scheme == CIRCL_HIP_HYBRID_XWING || scheme == CIRCL_HIP_HYBRID_X25519MLKEM768) {
        /* go/kem/hybrid/hipbatch/keytable.go: the same keys parsed once (an X-Wing seed is expanded when the table is built), index slice */
        const size_t nk = n < 9 ? n : 9;
        circl_hip_keytable *hp = NULL, *hs = NULL;
        uint8_t *kst = slice(nk, 1), *ct2 = slice(CT * n, 3), *ss3 = slice(SS * n, 5), *ss4 = slice(SS * n, 7);
        uint32_t *idx = (uint32_t *)slice(4 * n + 8, 4);
        for (size_t i = 0; i < n; i++) idx[i] = (uint32_t)(i % nk);
        CHECK(circl_hip_hybrid_keytable_new(scheme, 0, pk, nk, device, NULL, &hp) == 0 && hp != NULL);
        CHECK(circl_hip_hybrid_keytable_new(scheme, 1, sk, nk, device, kst, &hs) == 0 && hs != NULL);
        for (size_t i = 0; i < nk; i++) CHECK(kst[i] == 0);
        CHECK(circl_hip_hybrid_encaps_table(hp, idx, es, ct2, ss3, st, n) == 0);
        CHECK(circl_hip_hybrid_decaps_table(hs, idx, ct2, ss4, NULL, n) == 0);
        CHECK(memcmp(ss3, ss4, SS * n) == 0);
        if (nk == n) CHECK(memcmp(ct, ct2, CT * n) == 0 && memcmp(ss, ss3, SS * n) == 0); /* identity index: the per-item entry point's bytes */
        CHECK(circl_hip_hybrid_encaps_table(hs, NULL, es, ct2, ss3, st, 1) == CIRCL_HIP_EPARAM); /* a private table does not encapsulate */
        circl_hip_keytable_free(hp);
        circl_hip_keytable_free(hs);
    }
    /* the X25519 half alone: ct_X = KeyGen(ephemeral), and Shared agrees from both sides */
    uint8_t *a = slice(32 * n, 1), *b = slice(32 * n, 2), *pa = slice(32 * n, 3), *pb = slice(32 * n, 5), *s1 = slice(32 * n, 7), *s2 = slice(32 * n, 9), *ok = slice(n, 1);
    fill(a, 32 * n, 60);
    fill(b, 32 * n, 61);
    CHECK(circl_hip_x25519(a, NULL, pa, NULL, n, device) == 0);
    CHECK(circl_hip_x25519(b, NULL, pb, NULL, n, device) == 0);
    CHECK(circl_hip_x25519(a, pb, s1, ok, n, device) == 0);
    CHECK(circl_hip_x25519(b, pa, s2, NULL, n, device) == 0);
    CHECK(memcmp(s1, s2, 32 * n) == 0);
    for (size_t i = 0; i < n; i++) CHECK(ok[i] == 1);
}

static void *thread_main(void *arg) {
    kem_round_trip(768, 3000 + 100 * (size_t)(intptr_t)arg, 0);
    return NULL;
}

/* go/xof/hipbatch + go/simd/keccakf1600/hipbatch: blobs with spare capacity behind the last message, n + 1 offsets, nil contexts,
 * plain [25]uint64 states */
static void xof_shapes(int device) {
    const size_t n = 333, len = 200, outlen = 48;
    uint8_t *blob = slice(n * len + 16, 3), *eq = slice(n * len, 5), *o1 = slice(n * outlen, 7), *o2 = slice(n * outlen, 9), *o3 = slice(n * outlen, 11);
    uint64_t *off = (uint64_t *)slice(8 * (n + 1), 8), *zoff = (uint64_t *)slice(8 * (n + 1), 8);
    fill(blob, n * len, 91);
    memcpy(eq, blob, n * len);
    for (size_t i = 0; i <= n; i++) { off[i] = i * len; zoff[i] = 0; }
    CHECK(circl_hip_xof(136, 0x1f, 24, blob, off, o1, outlen, n, device) == 0);       /* SumBatch(xof.SHAKE256, ...) */
    CHECK(circl_hip_shake(136, 0x1f, eq, len, o2, outlen, n, device) == 0);           /* the equal-length form of the same sponges */
    CHECK(memcmp(o1, o2, n * outlen) == 0);
    CHECK(circl_hip_xof(168, 0x07, 12, blob, off, o1, outlen, n, device) == 0);       /* TurboShakeBatch(128, 0x07, ...) */
    CHECK(circl_hip_k12(blob, off, NULL, NULL, o2, outlen, n, device) == 0);          /* K12Batch(msgs, nil, ...) */
    uint8_t *cb = slice(16, 1);
    CHECK(circl_hip_k12(blob, off, cb, zoff, o3, outlen, n, device) == 0);            /* explicit empty contexts: the same digests */
    CHECK(memcmp(o2, o3, n * outlen) == 0);
    CHECK(memcmp(o1, o2, n * outlen) != 0);  /* (K12 hashes msg || 0x00 -- length_encode(0) -- where TurboSHAKE128 above hashed msg) */
    uint64_t *st = (uint64_t *)slice(8 * 25 * 77, 8), *st2 = (uint64_t *)slice(8 * 25 * 77, 8);
    fill((uint8_t *)st, 8 * 25 * 77, 17);
    memcpy(st2, st, 8 * 25 * 77);
    CHECK(circl_hip_keccak_f1600(st, 77, 24, device) == 0);                           /* PermuteBatch(states, false, ...) */
    CHECK(circl_hip_keccak_f1600(st2, 77, 24, CIRCL_HIP_ALL_DEVICES) == 0);
    CHECK(memcmp(st, st2, 8 * 25 * 77) == 0);
    CHECK(circl_hip_keccak_f1600(st, 77, 12, device) == 0);                           /* turbo */
    CHECK(circl_hip_keccak_f1600(st, 77, 7, device) == CIRCL_HIP_EPARAM);
}

/* go/kem/mlkem/hipbatch/keytable.go ResidentTable.SetCoalesce: eight "goroutines" each encapsulate / decapsulate ONE item per call
 * (the shape of kem.Scheme.Encapsulate / Decapsulate, kem/mlkem/mlkem768/kyber.go:347-386) through one coalescing table */
struct co_job { circl_hip_keytable *pub, *prv; const uint8_t *m, *ct, *ss; size_t CT, n; int id; };
static void *co_thread(void *arg) {
    const struct co_job *j = arg;
    uint8_t *ct1 = slice(j->CT, 1 + (size_t)j->id), *ss1 = slice(32, 3), *ss2 = slice(32, 5);
    for (size_t r = 0; r < 40; r++) {
        const size_t i = ((size_t)j->id * 29 + r * 7) % j->n;
        CHECK(circl_hip_mlkem_encaps_table(j->pub, NULL, j->m + 32 * i, ct1, ss1, NULL, 1) == 0);
        CHECK(memcmp(ct1, j->ct + j->CT * i, j->CT) == 0 && memcmp(ss1, j->ss + 32 * i, 32) == 0);
        CHECK(circl_hip_mlkem_decaps_table(j->prv, NULL, ct1, ss2, NULL, 1) == 0);
        CHECK(memcmp(ss1, ss2, 32) == 0);
    }
    return NULL;
}
static void coalesced_single_calls(void) {
    const int param = 768;
    const size_t n = 64, EK = circl_hip_mlkem_ek_size(param), DK = circl_hip_mlkem_dk_size(param), CT = circl_hip_mlkem_ct_size(param);
    uint8_t *seed = slice(64, 1), *ek = slice(EK, 3), *dk = slice(DK, 5), *m = slice(32 * n, 7), *ct = slice(CT * n, 9), *ss = slice(32 * n, 11);
    fill(seed, 64, 77);
    fill(m, 32 * n, 78);
    CHECK(circl_hip_mlkem_keygen(param, seed, ek, dk, 1, 0) == 0);
    CHECK(circl_hip_mlkem_encaps_shared(param, ek, m, ct, ss, NULL, n, 0) == 0); /* the answers, from one ordinary batch call */
    struct co_job j = {NULL, NULL, m, ct, ss, CT, n, 0};
    CHECK(circl_hip_mlkem_keytable_new(param, 0, ek, 1, CIRCL_HIP_ALL_DEVICES, NULL, &j.pub) == 0);
    CHECK(circl_hip_mlkem_keytable_new(param, 1, dk, 1, 0, NULL, &j.prv) == 0);
    CHECK(circl_hip_keytable_set_coalesce(j.pub, 64, 0) == 0 && circl_hip_keytable_set_coalesce(j.prv, 64, 100) == 0);
    CHECK(circl_hip_keytable_set_coalesce(NULL, 64, 0) == CIRCL_HIP_EPARAM);
    pthread_t th[8];
    struct co_job jobs[8];
    for (int t = 0; t < 8; t++) { jobs[t] = j; jobs[t].id = t; CHECK(pthread_create(&th[t], NULL, co_thread, &jobs[t]) == 0); }
    for (int t = 0; t < 8; t++) pthread_join(th[t], NULL);
    uint64_t calls = 0, items = 0, launches = 0;
    CHECK(circl_hip_keytable_coalesce_stats(j.pub, &calls, &items, &launches) == 0 && calls == 8 * 40 && items == calls && launches >= 1 && launches <= calls);
    CHECK(circl_hip_keytable_set_coalesce(j.pub, 0, 0) == 0); /* off again */
    circl_hip_keytable_free(j.pub);
    circl_hip_keytable_free(j.prv);
}

/* The asynchronous form as go/kem/mlkem/hipbatch/reactor.go drives it: request goroutines (threads here) hand their request to ONE
 * device goroutine and park on a channel (a condition variable here); the device goroutine submits, polls the head of its ticket FIFO
 * and wakes the owners.  cgo forbids C to keep a Go pointer after the call returns, so the OUTPUT rows that the library writes later
 * live in a C-allocated arena (malloc here, C.malloc there); the INPUT rows are ordinary (unaligned) Go slices: copied before the
 * submit returns. */
enum { AQ_SLOTS = 32, AQ_PRODUCERS = 6, AQ_PER_PRODUCER = 50 };
struct areq { size_t item; int decaps; int done; uint8_t ss[32]; uint8_t ct[1088]; };
struct aq {
    pthread_mutex_t mu; pthread_cond_t cv_req, cv_done;
    struct areq *pending[AQ_SLOTS * 4]; size_t head, tail; int producers_left;
    circl_hip_keytable *pub, *prv; const uint8_t *m, *ct, *ss; size_t CT, n;
};
static void *aq_producer(void *arg) {
    struct aq *q = arg;
    for (int r = 0; r < AQ_PER_PRODUCER; r++) {
        struct areq rq = {((size_t)pthread_self() / 64 + (size_t)r * 13) % q->n, r % 3 == 0, 0, {0}, {0}};
        pthread_mutex_lock(&q->mu);
        q->pending[q->tail++ % (AQ_SLOTS * 4)] = &rq;
        pthread_cond_signal(&q->cv_req);
        while (!rq.done) pthread_cond_wait(&q->cv_done, &q->mu); /* parked: no thread sits inside the library for this request */
        pthread_mutex_unlock(&q->mu);
        CHECK(rq.done == 1);
        CHECK(memcmp(rq.ss, q->ss + 32 * rq.item, 32) == 0);
        if (!rq.decaps) CHECK(memcmp(rq.ct, q->ct + q->CT * rq.item, q->CT) == 0);
    }
    pthread_mutex_lock(&q->mu);
    q->producers_left--;
    pthread_cond_signal(&q->cv_req);
    pthread_mutex_unlock(&q->mu);
    return NULL;
}
static void async_submit_poll(void) {
    const int param = 768;
    const size_t n = 64, EK = circl_hip_mlkem_ek_size(param), DK = circl_hip_mlkem_dk_size(param), CT = circl_hip_mlkem_ct_size(param);
    uint8_t *seed = slice(64, 1), *ek = slice(EK, 3), *dk = slice(DK, 5), *m = slice(32 * n, 7), *ct = slice(CT * n, 9), *ss = slice(32 * n, 11);
    fill(seed, 64, 91);
    fill(m, 32 * n, 92);
    CHECK(circl_hip_mlkem_keygen(param, seed, ek, dk, 1, 0) == 0);
    CHECK(circl_hip_mlkem_encaps_shared(param, ek, m, ct, ss, NULL, n, 0) == 0);
    struct aq q = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, AQ_PRODUCERS, NULL, NULL, m, ct, ss, CT, n};
    CHECK(circl_hip_mlkem_keytable_new(param, 0, ek, 1, 0, NULL, &q.pub) == 0);
    CHECK(circl_hip_mlkem_keytable_new(param, 1, dk, 1, 0, NULL, &q.prv) == 0);
    uint64_t tk = 0;
    CHECK(circl_hip_mlkem_encaps_table_submit(q.pub, NULL, m, ct, ss, NULL, 1, &tk) == CIRCL_HIP_EPARAM); /* no queue yet */
    CHECK(circl_hip_keytable_async_start(q.pub, 64, 0, 1) == 0 && circl_hip_keytable_async_start(q.prv, 64, 0, 0) == 0);
    CHECK(circl_hip_keytable_async_start(NULL, 64, 0, 0) == CIRCL_HIP_EPARAM);
    const int efd = circl_hip_keytable_eventfd(q.pub, 0);
    CHECK(efd >= 0 && circl_hip_keytable_eventfd(q.prv, 0) == -1);
    /* the C arena of the in-flight requests' output rows, and who owns each slot */
    uint8_t *arena_ct = malloc(CT * AQ_SLOTS), *arena_ss = malloc(32 * AQ_SLOTS), *arena_st = malloc(AQ_SLOTS);
    struct areq *owner[AQ_SLOTS];
    uint64_t ticket[AQ_SLOTS];
    size_t s_head = 0, s_tail = 0; /* slots in flight: [s_head, s_tail), FIFO -- per queue tickets finish in issue order; two queues: poll both heads */
    pthread_t th[AQ_PRODUCERS];
    for (int t = 0; t < AQ_PRODUCERS; t++) CHECK(pthread_create(&th[t], NULL, aq_producer, &q) == 0);
    size_t submitted = 0, finished = 0;
    for (;;) {
        /* take what is pending (never blocks while something is in flight) */
        pthread_mutex_lock(&q.mu);
        while (q.head == q.tail && s_head == s_tail && q.producers_left > 0) pthread_cond_wait(&q.cv_req, &q.mu);
        struct areq *batch[AQ_SLOTS];
        size_t nb = 0;
        while (q.head != q.tail && (s_tail - s_head) + nb < AQ_SLOTS) batch[nb++] = q.pending[q.head++ % (AQ_SLOTS * 4)];
        const int over = q.producers_left == 0 && q.head == q.tail;
        pthread_mutex_unlock(&q.mu);
        for (size_t k = 0; k < nb; k++) {
            const size_t sl = s_tail % AQ_SLOTS;
            struct areq *rq = batch[k];
            int rc;
            for (;;) {
                if (rq->decaps) rc = circl_hip_mlkem_decaps_table_submit(q.prv, NULL, ct + CT * rq->item, arena_ss + 32 * sl, arena_st + sl, 1, &ticket[sl]);
                else rc = circl_hip_mlkem_encaps_table_submit(q.pub, NULL, m + 32 * rq->item, arena_ct + CT * sl, arena_ss + 32 * sl, arena_st + sl, 1, &ticket[sl]);
                if (rc != CIRCL_HIP_EAGAIN) break;
                CHECK(s_head != s_tail); /* every batch busy means something of ours is in flight: wait for the oldest, try again */
                CHECK(circl_hip_wait(owner[s_head % AQ_SLOTS]->decaps ? q.prv : q.pub, ticket[s_head % AQ_SLOTS], 1000000) == 1);
            }
            CHECK(rc == 0);
            owner[sl] = rq;
            s_tail++;
            submitted++;
        }
        /* reap in issue order (the two tables are two queues: a head that is still pending stops the sweep; it is waited for below) */
        while (s_head != s_tail) {
            const size_t sl = s_head % AQ_SLOTS;
            int8_t st = 0;
            CHECK(circl_hip_poll(owner[sl]->decaps ? q.prv : q.pub, &ticket[sl], 1, &st) >= 0);
            if (st == 0) {
                if (nb) break; /* new work arrived this turn: look for more before blocking */
                CHECK(circl_hip_wait(owner[sl]->decaps ? q.prv : q.pub, ticket[sl], 1000000) == 1);
            } else {
                CHECK(st == 1);
            }
            CHECK(arena_st[sl] == 0);
            memcpy(owner[sl]->ss, arena_ss + 32 * sl, 32);
            if (!owner[sl]->decaps) memcpy(owner[sl]->ct, arena_ct + CT * sl, CT);
            pthread_mutex_lock(&q.mu);
            owner[sl]->done = 1;
            pthread_cond_broadcast(&q.cv_done);
            pthread_mutex_unlock(&q.mu);
            s_head++;
            finished++;
        }
        if (over && s_head == s_tail) break;
    }
    for (int t = 0; t < AQ_PRODUCERS; t++) pthread_join(th[t], NULL);
    CHECK(submitted == (size_t)AQ_PRODUCERS * AQ_PER_PRODUCER && finished == submitted);
    uint64_t calls = 0, items = 0, launches = 0, v = 0;
    CHECK(circl_hip_keytable_coalesce_stats(q.pub, &calls, &items, &launches) == 0 && calls >= 1 && items == calls && launches >= 1 && launches <= calls);
    CHECK(read(efd, &v, sizeof v) == (ssize_t)sizeof v && v == launches); /* the eventfd counted the finished batches */
    CHECK(circl_hip_keytable_async_stop(q.pub) == 0 && circl_hip_keytable_eventfd(q.pub, 0) == -1);
    CHECK(circl_hip_keytable_close(q.pub) == 0 && circl_hip_keytable_close(q.prv) == 0);
    free(arena_ct); free(arena_ss); free(arena_st);
}

int main(void) {
    CHECK(circl_hip_init() > 0);
    /* zero-length batches: nil slices become NULL pointers */
    CHECK(circl_hip_mlkem_encaps(768, NULL, NULL, NULL, NULL, NULL, 0, 0) == 0);
    CHECK(circl_hip_mlkem_decaps(768, NULL, NULL, NULL, NULL, 0, CIRCL_HIP_ALL_DEVICES) == 0);
    CHECK(circl_hip_mlkem_keygen(1024, NULL, NULL, NULL, 0, 0) == 0);
    CHECK(circl_hip_mldsa_verify(65, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, 0) == 0);
    CHECK(circl_hip_mldsa_sign(65, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, 0) == 0);
    CHECK(circl_hip_mlkem_encaps(999, NULL, NULL, NULL, NULL, NULL, 0, 0) == CIRCL_HIP_EPARAM);
    kem_round_trip(512, 1, 0);      /* a batch of one */
    kem_round_trip(768, 4099, CIRCL_HIP_ALL_DEVICES);
    kem_round_trip(1024, 777, 0);
    dsa_round_trip(44, 1, 0);
    dsa_round_trip(65, 333, CIRCL_HIP_ALL_DEVICES);
    dsa_round_trip(87, 100, 0);
    CHECK(circl_hip_hybrid_encaps(CIRCL_HIP_HYBRID_XWING, NULL, NULL, NULL, NULL, NULL, 0, 0) == 0);
    CHECK(circl_hip_x25519(NULL, NULL, NULL, NULL, 0, 0) == 0);
    CHECK(circl_hip_hybrid_keygen(7, NULL, NULL, NULL, 0, 0) == CIRCL_HIP_EPARAM);
    hybrid_round_trip(CIRCL_HIP_HYBRID_XWING, 1, 0);
    hybrid_round_trip(CIRCL_HIP_HYBRID_XWING, 7, 0);
    hybrid_round_trip(CIRCL_HIP_HYBRID_X25519MLKEM768, 5, CIRCL_HIP_ALL_DEVICES);
    hybrid_round_trip(CIRCL_HIP_HYBRID_XWING, 2051, CIRCL_HIP_ALL_DEVICES);
    hybrid_round_trip(CIRCL_HIP_HYBRID_X25519MLKEM768, 777, 0);
    CHECK(circl_hip_xof(168, 0x1f, 24, NULL, NULL, NULL, 32, 0, 0) == 0);
    CHECK(circl_hip_keccak_f1600(NULL, 0, 24, 0) == 0);
    xof_shapes(0);
    coalesced_single_calls();
    async_submit_poll();
    pthread_t th[3];
    for (intptr_t i = 0; i < 3; i++) CHECK(pthread_create(&th[i], NULL, thread_main, (void *)i) == 0);
    for (int i = 0; i < 3; i++) pthread_join(th[i], NULL);
    printf("OK\n");
    return 0;
}
