/* oracle/batch.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 * pthread fan-out of the per-item oracle functions over contiguous slices; used by
 * the parity tests (to check whole batches in seconds) and by bench.py's
 * cpu_baseline leg ("port": scalar CPU restatement, not the reference's AVX2 path). */
#include "oracle.h"
#include <pthread.h>
#include <string.h>

typedef struct job {
    int kind, param, r3; /* r3: round-3 Kyber instead of ML-KEM */
    size_t lo, hi;
    const uint8_t *a, *b, *c, *d, *e;
    const uint64_t *off1, *off2;
    uint8_t *o1, *o2, *st;
} job;

enum { J_KEM_KEYGEN, J_KEM_ENCAPS, J_KEM_DECAPS, J_DSA_KEYGEN, J_DSA_SIGN, J_DSA_VERIFY, J_KEM_ENCAPS_SHARED };

static void *run(void *arg) {
    job *j = (job *)arg;
    int p = j->param;
    if (j->kind == J_KEM_ENCAPS_SHARED) { /* one parsed key per thread, its slice of the messages */
        size_t ct = orc_mlkem_ct_size(p);
        orc_mlkem_encaps_cached(p, j->a, j->b + 32 * j->lo, j->o1 + ct * j->lo, j->o2 + 32 * j->lo, j->hi - j->lo);
        return 0;
    }
    if (j->kind <= J_KEM_DECAPS) {
        size_t ek = orc_mlkem_ek_size(p), dk = orc_mlkem_dk_size(p), ct = orc_mlkem_ct_size(p);
        for (size_t i = j->lo; i < j->hi; i++) {
            if (j->kind == J_KEM_KEYGEN) {
                if (j->r3) orc_kyber_r3_keygen(p, j->a + 64 * i, j->o1 + ek * i, j->o2 + dk * i);
                else orc_mlkem_keygen(p, j->a + 64 * i, j->o1 + ek * i, j->o2 + dk * i);
            } else if (j->kind == J_KEM_ENCAPS) {
                int r = j->r3 ? orc_kyber_r3_encaps(p, j->a + ek * i, j->b + 32 * i, j->o1 + ct * i, j->o2 + 32 * i)
                              : orc_mlkem_encaps(p, j->a + ek * i, j->b + 32 * i, j->o1 + ct * i, j->o2 + 32 * i);
                if (r) { memset(j->o1 + ct * i, 0, ct); memset(j->o2 + 32 * i, 0, 32); }
                if (j->st) j->st[i] = (uint8_t)r;
            } else {
                int r = j->r3 ? orc_kyber_r3_decaps(p, j->a + dk * i, j->b + ct * i, j->o1 + 32 * i)
                              : orc_mlkem_decaps(p, j->a + dk * i, j->b + ct * i, j->o1 + 32 * i);
                if (r) memset(j->o1 + 32 * i, 0, 32);
                if (j->st) j->st[i] = (uint8_t)r;
            }
        }
    } else {
        size_t pk = orc_mldsa_pk_size(p), sk = orc_mldsa_sk_size(p), sg = orc_mldsa_sig_size(p);
        for (size_t i = j->lo; i < j->hi; i++) {
            if (j->kind == J_DSA_KEYGEN) {
                orc_mldsa_keygen(p, j->a + 32 * i, j->o1 + pk * i, j->o2 + sk * i);
            } else {
                const uint8_t *msg = j->b + j->off1[i];
                size_t ml = (size_t)(j->off1[i + 1] - j->off1[i]);
                const uint8_t *ctx = j->c ? j->c + j->off2[i] : (const uint8_t *)"";
                size_t cl = j->c ? (size_t)(j->off2[i + 1] - j->off2[i]) : 0;
                if (j->kind == J_DSA_SIGN)
                    orc_mldsa_sign(p, j->a + sk * i, msg, ml, ctx, cl, j->d + 32 * i, 0, j->o1 + sg * i);
                else
                    j->o1[i] = (uint8_t)orc_mldsa_verify(p, j->a + pk * i, msg, ml, ctx, cl, 0, j->e + sg * i, sg);
            }
        }
    }
    return 0;
}

static int fan(job *proto, size_t n, int threads) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    pthread_t th[256];
    job jobs[256];
    if (threads > 256) threads = 256;
    for (int t = 0; t < threads; t++) {
        jobs[t] = *proto;
        jobs[t].lo = n * (size_t)t / (size_t)threads;
        jobs[t].hi = n * (size_t)(t + 1) / (size_t)threads;
        if (threads == 1) { run(&jobs[0]); return 0; }
        if (pthread_create(&th[t], 0, run, &jobs[t])) return -1;
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], 0);
    return 0;
}

int orc_mlkem_keygen_batch(int param, const uint8_t *seed, uint8_t *ek, uint8_t *dk, size_t n, int threads) {
    if (!orc_mlkem_ek_size(param)) return -1;
    job j = {.kind = J_KEM_KEYGEN, .param = param, .a = seed, .o1 = ek, .o2 = dk};
    return fan(&j, n, threads);
}
int orc_mlkem_encaps_batch(int param, const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss,
                           uint8_t *status, size_t n, int threads) {
    if (!orc_mlkem_ek_size(param)) return -1;
    job j = {.kind = J_KEM_ENCAPS, .param = param, .a = ek, .b = m, .o1 = ct, .o2 = ss, .st = status};
    return fan(&j, n, threads);
}
int orc_mlkem_encaps_shared_batch(int param, const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss, size_t n, int threads) {
    if (!orc_mlkem_ek_size(param)) return -1;
    job j = {.kind = J_KEM_ENCAPS_SHARED, .param = param, .a = ek, .b = m, .o1 = ct, .o2 = ss};
    return fan(&j, n, threads);
}
int orc_mlkem_decaps_batch(int param, const uint8_t *dk, const uint8_t *ct, uint8_t *ss,
                           uint8_t *status, size_t n, int threads) {
    if (!orc_mlkem_ek_size(param)) return -1;
    job j = {.kind = J_KEM_DECAPS, .param = param, .a = dk, .b = ct, .o1 = ss, .st = status};
    return fan(&j, n, threads);
}
int orc_kyber_r3_keygen_batch(int param, const uint8_t *seed, uint8_t *ek, uint8_t *dk, size_t n, int threads) {
    if (!orc_mlkem_ek_size(param)) return -1;
    job j = {.kind = J_KEM_KEYGEN, .param = param, .r3 = 1, .a = seed, .o1 = ek, .o2 = dk};
    return fan(&j, n, threads);
}
int orc_kyber_r3_encaps_batch(int param, const uint8_t *ek, const uint8_t *seed, uint8_t *ct, uint8_t *ss, size_t n, int threads) {
    if (!orc_mlkem_ek_size(param)) return -1;
    job j = {.kind = J_KEM_ENCAPS, .param = param, .r3 = 1, .a = ek, .b = seed, .o1 = ct, .o2 = ss};
    return fan(&j, n, threads);
}
int orc_kyber_r3_decaps_batch(int param, const uint8_t *dk, const uint8_t *ct, uint8_t *ss, size_t n, int threads) {
    if (!orc_mlkem_ek_size(param)) return -1;
    job j = {.kind = J_KEM_DECAPS, .param = param, .r3 = 1, .a = dk, .b = ct, .o1 = ss};
    return fan(&j, n, threads);
}
int orc_mldsa_keygen_batch(int param, const uint8_t *seed, uint8_t *pk, uint8_t *sk, size_t n, int threads) {
    if (!orc_mldsa_pk_size(param)) return -1;
    job j = {.kind = J_DSA_KEYGEN, .param = param, .a = seed, .o1 = pk, .o2 = sk};
    return fan(&j, n, threads);
}
int orc_mldsa_sign_batch(int param, const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off,
                         const uint8_t *ctx_blob, const uint64_t *ctx_off, const uint8_t *rnd,
                         uint8_t *sig, size_t n, int threads) {
    if (!orc_mldsa_pk_size(param)) return -1;
    job j = {.kind = J_DSA_SIGN, .param = param, .a = sk, .b = msg_blob, .off1 = msg_off,
             .c = ctx_blob, .off2 = ctx_off, .d = rnd, .o1 = sig};
    return fan(&j, n, threads);
}
int orc_mldsa_verify_batch(int param, const uint8_t *pk, const uint8_t *sig, const uint8_t *msg_blob,
                           const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off,
                           uint8_t *ok, size_t n, int threads) {
    if (!orc_mldsa_pk_size(param)) return -1;
    job j = {.kind = J_DSA_VERIFY, .param = param, .a = pk, .b = msg_blob, .off1 = msg_off,
             .c = ctx_blob, .off2 = ctx_off, .e = sig, .o1 = ok};
    return fan(&j, n, threads);
}
