/* oracle/vec/dispatch.c -- TEST INFRASTRUCTURE ONLY (see mlkem_vec.c): picks the AVX-512 or the AVX2 build of the batch
 * encapsulation at run time and fans contiguous slices of the batch out over pthreads (the partitioning of oracle/batch.c). */
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int orcv_mlkem_encaps_avx2(int, const uint8_t *, int, const uint8_t *, uint8_t *, uint8_t *, uint8_t *, size_t, size_t);
int orcv_mlkem_encaps_avx512(int, const uint8_t *, int, const uint8_t *, uint8_t *, uint8_t *, uint8_t *, size_t, size_t);
void orcv_tables_avx2(void);
void orcv_tables_avx512(void);
int orcv_width_avx2(void);
int orcv_width_avx512(void);
void orcv_f1600_avx2(uint64_t *);
void orcv_f1600_avx512(uint64_t *);
int orcv_states_avx2(void);
int orcv_states_avx512(void);

/* 2 = AVX-512 (F, BW, VL, DQ, VBMI, VBMI2), 1 = AVX2, 0 = neither; `want` (0 = best) caps it */
int orcv_isa(int want) {
    __builtin_cpu_init();
    int have = 0;
    if (__builtin_cpu_supports("avx2")) have = 1;
    if (have && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
        __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vbmi") && __builtin_cpu_supports("avx512vbmi2"))
        have = 2;
    return (want > 0 && want < have) ? want : have;
}

typedef struct {
    int isa, param, rc, shared;
    const uint8_t *ek, *m;
    uint8_t *ct, *ss, *st;
    size_t lo, hi;
} job;

static void *run(void *a) {
    job *j = (job *)a;
    j->rc = (j->isa == 2 ? orcv_mlkem_encaps_avx512 : orcv_mlkem_encaps_avx2)(j->param, j->ek, j->shared, j->m, j->ct, j->ss, j->st, j->lo, j->hi);
    return 0;
}

/* n encapsulations (distinct keys: ek[n][384K+32], or shared != 0: ONE key ek[1][384K+32] for all; m[n][32] -> ct, ss, status[n] (may be
 * NULL)); slices are multiples of the vector width so that only the last group of the batch is ragged.  Returns 0, -1 (parameter set),
 * -2 (memory), -3 (no AVX2). */
int orcv_mlkem_encaps2(int param, const uint8_t *ek, int shared, const uint8_t *m, uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, int threads, int isa_want) {
    const int isa = orcv_isa(isa_want);
    if (!isa) return -3;
    if (param != 768 && param != 1024) return -1;
    static pthread_once_t once2 = PTHREAD_ONCE_INIT, once5 = PTHREAD_ONCE_INIT;
    pthread_once(isa == 2 ? &once5 : &once2, isa == 2 ? orcv_tables_avx512 : orcv_tables_avx2);
    const size_t W = (size_t)(isa == 2 ? orcv_width_avx512() : orcv_width_avx2());
    const size_t groups = (n + W - 1) / W;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if ((size_t)threads > groups) threads = groups ? (int)groups : 1;
    pthread_t th[256];
    job jobs[256];
    for (int t = 0; t < threads; t++) {
        size_t lo = groups * (size_t)t / (size_t)threads * W, hi = groups * (size_t)(t + 1) / (size_t)threads * W;
        if (hi > n) hi = n;
        jobs[t] = (job){isa, param, 0, shared, ek, m, ct, ss, status, lo, hi};
        if (threads == 1) {
            run(&jobs[0]);
            return jobs[0].rc;
        }
        if (pthread_create(&th[t], 0, run, &jobs[t])) { /* finish what was started, then do the rest on this thread */
            for (int u = 0; u < t; u++) pthread_join(th[u], 0);
            jobs[t].hi = n;
            run(&jobs[t]);
            int rc = jobs[t].rc;
            for (int u = 0; u < t; u++)
                if (jobs[u].rc) rc = jobs[u].rc;
            return rc;
        }
    }
    int rc = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], 0);
        if (jobs[t].rc) rc = jobs[t].rc;
    }
    return rc;
}

int orcv_mlkem_encaps(int param, const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, int threads, int isa_want) {
    return orcv_mlkem_encaps2(param, ek, 0, m, ct, ss, status, n, threads, isa_want);
}

int orcv_states(int isa) { return isa == 2 ? orcv_states_avx512() : orcv_states_avx2(); }
void orcv_f1600(int isa, uint64_t *st) { (isa == 2 ? orcv_f1600_avx512 : orcv_f1600_avx2)(st); }
