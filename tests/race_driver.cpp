// tests/race_driver.cpp -- the concurrency test of the host runtime, meant to run under ThreadSanitizer / AddressSanitizer
// (`make tsan`, `make asan`; tests/test_gpu_sanitizers.py).  The reference's counterpart is `go test -race`
// (Makefile:44-47 of cloudflare/circl); SURVEY.md section 5 asks for a sanitizer build of the C ABI's host side.
//
// It drives exactly the state the functional tests cannot judge: staging-slot pools, byte-mover pools, the per-device copy /
// compute streams, call_once device tables, shard()'s thread-per-device branch (with CIRCL_HIP_LOGICAL_DEVICES > 1), the
// profiling records, and the error-path Drain guard -- from several caller threads at once.  Results are compared with a
// single-threaded, single-device reference run of the same library, so a race that corrupts data fails even without a
// sanitizer report.  Environment (set by the test): CIRCL_HIP_LOGICAL_DEVICES, CIRCL_HIP_HOST_SLOTS, CIRCL_HIP_HOST_THREADS.
//
//   usage: race_driver [callers=3] [rounds=2] [n_kem=20000] [n_dsa=600]
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "circl_hip.h"

#define CHECK(c)                                                                                                      \
    do {                                                                                                              \
        if (!(c)) {                                                                                                   \
            fprintf(stderr, "%s:%d: check failed: %s (%s)\n", __FILE__, __LINE__, #c, circl_hip_last_error());        \
            exit(1);                                                                                                  \
        }                                                                                                             \
    } while (0)

static std::vector<uint8_t> bytes(size_t n, unsigned seed) {
    std::vector<uint8_t> v(n + 1);  // (one spare byte: the blobs of the Go bridge carry one)
    uint32_t x = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; i++) {
        x = x * 1664525u + 1013904223u;
        v[i] = (uint8_t)(x >> 24);
    }
    return v;
}

struct Kem {
    int param;
    size_t n, EK, DK, CT;
    std::vector<uint8_t> seed, m, ek, dk, ct, ss, ssd, st;
    Kem(int p, size_t n_) : param(p), n(n_) {
        EK = circl_hip_mlkem_ek_size(p); DK = circl_hip_mlkem_dk_size(p); CT = circl_hip_mlkem_ct_size(p);
        seed = bytes(64 * n, 1); m = bytes(32 * n, 2);
        ek.resize(EK * n); dk.resize(DK * n); ct.resize(CT * n); ss.resize(32 * n); ssd.resize(32 * n); st.resize(n);
        CHECK(circl_hip_mlkem_keygen(p, seed.data(), ek.data(), dk.data(), n, 0) == 0);
        CHECK(circl_hip_mlkem_encaps(p, ek.data(), m.data(), ct.data(), ss.data(), st.data(), n, 0) == 0);
        CHECK(circl_hip_mlkem_decaps(p, dk.data(), ct.data(), ssd.data(), st.data(), n, 0) == 0);
        CHECK(ss == ssd);
    }
    // one resident table shared by every caller thread (tables are immutable: concurrent calls are allowed)
    circl_hip_keytable *pub = nullptr, *prv = nullptr;
    std::vector<uint32_t> idx;
    std::vector<uint8_t> ct_t, ss_t;
    void tables() {
        const size_t nk = n < 9 ? n : 9;
        // the public table REPLICATED on every (logical) device -- its calls shard their batch over the replicas from every caller
        // thread at once -- the private one on the last device alone
        CHECK(circl_hip_mlkem_keytable_new(param, 0, ek.data(), nk, CIRCL_HIP_ALL_DEVICES, nullptr, &pub) == 0);
        CHECK(circl_hip_mlkem_keytable_new(param, 1, dk.data(), nk, circl_hip_device_count() - 1, nullptr, &prv) == 0);
        idx.resize(n);
        for (size_t i = 0; i < n; i++) idx[i] = (uint32_t)((i * 7) % nk);
        ct_t.resize(CT * n); ss_t.resize(32 * n);
        std::vector<uint8_t> st_t(n);
        CHECK(circl_hip_mlkem_encaps_table(pub, idx.data(), m.data(), ct_t.data(), ss_t.data(), st_t.data(), n) == 0);
    }
    void again_tables() const {
        std::vector<uint8_t> ct2(CT * n), ss2(32 * n), ss3(32 * n), st2(n);
        CHECK(circl_hip_mlkem_encaps_table(pub, idx.data(), m.data(), ct2.data(), ss2.data(), st2.data(), n) == 0);
        CHECK(ct2 == ct_t && ss2 == ss_t);
        CHECK(circl_hip_mlkem_decaps_table(prv, idx.data(), ct_t.data(), ss3.data(), nullptr, n) == 0);
        CHECK(ss3 == ss_t);
    }
    // many ONE-ITEM calls (and a few of two or three) through the same tables, which coalesce them across the caller threads
    // (circl_hip_keytable_set_coalesce: batches, gates and staging rows shared between threads -- host_coalesce.hip)
    void small_calls(int caller, int count) const {
        uint8_t ct1[3 * 1568], ss1[3 * 32], ss2[3 * 32], st1[3];
        for (int i = 0; i < count; i++) {
            const size_t k = 1 + (size_t)((caller + i) % 3), at = ((size_t)caller * 131 + (size_t)i * 17) % (n - 3);
            CHECK(circl_hip_mlkem_encaps_table(pub, &idx[at], &m[32 * at], ct1, ss1, st1, k) == 0);
            CHECK(!memcmp(ct1, &ct_t[CT * at], CT * k) && !memcmp(ss1, &ss_t[32 * at], 32 * k));
            CHECK(circl_hip_mlkem_decaps_table(prv, &idx[at], &ct_t[CT * at], ss2, i % 2 ? st1 : nullptr, k) == 0);
            CHECK(!memcmp(ss2, &ss_t[32 * at], 32 * k));
            // ... and with keys that come with the call (circl_hip_set_coalesce: process-wide batches per entry point and device)
            CHECK(circl_hip_mlkem_encaps(param, &ek[EK * at], &m[32 * at], ct1, ss1, st1, k, (caller + i) % 2 ? 0 : CIRCL_HIP_ALL_DEVICES) == 0);
            CHECK(!memcmp(ct1, &ct[CT * at], CT * k) && !memcmp(ss1, &ss[32 * at], 32 * k));
            CHECK(circl_hip_mlkem_decaps(param, &dk[DK * at], &ct[CT * at], ss2, nullptr, k, CIRCL_HIP_ALL_DEVICES) == 0);
            CHECK(!memcmp(ss2, &ss[32 * at], 32 * k));
        }
    }
    // The ASYNCHRONOUS form (circl_hip_keytable_async_start / *_table_submit / circl_hip_poll / circl_hip_wait): every caller thread is a
    // reactor that keeps a window of submitted one-to-three-item calls outstanding on tables shared by all of them -- reservation word,
    // records, staging rows, the dispatcher's completion copies into THIS thread's buffers: all cross-thread (host_coalesce.hip) -- and
    // now and then makes a BLOCKING call through the same queue.
    circl_hip_keytable *apub = nullptr, *aprv = nullptr;
    void async_tables() {
        const size_t nk = n < 9 ? n : 9;
        CHECK(circl_hip_mlkem_keytable_new(param, 0, ek.data(), nk, CIRCL_HIP_ALL_DEVICES, nullptr, &apub) == 0);  // one queue + dispatcher per replica
        CHECK(circl_hip_mlkem_keytable_new(param, 1, dk.data(), nk, 0, nullptr, &aprv) == 0);
        CHECK(circl_hip_keytable_async_start(apub, 32, 0, 1) == 0);
        CHECK(circl_hip_keytable_async_start(aprv, 16, 30, 0) == 0);
        uint64_t tk = 0;
        uint8_t junk[1568] = {0};
        CHECK(circl_hip_mlkem_encaps_table_submit(apub, nullptr, nullptr, junk, junk, junk, 1, &tk) == CIRCL_HIP_EPARAM);  // a NULL required input
        CHECK(circl_hip_mlkem_decaps_table_submit(aprv, nullptr, junk, junk, junk, 5, &tk) == CIRCL_HIP_EPARAM);           // more than max_items / 4
    }
    void async_calls(int caller, int count) const {
        constexpr int W = 6;
        struct Slot { uint64_t tk = 0; size_t at = 0, k = 0; bool enc = false, busy = false; uint8_t ct1[3 * 1568], ss1[3 * 32], st1[3]; };
        std::vector<Slot> slots(W);
        int head = 0, tail = 0, outstanding = 0, issued = 0;
        auto reap = [&](bool block) {
            while (outstanding > 0) {
                Slot &sl = slots[head % W];
                const circl_hip_keytable *tab = sl.enc ? apub : aprv;
                int8_t state = 0;
                CHECK(circl_hip_poll(tab, &sl.tk, 1, &state) == (state != 0));
                if (state == 0) {
                    if (!block) return;
                    CHECK(circl_hip_wait(tab, sl.tk, 5000000) == 1);
                } else {
                    CHECK(state == 1);
                }
                if (sl.enc) CHECK(!memcmp(sl.ct1, &ct_t[CT * sl.at], CT * sl.k) && !memcmp(sl.ss1, &ss_t[32 * sl.at], 32 * sl.k) && !sl.st1[0]);
                else CHECK(!memcmp(sl.ss1, &ss_t[32 * sl.at], 32 * sl.k));
                sl.busy = false;
                head++;
                outstanding--;
            }
        };
        while (issued < count) {
            if (outstanding == W) reap(true);
            Slot &sl = slots[tail % W];
            sl.k = 1 + (size_t)((caller + issued) % 3);
            sl.at = ((size_t)caller * 211 + (size_t)issued * 29) % (n - 3);
            sl.enc = (caller + issued) % 2 == 0;
            int rc;
            if (sl.enc) rc = circl_hip_mlkem_encaps_table_submit(apub, &idx[sl.at], &m[32 * sl.at], sl.ct1, sl.ss1, sl.st1, sl.k, &sl.tk);
            else rc = circl_hip_mlkem_decaps_table_submit(aprv, &idx[sl.at], &ct_t[CT * sl.at], sl.ss1, sl.st1, sl.k, &sl.tk);
            if (rc == CIRCL_HIP_EAGAIN) { reap(true); if (!outstanding) std::this_thread::yield(); continue; }
            CHECK(rc == 0);
            sl.busy = true;
            tail++;
            outstanding++;
            issued++;
            reap(false);
            if (issued % 7 == 3) {  // a blocking call through the asynchronous queue (submit + wait inside the library)
                uint8_t ss2[32];
                const size_t at = (sl.at + 1) % (n - 3);
                CHECK(circl_hip_mlkem_decaps_table(aprv, &idx[at], &ct_t[CT * at], ss2, nullptr, 1) == 0);
                CHECK(!memcmp(ss2, &ss_t[32 * at], 32));
            }
        }
        reap(true);
    }
    // circl_hip_queue: the asynchronous form with the key in the call, one queue shared by every caller thread; each keeps two encapsulations outstanding
    circl_hip_queue *cq = nullptr;
    void call_queue() { CHECK(circl_hip_queue_open(CIRCL_HIP_QUEUE_MLKEM_ENCAPS, param, 0, 32, 1, &cq) == 0); }
    void queue_calls(int caller, int count) const {
        uint64_t tk[2] = {0, 0};
        size_t at[2] = {0, 0};
        std::vector<uint8_t> ct1(2 * CT), ss1(2 * 32), st1(2, 9);
        for (int i = 0; i < count + 2; i++) {
            const int sl = i & 1;
            if (i >= 2) {
                CHECK(circl_hip_queue_wait(cq, tk[sl], 5000000) == 1);
                CHECK(!memcmp(&ct1[CT * sl], &ct[CT * at[sl]], CT) && !memcmp(&ss1[32 * sl], &ss[32 * at[sl]], 32) && st1[sl] == 0);
            }
            if (i >= count) continue;
            at[sl] = ((size_t)caller * 97 + (size_t)i * 31) % n;  // item k under ITS OWN key (row k): the answers of the plain batch call in the constructor
            int rc;
            while ((rc = circl_hip_queue_submit(cq, &ek[EK * at[sl]], &m[32 * at[sl]], &ct1[CT * sl], &ss1[32 * sl], &st1[sl], 1, &tk[sl])) == CIRCL_HIP_EAGAIN)
                std::this_thread::yield();
            CHECK(rc == 0);
        }
    }
    // a setter on a table that callers are inside must answer CIRCL_HIP_EBUSY or succeed -- never free under them.
    // `tog` is a third public table; one thread flips its coalescing (and, every few flips, an asynchronous queue) while the others call.
    circl_hip_keytable *tog = nullptr;
    void toggle_table() { CHECK(circl_hip_mlkem_keytable_new(param, 0, ek.data(), n < 9 ? n : 9, 0, nullptr, &tog) == 0); }
    void toggle(int flips, std::atomic<int> &busy, std::atomic<int> &ok) const {
        for (int k = 0; k < flips; k++) {
            int rc;
            if (k % 5 == 4) rc = circl_hip_keytable_async_start(tog, 16, 0, 0);
            else rc = circl_hip_keytable_set_coalesce(tog, k % 2 ? 16 : 0, 0);
            CHECK(rc == 0 || rc == CIRCL_HIP_EBUSY);
            (rc ? busy : ok).fetch_add(1);
            std::this_thread::yield();
        }
    }
    void toggled_calls(int caller, int count) const {
        uint8_t ct1[1568], ss1[32], st1[1];
        for (int i = 0; i < count; i++) {
            const size_t at = ((size_t)caller * 53 + (size_t)i * 13) % (n - 3);
            CHECK(circl_hip_mlkem_encaps_table(tog, &idx[at], &m[32 * at], ct1, ss1, st1, 1) == 0);  // whichever path the table offers right now
            CHECK(!memcmp(ct1, &ct_t[CT * at], CT) && !memcmp(ss1, &ss_t[32 * at], 32));
        }
    }
    void again(int device) const {
        std::vector<uint8_t> ek2(EK * n), dk2(DK * n), ct2(CT * n), ss2(32 * n), ss3(32 * n), st2(n);
        CHECK(circl_hip_mlkem_keygen(param, seed.data(), ek2.data(), dk2.data(), n, device) == 0);
        CHECK(ek2 == ek && dk2 == dk);
        CHECK(circl_hip_mlkem_encaps(param, ek.data(), m.data(), ct2.data(), ss2.data(), st2.data(), n, device) == 0);
        CHECK(ct2 == ct && ss2 == ss);
        CHECK(circl_hip_mlkem_decaps(param, dk.data(), ct.data(), ss3.data(), nullptr, n, device) == 0);
        CHECK(ss3 == ss);
    }
};

struct Dsa {
    int param;
    size_t n, PK, SK, SIG;
    std::vector<uint8_t> seed, pk, sk, sig, mblob, cblob;
    std::vector<uint64_t> moff, coff;
    Dsa(int p, size_t n_) : param(p), n(n_) {
        PK = circl_hip_mldsa_pk_size(p); SK = circl_hip_mldsa_sk_size(p); SIG = circl_hip_mldsa_sig_size(p);
        seed = bytes(32 * n, 3);
        pk.resize(PK * n); sk.resize(SK * n); sig.resize(SIG * n + 4);
        moff.resize(n + 1); coff.resize(n + 1);
        size_t mt = 0, ctot = 0;
        for (size_t i = 0; i < n; i++) { moff[i] = mt; coff[i] = ctot; mt += 1 + (i * 37) % 300; ctot += i % 11; }
        moff[n] = mt; coff[n] = ctot;
        mblob = bytes(mt, 4); cblob = bytes(ctot, 5);
        CHECK(circl_hip_mldsa_keygen(p, seed.data(), pk.data(), sk.data(), n, 0) == 0);
        CHECK(circl_hip_mldsa_sign(p, sk.data(), mblob.data(), moff.data(), cblob.data(), coff.data(), nullptr, sig.data(), n, 0) == 0);
    }
    // several prepared private keys (replicated) and their public keys, shared by every caller thread: message i is signed with
    // entry i mod nk (the signer's per-call key index travels in a thread-local of the library: concurrent calls must not mix theirs)
    circl_hip_keytable *signer = nullptr, *verifier = nullptr;
    std::vector<uint32_t> kidx;
    std::vector<uint8_t> sig_t;
    void tables() {
        const size_t nk = n < 5 ? n : 5;
        CHECK(circl_hip_mldsa_privkeys_new(param, sk.data(), nk, CIRCL_HIP_ALL_DEVICES, &signer) == 0);
        CHECK(circl_hip_mldsa_keytable_new(param, pk.data(), nk, CIRCL_HIP_ALL_DEVICES, &verifier) == 0);
        kidx.resize(n);
        for (size_t i = 0; i < n; i++) kidx[i] = (uint32_t)((i * 3) % nk);
        sig_t.resize(SIG * n + 4);
        CHECK(circl_hip_mldsa_sign_table_keyed(signer, kidx.data(), mblob.data(), moff.data(), cblob.data(), coff.data(), nullptr, sig_t.data(), n) == 0);
    }
    void again_tables() const {
        std::vector<uint8_t> sig2(SIG * n + 4), ok(n);
        CHECK(circl_hip_mldsa_sign_table_keyed(signer, kidx.data(), mblob.data(), moff.data(), cblob.data(), coff.data(), nullptr, sig2.data(), n) == 0);
        CHECK(memcmp(sig2.data(), sig_t.data(), SIG * n) == 0);
        CHECK(circl_hip_mldsa_verify_table(verifier, kidx.data(), sig2.data(), mblob.data(), moff.data(), cblob.data(), coff.data(), ok.data(), n) == 0);
        for (size_t i = 0; i < n; i++) CHECK(ok[i] == 1);
    }
    void small_calls(int caller, int count) const {  // one signature per call, ragged messages and contexts, through the coalescing verifier table
        for (int i = 0; i < count; i++) {
            const size_t at = ((size_t)caller * 37 + (size_t)i * 11) % n;
            uint8_t ok1 = 9;
            CHECK(circl_hip_mldsa_verify_table(verifier, &kidx[at], &sig_t[SIG * at], mblob.data(), &moff[at], cblob.data(), &coff[at], &ok1, 1) == 0);
            CHECK(ok1 == 1);
        }
    }
    circl_hip_keytable *averifier = nullptr;
    void async_tables() {
        CHECK(circl_hip_mldsa_keytable_new(param, pk.data(), n < 5 ? n : 5, 0, &averifier) == 0);
        CHECK(circl_hip_keytable_async_start(averifier, 8, 0, 0) == 0);
    }
    void async_calls(int caller, int count) const {  // two submitted verifications outstanding, ragged messages and contexts
        uint64_t tk[2] = {0, 0};
        uint8_t ok2[2] = {9, 9};
        for (int i = 0; i < count; i++) {
            const int s = i & 1;
            if (i >= 2) { CHECK(circl_hip_wait(averifier, tk[s], 5000000) == 1); CHECK(ok2[s] == 1); }
            const size_t at = ((size_t)caller * 41 + (size_t)i * 7) % n;
            ok2[s] = 9;
            int rc;
            while ((rc = circl_hip_mldsa_verify_table_submit(averifier, &kidx[at], &sig_t[SIG * at], mblob.data(), &moff[at], cblob.data(), &coff[at], &ok2[s], 1, &tk[s])) ==
                   CIRCL_HIP_EAGAIN)
                std::this_thread::yield();
            CHECK(rc == 0);
        }
        for (int s = 0; s < 2 && s < count; s++) { CHECK(circl_hip_wait(averifier, tk[s], 5000000) == 1); CHECK(ok2[s] == 1); }
    }
    void again(int device) const {
        std::vector<uint8_t> sig2(SIG * n + 4), ok(n);
        CHECK(circl_hip_mldsa_sign(param, sk.data(), mblob.data(), moff.data(), cblob.data(), coff.data(), nullptr, sig2.data(), n, device) == 0);
        CHECK(memcmp(sig2.data(), sig.data(), SIG * n) == 0);
        sig2[SIG * (n / 3) + 50] ^= 2;
        CHECK(circl_hip_mldsa_verify(param, pk.data(), sig2.data(), mblob.data(), moff.data(), cblob.data(), coff.data(), ok.data(), n, device) == 0);
        for (size_t i = 0; i < n; i++) CHECK(ok[i] == (i == n / 3 ? 0 : 1));
    }
};

struct Hyb {
    int scheme;
    size_t n, SEED, ES, PK, SK, CT, SS;
    std::vector<uint8_t> seed, es, pk, sk, ct, ss;
    Hyb(int s, size_t n_) : scheme(s), n(n_) {
        SEED = circl_hip_hybrid_seed_size(s); ES = circl_hip_hybrid_eseed_size(s); PK = circl_hip_hybrid_pk_size(s);
        SK = circl_hip_hybrid_sk_size(s); CT = circl_hip_hybrid_ct_size(s); SS = circl_hip_hybrid_ss_size(s);
        seed = bytes(SEED * n, 6); es = bytes(ES * n, 7);
        pk.resize(PK * n); sk.resize(SK * n); ct.resize(CT * n); ss.resize(SS * n);
        std::vector<uint8_t> st(n);
        CHECK(circl_hip_hybrid_keygen(s, seed.data(), pk.data(), sk.data(), n, 0) == 0);
        CHECK(circl_hip_hybrid_encaps(s, pk.data(), es.data(), ct.data(), ss.data(), st.data(), n, 0) == 0);
    }
    // a resident hybrid table with an asynchronous queue, shared by every caller thread (one hybrid launch = an X25519 ladder: the calls
    // that share it gain most); each caller keeps two submitted encapsulations outstanding
    circl_hip_keytable *apub = nullptr;
    void async_table() {
        CHECK(circl_hip_hybrid_keytable_new(scheme, 0, pk.data(), n < 4 ? n : 4, 0, nullptr, &apub) == 0);
        CHECK(circl_hip_keytable_async_start(apub, 16, 0, 0) == 0);
    }
    void async_calls(int caller, int count) const {
        const size_t nk = n < 4 ? n : 4;
        uint64_t tk[2] = {0, 0};
        size_t at[2] = {0, 0};
        std::vector<uint8_t> ct1(2 * CT), ss1(2 * SS), st1(2, 9);
        for (int i = 0; i < count + 2; i++) {
            const int sl = i & 1;
            if (i >= 2) {
                CHECK(circl_hip_wait(apub, tk[sl], 5000000) == 1);
                CHECK(!memcmp(&ct1[CT * sl], &ct[CT * at[sl]], CT) && !memcmp(&ss1[SS * sl], &ss[SS * at[sl]], SS) && st1[sl] == 0);
            }
            if (i >= count) continue;
            at[sl] = ((size_t)caller * 5 + (size_t)i) % nk;  // item k encapsulates to ITS OWN key (entry k): the answers of the plain batch call above
            const uint32_t ki = (uint32_t)at[sl];
            int rc;
            while ((rc = circl_hip_hybrid_encaps_table_submit(apub, &ki, &es[ES * at[sl]], &ct1[CT * sl], &ss1[SS * sl], &st1[sl], 1, &tk[sl])) == CIRCL_HIP_EAGAIN)
                std::this_thread::yield();
            CHECK(rc == 0);
        }
    }
    void again(int device) const {
        std::vector<uint8_t> ct2(CT * n), ss2(SS * n), ss3(SS * n), st(n);
        CHECK(circl_hip_hybrid_encaps(scheme, pk.data(), es.data(), ct2.data(), ss2.data(), st.data(), n, device) == 0);
        CHECK(ct2 == ct && ss2 == ss);
        CHECK(circl_hip_hybrid_decaps(scheme, sk.data(), ct.data(), ss3.data(), st.data(), n, device) == 0);
        CHECK(ss3 == ss);
    }
};

int main(int argc, char **argv) {
    const int callers = argc > 1 ? atoi(argv[1]) : 3;
    const int rounds = argc > 2 ? atoi(argv[2]) : 2;
    const size_t n_kem = argc > 3 ? (size_t)atol(argv[3]) : 20000, n_dsa = argc > 4 ? (size_t)atol(argv[4]) : 600;
    const int nd = circl_hip_init();
    if (nd <= 0) { fprintf(stderr, "race_driver: no HIP device\n"); return 2; }
    printf("race_driver: %d (logical) devices, %d callers x %d rounds, %zu ML-KEM items, %zu ML-DSA items\n", nd, callers, rounds, n_kem, n_dsa);
    Kem kem768(768, n_kem), kem1024(1024, n_kem / 4 + 3);
    kem768.tables();
    Dsa dsa65(65, n_dsa), dsa44(44, 9);  // 9 items: shards below the 16-item switch of the signer, and empty shards
    dsa65.tables();
    // small calls through these tables share launches across the caller threads (the big calls below still take the ordinary path)
    CHECK(circl_hip_keytable_set_coalesce(kem768.pub, 16, 0) == 0);
    CHECK(circl_hip_keytable_set_coalesce(kem768.prv, 64, 50) == 0);
    CHECK(circl_hip_keytable_set_coalesce(dsa65.verifier, 8, 0) == 0);
    CHECK(circl_hip_set_coalesce(16, 0) == 0);
    kem768.async_tables();
    kem768.call_queue();
    dsa65.async_tables();
    kem768.toggle_table();
    std::atomic<int> tog_busy{0}, tog_ok{0};
    Hyb xwing(1, n_kem / 8 + 5);
    xwing.async_table();
    circl_hip_profile_enable(1);  // the profiling records are shared state too
    std::atomic<int> started{0};
    std::vector<std::thread> th;
    for (int c = 0; c < callers; c++) {
        th.emplace_back([&, c] {
            started.fetch_add(1);
            while (started.load() < callers) std::this_thread::yield();  // everybody enters the library together
            for (int r = 0; r < rounds; r++) {
                const int device = (c + r) % 3 == 2 ? nd - 1 : CIRCL_HIP_ALL_DEVICES;  // mostly all devices; now and then one device alone
                switch ((c + r) % 4) {
                case 0: kem768.again(device); kem768.again_tables(); break;
                case 1: dsa65.again(device); dsa44.again(CIRCL_HIP_ALL_DEVICES); dsa65.again_tables(); break;
                case 2: kem1024.again(device); xwing.again(device); break;
                case 3: kem768.again(device); dsa44.again(device); kem768.again_tables(); break;
                }
                kem768.small_calls(c, 24);
                dsa65.small_calls(c, 6);
                kem768.async_calls(c, 40);
                dsa65.async_calls(c, 5);
                xwing.async_calls(c, 4);
                kem768.queue_calls(c, 12);
                if (c == 0) kem768.toggle(60, tog_busy, tog_ok);  // ... while the other callers are inside the toggled table
                else kem768.toggled_calls(c, 30);
                // an error path in the middle of everything: the Drain guard must give its slots back
                uint8_t junk[64] = {0};
                CHECK(circl_hip_mlkem_encaps(768, nullptr, junk, junk, junk, junk, 1, 0) == CIRCL_HIP_EPARAM);
                CHECK(circl_hip_mlkem_encaps(768, junk, junk, junk, junk, junk, 1, nd) == CIRCL_HIP_ENODEV);
            }
        });
    }
    for (auto &t : th) t.join();
    // the asynchronous queues: a ticket left outstanding is finished by the close (the dispatcher drains before it leaves)
    {
        uint64_t tk = 0;
        uint8_t ct1[1568], ss1[32], st1[1] = {9};
        CHECK(circl_hip_mlkem_encaps_table_submit(kem768.apub, &kem768.idx[1], &kem768.m[32], ct1, ss1, st1, 1, &tk) == 0);
        CHECK(circl_hip_keytable_eventfd(kem768.apub, 0) >= 0 && circl_hip_keytable_eventfd(kem768.aprv, 0) == -1);
        CHECK(circl_hip_keytable_close(kem768.apub) == 0);
        CHECK(!memcmp(ct1, &kem768.ct_t[kem768.CT], kem768.CT) && st1[0] == 0);
    }
    CHECK(circl_hip_keytable_close(kem768.aprv) == 0);
    CHECK(circl_hip_keytable_close(dsa65.averifier) == 0);
    CHECK(circl_hip_keytable_close(kem768.tog) == 0);
    CHECK(circl_hip_keytable_close(xwing.apub) == 0);
    CHECK(circl_hip_queue_eventfd(kem768.cq) >= 0 && circl_hip_queue_close(kem768.cq) == 0);
    CHECK(circl_hip_set_coalesce(0, 0) == 0);  // off AND drained: nobody is inside a process-wide batch any more
    printf("race_driver: the toggled table answered %d setters with OK, %d with EBUSY\n", tog_ok.load(), tog_busy.load());
    circl_hip_keytable_free(kem768.pub);
    circl_hip_keytable_free(kem768.prv);
    circl_hip_keytable_free(dsa65.signer);
    circl_hip_keytable_free(dsa65.verifier);
    double ms = 0;
    uint64_t launches = 0;
    for (int k = 0; k < CIRCL_HIP_KERNEL_COUNT; k++) {
        double m1 = 0;
        uint64_t l1 = 0;
        CHECK(circl_hip_profile_read(k, &m1, &l1) == 0);
        ms += m1;
        launches += l1;
    }
    circl_hip_profile_enable(0);
    printf("race_driver ok: %llu profiled launch groups, %.1f ms of kernels\n", (unsigned long long)launches, ms);
    return 0;
}
