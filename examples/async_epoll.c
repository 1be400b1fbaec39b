/* examples/async_epoll.c -- the asynchronous table API from an event loop: ONE thread, an epoll set, no thread ever asleep inside the library.
 * A decapsulation server's shape (the caller of kem.Scheme.Decapsulate in hpke/algs.go:283-285): the private key is parsed once into a resident
 * table, the table gets a queue (circl_hip_keytable_async_start) whose eventfd sits in the loop's epoll set next to the sockets a real server
 * would have there; requests are submitted as they arrive (here: a timerfd stands in for the network, a burst of ciphertexts per tick) and reaped
 * when the eventfd says a batch is done.  Tickets of a queue finish in issue order, so the loop keeps them in a ring and only looks at its head.
 * Build (after `make lib`):
 *   gcc -O2 -Iinclude examples/async_epoll.c -Lcircl_amd -lcirclhip -Wl,-rpath,$PWD/circl_amd -Wl,-rpath,/opt/rocm/lib -o build/async_epoll */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/epoll.h>
#include <sys/timerfd.h>
#include <time.h>
#include <unistd.h>

#include "circl_hip.h"

enum { EK = 1184, DK = 2400, CT = 1088, POOL = 256, WINDOW = 512, TOTAL = 20000, BURST = 24 };

static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

int main(void) {
    if (circl_hip_init() <= 0) { fprintf(stderr, "no HIP device: %s\n", circl_hip_last_error()); return 2; }
    static uint8_t seed[64], ek[EK], dk[DK], m[32 * POOL], ct[CT * POOL], ss[32 * POOL], st[POOL];
    for (size_t i = 0; i < sizeof seed; i++) seed[i] = (uint8_t)(5 * i + 9);
    for (size_t i = 0; i < sizeof m; i++) m[i] = (uint8_t)(i * 31 + (i >> 6));
    int rc = circl_hip_mlkem_keygen(768, seed, ek, dk, 1, 0);
    if (!rc) rc = circl_hip_mlkem_encaps_shared(768, ek, m, ct, ss, st, POOL, 0); /* the clients' ciphertexts and the answers */
    circl_hip_keytable *prv = NULL;
    uint8_t verdict = 0;
    if (!rc) rc = circl_hip_mlkem_keytable_new(768, 1, dk, 1, 0, &verdict, &prv);
    if (!rc) rc = circl_hip_keytable_async_start(prv, 1024, 0, /*want_eventfd=*/1);
    if (rc || verdict) { fprintf(stderr, "setup failed: %d / %d: %s\n", rc, verdict, circl_hip_last_error()); return 1; }
    const int efd = circl_hip_keytable_eventfd(prv, 0), tfd = timerfd_create(CLOCK_MONOTONIC, TFD_NONBLOCK), ep = epoll_create1(0);
    struct itimerspec every = {{0, 50000}, {0, 50000}}; /* "the network": a burst of requests every 50 us */
    timerfd_settime(tfd, 0, &every, NULL);
    struct epoll_event ev = {EPOLLIN, {.fd = efd}};
    epoll_ctl(ep, EPOLL_CTL_ADD, efd, &ev);
    ev.data.fd = tfd;
    epoll_ctl(ep, EPOLL_CTL_ADD, tfd, &ev);

    /* requests in flight: a ring of (ticket, which ciphertext, when, where its shared secret lands) */
    static uint64_t ticket[WINDOW];
    static size_t which[WINDOW];
    static double t0[WINDOW];
    static uint8_t out_ss[32 * WINDOW], out_st[WINDOW];
    size_t head = 0, tail = 0, submitted = 0, finished = 0, mismatches = 0, again = 0;
    double lat_sum = 0, lat_max = 0;
    const double t_begin = now_us();
    while (finished < TOTAL) {
        struct epoll_event got[4];
        const int ng = epoll_wait(ep, got, 4, 100);
        for (int g = 0; g < ng; g++) {
            uint64_t cnt = 0;
            if (read(got[g].data.fd, &cnt, sizeof cnt) != (ssize_t)sizeof cnt) continue; /* (both are counters: drained by the read) */
            if (got[g].data.fd == tfd) { /* requests arrived: submit them, one call each (inputs are copied before the call returns) */
                for (int b = 0; b < BURST && submitted < TOTAL && tail - head < WINDOW; b++) {
                    const size_t sl = tail % WINDOW, i = (submitted * 37 + 11) % POOL;
                    rc = circl_hip_mlkem_decaps_table_submit(prv, NULL, ct + CT * i, out_ss + 32 * sl, out_st + sl, 1, &ticket[sl]);
                    if (rc == CIRCL_HIP_EAGAIN) { again++; break; } /* every device batch busy: the next completion makes room */
                    if (rc) { fprintf(stderr, "submit failed: %d %s\n", rc, circl_hip_last_error()); return 1; }
                    which[sl] = i;
                    t0[sl] = now_us();
                    tail++;
                    submitted++;
                }
            }
        }
        /* reap: the head of the ring, as long as it is done (one atomic load per look) */
        while (head != tail) {
            const size_t sl = head % WINDOW;
            int8_t state = 0;
            circl_hip_poll(prv, &ticket[sl], 1, &state);
            if (state == 0) break;
            if (state != 1) { fprintf(stderr, "batch failed: %d\n", state); return 1; }
            if (out_st[sl] != 0 || memcmp(out_ss + 32 * sl, ss + 32 * which[sl], 32)) mismatches++;
            const double l = now_us() - t0[sl];
            lat_sum += l;
            if (l > lat_max) lat_max = l;
            head++;
            finished++;
        }
    }
    const double el = now_us() - t_begin;
    uint64_t calls = 0, items = 0, launches = 0;
    circl_hip_keytable_coalesce_stats(prv, &calls, &items, &launches);
    printf("%d one-item decapsulations through ONE event-loop thread: %.0f/s, mean latency %.0f us (max %.0f), %.1f items per launch, %zu EAGAIN, mismatches %zu\n",
           TOTAL, TOTAL / (el * 1e-6), lat_sum / TOTAL, lat_max, launches ? (double)items / launches : 0.0, again, mismatches);
    rc = circl_hip_keytable_close(prv); /* CIRCL_HIP_EBUSY would mean a call is still inside the table */
    close(tfd);
    close(ep);
    return rc || mismatches ? 1 : 0;
}
