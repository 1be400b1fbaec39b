// tools/concurrent_bench.cpp -- not a test: what MANY concurrent small callers get from one resident key table.
//
// The shape of every consumer of the reference's kem.Scheme / sign.Scheme (kem/hybrid/hybrid.go:95-99, hpke/algs.go:283-285,
// kem/mlkem/mlkem768/kyber.go:347-386): T host threads, each looping calls of `items` items (default 1) through host buffers --
// circl_hip_mlkem_encaps_table / circl_hip_mlkem_decaps_table / circl_hip_mldsa_verify_table -- closed loop (a thread issues its next
// call when the previous one has returned, so aggregate <= T / latency: Little's law).  Every result is compared with the answer of
// ONE ordinary batch call made beforehand (bit-exact under concurrency, whatever batches the calls ended up in).
//
//   concurrent_bench <op: encaps|decaps|verify|sign|encaps_item> <coalesce max_items (0 = off)> <max_wait_us> <items per call> <seconds> <T> [T ...]
//   concurrent_bench --async <op: encaps|decaps|verify|encaps_call|decaps_call> <R reactor threads> <W requests outstanding per reactor> <seconds> [max_items] [items per request]
//       the ASYNCHRONOUS form (circl_hip_keytable_async_start / *_table_submit / circl_hip_poll / circl_hip_wait): every reactor keeps W one-item
//       requests outstanding -- submit until the window is full, poll the oldest tickets, block in circl_hip_wait only when nothing moved --
//       the shape of ONE goroutine (or one epoll loop) per device serving every connection's handshake.  Latency = submit -> seen done.
//   concurrent_bench --one-call [calls]
//       ONE blocking caller, one-item resident-key calls through a coalescing table, with the library's own time stamps
//       (circl_hip_profile_call_stamps): where the microseconds between call and return go, for each completion mode (CIRCL_HIP_COALESCE_DONE).
//
// Output: one line per T: aggregate ops/s, p50 / p99 / max latency of a call (us), calls per launch when coalescing.
// Built by tools/build_tools.sh (g++ against libcirclhip.so); profiles/r05_concurrent.txt is its output on one MI355X.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>
#include <sys/resource.h>

#include <deque>

#include "circl_hip.h"

#define CHECK(c)                                                                                               \
    do {                                                                                                       \
        if (!(c)) {                                                                                            \
            fprintf(stderr, "%s:%d: check failed: %s (%s)\n", __FILE__, __LINE__, #c, circl_hip_last_error()); \
            exit(1);                                                                                           \
        }                                                                                                      \
    } while (0)

static std::vector<uint8_t> bytes(size_t n, unsigned seed) {
    std::vector<uint8_t> v(n + 16);
    uint32_t x = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; i++) {
        x = x * 1664525u + 1013904223u;
        v[i] = (uint8_t)(x >> 24);
    }
    return v;
}
using Clock = std::chrono::steady_clock;

// CPU time of the whole process and what the container's CPU quota did to it (cgroup v2 cpu.stat; zeros where there is none)
struct CpuStat { double usage_us = 0, throttled_us = 0; long nr_throttled = 0; };
static CpuStat cpu_stat() {
    CpuStat c;
    FILE *f = fopen("/sys/fs/cgroup/cpu.stat", "r");
    if (!f) return c;
    char key[64];
    double v;
    while (fscanf(f, "%63s %lf", key, &v) == 2) {
        if (!strcmp(key, "usage_usec")) c.usage_us = v;
        else if (!strcmp(key, "throttled_usec")) c.throttled_us = v;
        else if (!strcmp(key, "nr_throttled")) c.nr_throttled = (long)v;
    }
    fclose(f);
    return c;
}


// ---- shared set-up: a table of NK resident keys, POOL items with known answers (made by ONE ordinary batch call) ----
struct Work {
    static constexpr size_t NK = 8, POOL = 4096, MSG = 32;
    static constexpr int kem = 768, dsa = 65;
    size_t EK, DK, CT, PK, SK, SIG;
    std::vector<uint32_t> kidx;
    circl_hip_keytable *table = nullptr;
    circl_hip_queue *queue = nullptr;   // ops "encaps_call" / "decaps_call": the key comes with every item (circl_hip_queue)
    std::vector<uint8_t> ekeys, dkeys;  // ... NK key rows each
    std::vector<uint8_t> m, ct, ss, st, sig, ok, mblob;
    std::vector<uint64_t> moff;
    void make(const std::string &op) {
        EK = circl_hip_mlkem_ek_size(kem); DK = circl_hip_mlkem_dk_size(kem); CT = circl_hip_mlkem_ct_size(kem);
        PK = circl_hip_mldsa_pk_size(dsa); SK = circl_hip_mldsa_sk_size(dsa); SIG = circl_hip_mldsa_sig_size(dsa);
        kidx.resize(POOL);
        for (size_t i = 0; i < POOL; i++) kidx[i] = (uint32_t)((i * 5 + i / 7) % NK);
        if (op == "encaps_call" || op == "decaps_call") {
            std::vector<uint8_t> seed = bytes(64 * NK, 1);
            ekeys.resize(EK * NK); dkeys.resize(DK * NK);
            CHECK(circl_hip_mlkem_keygen(kem, seed.data(), ekeys.data(), dkeys.data(), NK, 0) == 0);
            m = bytes(32 * POOL, 2);
            ct.resize(CT * POOL); ss.resize(32 * POOL); st.resize(POOL);
            circl_hip_keytable *pub = nullptr;
            CHECK(circl_hip_mlkem_keytable_new(kem, 0, ekeys.data(), NK, 0, nullptr, &pub) == 0);
            CHECK(circl_hip_mlkem_encaps_table(pub, kidx.data(), m.data(), ct.data(), ss.data(), st.data(), POOL) == 0);  // the answers
            circl_hip_keytable_free(pub);
        } else if (op == "encaps" || op == "decaps") {
            std::vector<uint8_t> seed = bytes(64 * NK, 1), ek(EK * NK), dk(DK * NK);
            CHECK(circl_hip_mlkem_keygen(kem, seed.data(), ek.data(), dk.data(), NK, 0) == 0);
            m = bytes(32 * POOL, 2);
            ct.resize(CT * POOL); ss.resize(32 * POOL); st.resize(POOL);
            circl_hip_keytable *pub = nullptr;
            CHECK(circl_hip_mlkem_keytable_new(kem, 0, ek.data(), NK, 0, nullptr, &pub) == 0);
            CHECK(circl_hip_mlkem_encaps_table(pub, kidx.data(), m.data(), ct.data(), ss.data(), st.data(), POOL) == 0);
            if (op == "encaps") table = pub;
            else {
                circl_hip_keytable_free(pub);
                CHECK(circl_hip_mlkem_keytable_new(kem, 1, dk.data(), NK, 0, nullptr, &table) == 0);
            }
        } else if (op == "verify") {
            std::vector<uint8_t> seed = bytes(32 * NK, 3), pk(PK * NK), sk(SK * NK);
            CHECK(circl_hip_mldsa_keygen(dsa, seed.data(), pk.data(), sk.data(), NK, 0) == 0);
            circl_hip_keytable *signer = nullptr;
            CHECK(circl_hip_mldsa_privkeys_new(dsa, sk.data(), NK, 0, &signer) == 0);
            mblob = bytes(MSG * POOL, 4);
            moff.resize(POOL + 1);
            for (size_t i = 0; i <= POOL; i++) moff[i] = MSG * i;
            sig.resize(SIG * POOL + 16);
            CHECK(circl_hip_mldsa_sign_table_keyed(signer, kidx.data(), mblob.data(), moff.data(), nullptr, nullptr, nullptr, sig.data(), POOL) == 0);
            circl_hip_keytable_free(signer);
            for (size_t i = 0; i < POOL; i += 5) sig[SIG * i + 40 + (i % 64)] ^= 1;
            ok.resize(POOL);
            CHECK(circl_hip_mldsa_keytable_new(dsa, pk.data(), NK, 0, &table) == 0);
            CHECK(circl_hip_mldsa_verify_table(table, kidx.data(), sig.data(), mblob.data(), moff.data(), nullptr, nullptr, ok.data(), POOL) == 0);
        } else {
            fprintf(stderr, "unknown op %s\n", op.c_str());
            exit(2);
        }
    }
};

static double us_of(const timeval &a, const timeval &b) { return (double)(a.tv_sec - b.tv_sec) * 1e6 + (a.tv_usec - b.tv_usec); }

// CB_PIN=1: the reactor threads stay on the CPUs of the device's NUMA node (where the library's dispatcher and its page-locked staging live) --
// what a server that cares about its tail does; the default leaves them where the scheduler puts them
static void pin_to_device_node(int device) {
    const char *e = getenv("CB_PIN");
    if (!e || atoi(e) == 0) return;
    int cus = 0, node = -1;
    if (circl_hip_device_info(device, &cus, &node) != 0 || node < 0) return;
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return;
    cpu_set_t set, allowed;
    CPU_ZERO(&set);
    sched_getaffinity(0, sizeof allowed, &allowed);
    int a = 0, b = 0;
    char sep = 0;
    while (fscanf(f, "%d", &a) == 1) {
        b = a;
        if (fscanf(f, "%c", &sep) == 1 && sep == '-') { if (fscanf(f, "%d", &b) != 1) b = a; if (fscanf(f, "%c", &sep) != 1) sep = 0; }
        for (int c = a; c <= b; c++) if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) CPU_SET(c, &set);
        if (sep != ',') break;
    }
    fclose(f);
    if (CPU_COUNT(&set) > 0) sched_setaffinity(0, sizeof set, &set);
}

// ---- --async: R reactors x W outstanding requests ----
static int async_main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s --async <encaps|decaps|verify> <R> <W> <seconds> [max_items] [items per request]\n", argv[0]); return 2; }
    const std::string op = argv[2];
    const int R = atoi(argv[3]);
    const size_t W = (size_t)atol(argv[4]);
    const double seconds = atof(argv[5]);
    const size_t max_items = argc > 6 ? (size_t)atol(argv[6]) : 2048;
    const size_t per = argc > 7 ? (size_t)std::max(1L, atol(argv[7])) : 1;
    CHECK(circl_hip_init() > 0);
    Work w;
    w.make(op);
    const bool by_call = op == "encaps_call" || op == "decaps_call";
    if (by_call) CHECK(circl_hip_queue_open(op == "encaps_call" ? CIRCL_HIP_QUEUE_MLKEM_ENCAPS : CIRCL_HIP_QUEUE_MLKEM_DECAPS, Work::kem, 0, max_items, 0, &w.queue) == 0);
    else CHECK(circl_hip_keytable_async_start(w.table, max_items, 0, 0) == 0);
    const size_t CT = w.CT, SIG = w.SIG, POOL = Work::POOL, EK = w.EK, DK = w.DK;
    CHECK(!by_call || per == 1);  // (one key row per item: the pool's answers are per (item, key index))
    struct Slot { uint64_t ticket; size_t at; Clock::time_point t0; std::vector<uint8_t> o_ct, o_ss, o_st; };
    std::atomic<int> started{0};
    std::atomic<bool> stop{false};
    std::vector<std::vector<float>> lat(R);
    std::vector<uint64_t> done(R, 0), again(R, 0), waits(R, 0);
    std::atomic<uint64_t> mismatches{0}, thr_user_us{0}, thr_sys_us{0};
    rusage ru0{};
    getrusage(RUSAGE_SELF, &ru0);
    uint64_t c0 = 0, i0 = 0, l0 = 0;
    if (by_call) circl_hip_queue_stats(w.queue, &c0, &i0, &l0);
    else circl_hip_keytable_coalesce_stats(w.table, &c0, &i0, &l0);
    std::vector<std::thread> th;
    for (int t = 0; t < R; t++) {
        th.emplace_back([&, t] {
            pin_to_device_node(0);
            std::vector<Slot> slots(W);
            for (auto &sl : slots) { sl.o_ct.resize(CT * per); sl.o_ss.resize(32 * per); sl.o_st.resize(per); }
            std::deque<size_t> fifo, freel;  // outstanding (oldest first) / free slot indices
            for (size_t i = 0; i < W; i++) freel.push_back(i);
            lat[t].reserve(1 << 18);
            size_t at = ((size_t)t * 997) % (POOL - per);
            started.fetch_add(1);
            while (started.load() < R + 1) std::this_thread::yield();
            auto drain_one = [&](Slot &sl) {
                bool good;
                if (op == "encaps" || op == "encaps_call") good = !memcmp(sl.o_ct.data(), &w.ct[CT * sl.at], CT * per) && !memcmp(sl.o_ss.data(), &w.ss[32 * sl.at], 32 * per);
                else if (op == "decaps" || op == "decaps_call") good = !memcmp(sl.o_ss.data(), &w.ss[32 * sl.at], 32 * per);
                else good = !memcmp(sl.o_st.data(), &w.ok[sl.at], per);
                if (!good) mismatches.fetch_add(1);
                if (lat[t].size() < lat[t].capacity()) lat[t].push_back(std::chrono::duration<float, std::micro>(Clock::now() - sl.t0).count());
                done[t] += per;
            };
            while (!stop.load(std::memory_order_relaxed) || !fifo.empty()) {
                bool moved = false;
                while (!freel.empty() && !stop.load(std::memory_order_relaxed)) {
                    Slot &sl = slots[freel.front()];
                    sl.at = at;
                    sl.t0 = Clock::now();
                    int rc;
                    if (op == "encaps_call") rc = circl_hip_queue_submit(w.queue, &w.ekeys[EK * w.kidx[at]], &w.m[32 * at], sl.o_ct.data(), sl.o_ss.data(), sl.o_st.data(), 1, &sl.ticket);
                    else if (op == "decaps_call") rc = circl_hip_queue_submit(w.queue, &w.dkeys[DK * w.kidx[at]], &w.ct[CT * at], nullptr, sl.o_ss.data(), sl.o_st.data(), 1, &sl.ticket);
                    else if (op == "encaps") rc = circl_hip_mlkem_encaps_table_submit(w.table, &w.kidx[at], &w.m[32 * at], sl.o_ct.data(), sl.o_ss.data(), sl.o_st.data(), per, &sl.ticket);
                    else if (op == "decaps") rc = circl_hip_mlkem_decaps_table_submit(w.table, &w.kidx[at], &w.ct[CT * at], sl.o_ss.data(), sl.o_st.data(), per, &sl.ticket);
                    else rc = circl_hip_mldsa_verify_table_submit(w.table, &w.kidx[at], &w.sig[SIG * at], w.mblob.data(), &w.moff[at], nullptr, nullptr, sl.o_st.data(), per, &sl.ticket);
                    if (rc == CIRCL_HIP_EAGAIN) { again[t]++; break; }
                    CHECK(rc == 0);
                    fifo.push_back(freel.front());
                    freel.pop_front();
                    at = (at + per * 131 + 1) % (POOL - per);
                    moved = true;
                }
                while (!fifo.empty()) {  // tickets of one queue complete in issue order: only the head needs a look
                    int8_t state = 0;
                    if (by_call) circl_hip_queue_poll(w.queue, &slots[fifo.front()].ticket, 1, &state);
                    else circl_hip_poll(w.table, &slots[fifo.front()].ticket, 1, &state);
                    if (state == 0) break;
                    CHECK(state == 1);
                    drain_one(slots[fifo.front()]);
                    freel.push_back(fifo.front());
                    fifo.pop_front();
                    moved = true;
                }
                if (!moved && !fifo.empty()) {  // nothing to submit, nothing finished: park this ONE thread until the oldest ticket is done
                    waits[t]++;
                    if (by_call) (void)circl_hip_queue_wait(w.queue, slots[fifo.front()].ticket, 200);
                    else (void)circl_hip_wait(w.table, slots[fifo.front()].ticket, 200);
                }
            }
            rusage ru{};
            if (getrusage(RUSAGE_THREAD, &ru) == 0) {
                thr_user_us.fetch_add((uint64_t)ru.ru_utime.tv_sec * 1000000 + ru.ru_utime.tv_usec);
                thr_sys_us.fetch_add((uint64_t)ru.ru_stime.tv_sec * 1000000 + ru.ru_stime.tv_usec);
            }
        });
    }
    while (started.load() < R) std::this_thread::yield();
    const CpuStat cs0 = cpu_stat();
    const auto t_begin = Clock::now();
    started.fetch_add(1);
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop.store(true);
    const double el = std::chrono::duration<double>(Clock::now() - t_begin).count();
    for (auto &x : th) x.join();
    const CpuStat cs1 = cpu_stat();
    uint64_t total = 0, eag = 0, wt = 0;
    std::vector<float> all;
    for (int t = 0; t < R; t++) { total += done[t]; eag += again[t]; wt += waits[t]; all.insert(all.end(), lat[t].begin(), lat[t].end()); }
    std::sort(all.begin(), all.end());
    auto q = [&](double f) { return all.empty() ? 0.f : all[std::min(all.size() - 1, (size_t)(f * all.size()))]; };
    uint64_t c1 = 0, i1 = 0, l1 = 0;
    if (by_call) circl_hip_queue_stats(w.queue, &c1, &i1, &l1);
    else circl_hip_keytable_coalesce_stats(w.table, &c1, &i1, &l1);
    rusage ru1{};
    getrusage(RUSAGE_SELF, &ru1);
    const double pu = us_of(ru1.ru_utime, ru0.ru_utime), ps = us_of(ru1.ru_stime, ru0.ru_stime), n = (double)std::max<uint64_t>(total, 1);
    printf("async %-6s R=%d W=%-5zu %10.0f items/s  latency us p50 %7.1f p99 %7.1f max %8.1f | %.1f items per launch | host CPU %.2f us per item "
           "(reactors user %.2f sys %.2f, dispatcher + runtime threads user %.2f sys %.2f), %.1f CPUs busy | EAGAIN %llu waits %llu  mismatches %llu\n",
           op.c_str(), R, W, total / el, q(0.50), q(0.99), all.empty() ? 0.f : all.back(), l1 > l0 ? (double)(i1 - i0) / (l1 - l0) : 0.0,
           cs1.usage_us > cs0.usage_us ? (cs1.usage_us - cs0.usage_us) / n : (pu + ps) / n, thr_user_us.load() / n, thr_sys_us.load() / n,
           (pu - thr_user_us.load()) / n, (ps - thr_sys_us.load()) / n, cs1.usage_us > cs0.usage_us ? (cs1.usage_us - cs0.usage_us) / (el * 1e6) : (pu + ps) / (el * 1e6),
           (unsigned long long)eag, (unsigned long long)wt, (unsigned long long)mismatches.load());
    fflush(stdout);
    CHECK(mismatches.load() == 0);
    if (by_call) CHECK(circl_hip_queue_close(w.queue) == 0);
    else CHECK(circl_hip_keytable_close(w.table) == 0);
    return 0;
}

// ---- --one-call: the time stamps of one blocking caller ----
static int one_call_main(int argc, char **argv) {
    const int calls = argc > 2 ? atoi(argv[2]) : 3000;
    CHECK(circl_hip_init() > 0);
    const char *names[7] = {"reserve rows", "copy inputs in", "turn + close", "wait for copies", "enqueue (launch)", "device + completion", "copy results out"};
    for (const char *opn : {"encaps", "decaps"}) {
        const std::string op = opn;
        for (int mode = 0; mode <= 2; mode++) {
            char buf[8];
            snprintf(buf, sizeof buf, "%d", mode);
            setenv("CIRCL_HIP_COALESCE_DONE", buf, 1);
            Work w;
            w.make(op);
            CHECK(circl_hip_keytable_set_coalesce(w.table, 256, 0) == 0);
            circl_hip_profile_call_stamps(1, nullptr);
            std::vector<uint8_t> o_ct(w.CT), o_ss(32), o_st(1);
            std::vector<std::vector<double>> d(8);
            size_t at = 0, bad = 0;
            for (int i = 0; i < calls + 200; i++) {
                const auto a = Clock::now();
                if (op == "encaps") CHECK(circl_hip_mlkem_encaps_table(w.table, &w.kidx[at], &w.m[32 * at], o_ct.data(), o_ss.data(), o_st.data(), 1) == 0);
                else CHECK(circl_hip_mlkem_decaps_table(w.table, &w.kidx[at], &w.ct[w.CT * at], o_ss.data(), o_st.data(), 1) == 0);
                const double tot = std::chrono::duration<double, std::micro>(Clock::now() - a).count();
                bad += memcmp(o_ss.data(), &w.ss[32 * at], 32) != 0;
                uint64_t s[8];
                circl_hip_profile_call_stamps(-1, s);
                if (i >= 200) {
                    for (int k = 0; k < 7; k++) d[k].push_back(s[k + 1] > s[k] ? (s[k + 1] - s[k]) / 1e3 : 0.0);
                    d[7].push_back(tot);
                }
                at = (at + 131) % (Work::POOL - 1);
            }
            circl_hip_profile_call_stamps(0, nullptr);
            auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            printf("%s, one blocking caller, completion mode %d (%s): call %.1f us median =", opn, mode,
                   mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "flag via hipStreamWriteValue32, polled" : "flag raised by the batch's own launch (or its finish kernel), polled", med(d[7]));
            double sum = 0;
            for (int k = 0; k < 7; k++) { printf(" %s %.1f |", names[k], med(d[k])); sum += med(d[k]); }
            printf(" (stages sum %.1f)  mismatches %zu\n", sum, bad);
            fflush(stdout);
            CHECK(bad == 0);
            CHECK(circl_hip_keytable_close(w.table) == 0);
        }
    }
    unsetenv("CIRCL_HIP_COALESCE_DONE");
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "--async")) return async_main(argc, argv);
    if (argc >= 2 && !strcmp(argv[1], "--one-call")) return one_call_main(argc, argv);
    if (argc < 7) {
        fprintf(stderr, "usage: %s <encaps|decaps|verify|sign|encaps_item> <coalesce max_items> <max_wait_us> <items per call> <seconds> <T> [T ...]\n", argv[0]);
        return 2;
    }
    const std::string op = argv[1];
    const size_t co_items = (size_t)atol(argv[2]);
    const unsigned co_wait = (unsigned)atoi(argv[3]);
    const size_t per_call = (size_t)std::max(1L, atol(argv[4]));
    const double seconds = atof(argv[5]);
    std::vector<int> Ts;
    for (int a = 6; a < argc; a++) Ts.push_back(atoi(argv[a]));
    CHECK(circl_hip_init() > 0);

    // ---- the pool of work: POOL items with known answers, NK resident keys ----
    const size_t NK = 8, POOL = 4096;
    const int kem = 768, dsa = 65;
    const size_t EK = circl_hip_mlkem_ek_size(kem), DK = circl_hip_mlkem_dk_size(kem), CT = circl_hip_mlkem_ct_size(kem);
    const size_t PK = circl_hip_mldsa_pk_size(dsa), SK = circl_hip_mldsa_sk_size(dsa), SIG = circl_hip_mldsa_sig_size(dsa);
    std::vector<uint32_t> kidx(POOL);
    for (size_t i = 0; i < POOL; i++) kidx[i] = (uint32_t)((i * 5 + i / 7) % NK);
    circl_hip_keytable *table = nullptr;
    circl_hip_queue *queue = nullptr;   // ops "encaps_call" / "decaps_call": the key comes with every item (circl_hip_queue)
    std::vector<uint8_t> ekeys, dkeys;  // ... NK key rows each
    std::vector<uint8_t> m, ct, ss, st, sig, ok, mblob;
    std::vector<uint64_t> moff;
    const size_t MSG = 32;
    std::vector<uint8_t> ek_rows;  // encaps_item: every item's own key (the rows a caller without a resident table passes)
    if (op == "encaps_item") {
        // the TLS-server shape: every call encapsulates ONCE, to a key that comes with the call (POOL distinct keys)
        std::vector<uint8_t> seed = bytes(64 * POOL, 1), dk(DK * POOL);
        ek_rows.resize(EK * POOL);
        CHECK(circl_hip_mlkem_keygen(kem, seed.data(), ek_rows.data(), dk.data(), POOL, 0) == 0);
        m = bytes(32 * POOL, 2);
        ct.resize(CT * POOL); ss.resize(32 * POOL); st.resize(POOL);
        CHECK(circl_hip_mlkem_encaps(kem, ek_rows.data(), m.data(), ct.data(), ss.data(), st.data(), POOL, 0) == 0);  // the answers
        if (co_items) CHECK(circl_hip_set_coalesce(co_items, co_wait) == 0);
    } else if (op == "encaps" || op == "decaps") {
        std::vector<uint8_t> seed = bytes(64 * NK, 1), ek(EK * NK), dk(DK * NK);
        CHECK(circl_hip_mlkem_keygen(kem, seed.data(), ek.data(), dk.data(), NK, 0) == 0);
        m = bytes(32 * POOL, 2);
        ct.resize(CT * POOL); ss.resize(32 * POOL); st.resize(POOL);
        circl_hip_keytable *pub = nullptr;
        CHECK(circl_hip_mlkem_keytable_new(kem, 0, ek.data(), NK, 0, nullptr, &pub) == 0);
        CHECK(circl_hip_mlkem_encaps_table(pub, kidx.data(), m.data(), ct.data(), ss.data(), st.data(), POOL) == 0);  // the reference answers
        if (op == "encaps") table = pub;
        else {
            circl_hip_keytable_free(pub);
            CHECK(circl_hip_mlkem_keytable_new(kem, 1, dk.data(), NK, 0, nullptr, &table) == 0);
        }
    } else if (op == "verify") {
        std::vector<uint8_t> seed = bytes(32 * NK, 3), pk(PK * NK), sk(SK * NK);
        CHECK(circl_hip_mldsa_keygen(dsa, seed.data(), pk.data(), sk.data(), NK, 0) == 0);
        circl_hip_keytable *signer = nullptr;
        CHECK(circl_hip_mldsa_privkeys_new(dsa, sk.data(), NK, 0, &signer) == 0);
        mblob = bytes(MSG * POOL, 4);
        moff.resize(POOL + 1);
        for (size_t i = 0; i <= POOL; i++) moff[i] = MSG * i;
        sig.resize(SIG * POOL + 16);
        CHECK(circl_hip_mldsa_sign_table_keyed(signer, kidx.data(), mblob.data(), moff.data(), nullptr, nullptr, nullptr, sig.data(), POOL) == 0);
        circl_hip_keytable_free(signer);
        for (size_t i = 0; i < POOL; i += 5) sig[SIG * i + 40 + (i % 64)] ^= 1;  // a fifth of the signatures are bad: both verdicts occur
        ok.resize(POOL);
        CHECK(circl_hip_mldsa_keytable_new(dsa, pk.data(), NK, 0, &table) == 0);
        CHECK(circl_hip_mldsa_verify_table(table, kidx.data(), sig.data(), mblob.data(), moff.data(), nullptr, nullptr, ok.data(), POOL) == 0);
        size_t good = 0;
        for (size_t i = 0; i < POOL; i++) good += ok[i];
        CHECK(good == POOL - (POOL + 4) / 5);
    } else if (op == "sign") {
        std::vector<uint8_t> seed = bytes(32 * NK, 3), pk(PK * NK), sk(SK * NK);
        CHECK(circl_hip_mldsa_keygen(dsa, seed.data(), pk.data(), sk.data(), NK, 0) == 0);
        CHECK(circl_hip_mldsa_privkeys_new(dsa, sk.data(), NK, 0, &table) == 0);
        mblob = bytes(MSG * POOL, 4);
        moff.resize(POOL + 1);
        for (size_t i = 0; i <= POOL; i++) moff[i] = MSG * i;
        sig.resize(SIG * POOL + 16);
        CHECK(circl_hip_mldsa_sign_table_keyed(table, kidx.data(), mblob.data(), moff.data(), nullptr, nullptr, nullptr, sig.data(), POOL) == 0);  // the answers
    } else {
        fprintf(stderr, "unknown op %s\n", op.c_str());
        return 2;
    }
    if (co_items && table) CHECK(circl_hip_keytable_set_coalesce(table, co_items, co_wait) == 0);
    printf("# %s, %zu item(s) per call, %zu resident keys, coalesce max_items=%zu max_wait_us=%u, %.1f s per point\n", op.c_str(), per_call, NK, co_items,
           co_wait, seconds);

    for (int T : Ts) {
        std::atomic<int> started{0};
        std::atomic<bool> stop{false};
        std::vector<std::vector<float>> lat(T);
        std::vector<uint64_t> calls(T, 0);
        std::atomic<uint64_t> mismatches{0};
        std::atomic<uint64_t> thr_user_us{0}, thr_sys_us{0};  // CPU of the caller threads themselves (the rest of the process: the HIP runtime's threads)
        rusage ru0{};
        getrusage(RUSAGE_SELF, &ru0);
        uint64_t c0 = 0, i0 = 0, l0 = 0;
        if (table) circl_hip_keytable_coalesce_stats(table, &c0, &i0, &l0);
        std::vector<std::thread> th;
        Clock::time_point t_begin;
        for (int t = 0; t < T; t++) {
            th.emplace_back([&, t] {
                std::vector<uint8_t> o_ct(CT * per_call), o_ss(32 * per_call), o_st(per_call), o_ok(per_call), o_sig(SIG * per_call + 16);
                std::vector<uint64_t> off(per_call + 1);
                lat[t].reserve(1 << 16);
                size_t at = ((size_t)t * 997) % (POOL - per_call);
                started.fetch_add(1);
                while (started.load() < T + 1) std::this_thread::yield();
                while (!stop.load(std::memory_order_relaxed)) {
                    const auto a = Clock::now();
                    bool good = true;
                    if (op == "encaps_item") {
                        CHECK(circl_hip_mlkem_encaps(kem, &ek_rows[EK * at], &m[32 * at], o_ct.data(), o_ss.data(), o_st.data(), per_call, 0) == 0);
                        good = !memcmp(o_ct.data(), &ct[CT * at], CT * per_call) && !memcmp(o_ss.data(), &ss[32 * at], 32 * per_call);
                    } else if (op == "encaps") {
                        CHECK(circl_hip_mlkem_encaps_table(table, &kidx[at], &m[32 * at], o_ct.data(), o_ss.data(), o_st.data(), per_call) == 0);
                        good = !memcmp(o_ct.data(), &ct[CT * at], CT * per_call) && !memcmp(o_ss.data(), &ss[32 * at], 32 * per_call);
                    } else if (op == "decaps") {
                        CHECK(circl_hip_mlkem_decaps_table(table, &kidx[at], &ct[CT * at], o_ss.data(), o_st.data(), per_call) == 0);
                        good = !memcmp(o_ss.data(), &ss[32 * at], 32 * per_call);
                    } else if (op == "sign") {
                        CHECK(circl_hip_mldsa_sign_table_keyed(table, &kidx[at], mblob.data(), &moff[at], nullptr, nullptr, nullptr, o_sig.data(), per_call) == 0);
                        good = !memcmp(o_sig.data(), &sig[SIG * at], SIG * per_call);
                    } else {
                        CHECK(circl_hip_mldsa_verify_table(table, &kidx[at], &sig[SIG * at], mblob.data(), &moff[at], nullptr, nullptr, o_ok.data(), per_call) == 0);
                        good = !memcmp(o_ok.data(), &ok[at], per_call);
                    }
                    const auto b = Clock::now();
                    if (!good) mismatches.fetch_add(1);
                    if (lat[t].size() < lat[t].capacity() || (calls[t] & 15) == 0) lat[t].push_back(std::chrono::duration<float, std::micro>(b - a).count());
                    calls[t]++;
                    at = (at + per_call * 131 + 1) % (POOL - per_call);
                }
                rusage ru{};
                if (getrusage(RUSAGE_THREAD, &ru) == 0) {
                    thr_user_us.fetch_add((uint64_t)ru.ru_utime.tv_sec * 1000000 + ru.ru_utime.tv_usec);
                    thr_sys_us.fetch_add((uint64_t)ru.ru_stime.tv_sec * 1000000 + ru.ru_stime.tv_usec);
                }
            });
        }
        while (started.load() < T) std::this_thread::yield();
        const CpuStat cs0 = cpu_stat();
        t_begin = Clock::now();
        started.fetch_add(1);
        std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
        stop.store(true);
        for (auto &x : th) x.join();
        const double el = std::chrono::duration<double>(Clock::now() - t_begin).count();
        const CpuStat cs1 = cpu_stat();
        uint64_t total = 0;
        std::vector<float> all;
        for (int t = 0; t < T; t++) { total += calls[t]; all.insert(all.end(), lat[t].begin(), lat[t].end()); }
        std::sort(all.begin(), all.end());
        auto q = [&](double f) { return all.empty() ? 0.f : all[std::min(all.size() - 1, (size_t)(f * all.size()))]; };
        uint64_t c1 = 0, i1 = 0, l1 = 0;
        if (table) circl_hip_keytable_coalesce_stats(table, &c1, &i1, &l1);
        printf("T=%-4d %10.0f ops/s  (%8.0f calls/s)  latency us p50 %7.1f  p99 %7.1f  max %8.1f", T, total * per_call / el, total / el, q(0.50), q(0.99),
               all.empty() ? 0.f : all.back());
        if (l1 > l0) printf("  | %.1f calls, %.1f items per launch", (double)(c1 - c0) / (l1 - l0), (double)(i1 - i0) / (l1 - l0));
        if (cs1.usage_us > cs0.usage_us)
            printf("  | CPU %.1f us per call, %.1f CPUs busy, throttled %ld x %.0f ms", (cs1.usage_us - cs0.usage_us) / std::max<uint64_t>(total, 1),
                   (cs1.usage_us - cs0.usage_us) / (el * 1e6), cs1.nr_throttled - cs0.nr_throttled, (cs1.throttled_us - cs0.throttled_us) / 1e3);
        {
            rusage ru1{};
            getrusage(RUSAGE_SELF, &ru1);
            auto us = [](const timeval &a, const timeval &b) { return (double)(a.tv_sec - b.tv_sec) * 1e6 + (a.tv_usec - b.tv_usec); };
            const double pu = us(ru1.ru_utime, ru0.ru_utime), ps = us(ru1.ru_stime, ru0.ru_stime), n = (double)std::max<uint64_t>(total, 1);
            printf("  | callers user %.1f sys %.1f, other threads user %.1f sys %.1f us per call", thr_user_us.load() / n, thr_sys_us.load() / n,
                   (pu - thr_user_us.load()) / n, (ps - thr_sys_us.load()) / n);
        }
        printf("  mismatches %llu\n", (unsigned long long)mismatches.load());
        fflush(stdout);
        CHECK(mismatches.load() == 0);
    }
    if (table) circl_hip_keytable_free(table);
    return 0;
}
