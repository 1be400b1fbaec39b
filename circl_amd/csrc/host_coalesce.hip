// host_coalesce.hip -- cross-caller coalescing of small host-buffer calls through one resident key table (host_common.h).
//
// The shape it serves is the reference's own: kem.Scheme.Encapsulate / Decapsulate and sign.Scheme.Verify take ONE key and ONE
// item (kem/mlkem/mlkem768/kyber.go:347-386, sign/mldsa/mldsa65/dilithium.go:305), and every in-tree consumer calls them that way
// from whatever goroutine handles the connection (kem/hybrid/hybrid.go:95-99, kem/xwing/xwing.go:259,288, hpke/algs.go:283-285).
// One such call costs a launch and a wait whatever the kernel does (~30 us), so a GPU serves them only if the calls of MANY
// callers become one launch.  No compute happens here: callers copy bytes, one of them launches the table's ordinary batch kernels.
//
//   caller      reserve rows of the OPEN batch (one compare-and-swap on the batch's packed reservation word: no lock, no system call) ->
//               copy inputs into its page-locked staging -> the caller that opened the batch LEADS it, the others sleep on the batch's
//               gate -> copy own rows out -> the one that returns the batch's last rows recycles it
//   leader      waits until it is its turn and the device has room (at most kInflight batches run at a time): the batch thus
//               collects exactly the calls that arrive while its predecessors run -- no timer at low load, big batches at high load
//               (group commit).  max_wait_us > 0 additionally lingers that long for company.  Then: close, wait for the copies of
//               the batch's callers, launch (zero-copy below zero_copy_bytes(), else one H2D per array, kernels, one D2H per array),
//               drain the stream, open the gate.
//   gate        a futex word; the leader wakes every sleeper with ONE system call.  (Measured, profiles/r05_concurrent.txt: with the
//               reservation under a lock and woken callers waking two more each -- a tree -- the callers' SYSTEM time grew from 4 us per
//               call at 32 callers to 19 at 64 and 73 at 128: lock convoys and futex-bucket contention; that is what this form removes.)
//
// Secrets: the page-locked rows of secret inputs / outputs are zeroed by the last caller out, the device staging by the leader
// (same rule as run_pipeline).
#include "host_common.h"

#include <immintrin.h>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <chrono>
#include <climits>

namespace circl {
namespace host {

namespace {

long futex_op(std::atomic<uint32_t> *addr, int op, uint32_t val, const timespec *ts = nullptr) {
    return syscall(SYS_futex, reinterpret_cast<uint32_t *>(addr), op | FUTEX_PRIVATE_FLAG, val, ts, nullptr, 0);
}

// The lock of the slow paths (opening a batch, a leader taking its turn, recycling: a few times per BATCH, never per call -- a call joins
// with a compare-and-swap): spins a few dozen times, then SLEEPS on its word -- callers may outnumber the CPUs the process may use by far
// (a container's CPU quota: threads that spin or yield there burn the quota of the threads that hold the lock or run the launch;
// measured, profiles/r05_concurrent.txt).  0 free, 1 held, 2 held with sleepers (Drepper's mutex).
struct SpinLock {
    std::atomic<uint32_t> word{0};
    void lock() {
        for (int spins = 0; spins < 64; spins++) {
            uint32_t z = 0;
            if (word.load(std::memory_order_relaxed) == 0 && word.compare_exchange_weak(z, 1, std::memory_order_acquire)) return;
            _mm_pause();
        }
        while (word.exchange(2, std::memory_order_acquire) != 0) futex_op(&word, FUTEX_WAIT, 2);
    }
    void unlock() {
        if (word.exchange(0, std::memory_order_release) == 2) futex_op(&word, FUTEX_WAKE, 1);
    }
};

// One-shot gate: wait() returns once open() was called.
struct Gate {
    std::atomic<uint32_t> word{0};
    std::atomic<int> sleepers{0};
    void wait(int spin) {
        for (int i = 0; i < spin; i++) {
            if (word.load(std::memory_order_acquire)) return;
            _mm_pause();
        }
        if (word.load(std::memory_order_acquire)) return;
        sleepers.fetch_add(1);  // (seq_cst: ordered against open()'s store / load pair)
        while (!word.load()) futex_op(&word, FUTEX_WAIT, 0);
        sleepers.fetch_sub(1);
    }
    void open() {
        word.store(1);
        if (sleepers.load() > 0) futex_op(&word, FUTEX_WAKE, INT_MAX);
    }
    void reset() { word.store(0); }
};

// A batch's reservation word: items (16 bits), bytes of its two ragged arrays (24 + 23 bits), and the CLOSED bit.  Joining a batch is one
// compare-and-swap on it; closing it (the leader, once) is one fetch_or, which also tells the leader the final counts.
constexpr uint64_t kRsvClosed = 1ull << 63;
inline size_t rsv_items(uint64_t r) { return (size_t)(r & 0xffff); }
inline size_t rsv_blob0(uint64_t r) { return (size_t)((r >> 16) & 0xffffff); }
inline size_t rsv_blob1(uint64_t r) { return (size_t)((r >> 40) & 0x7fffff); }
inline uint64_t rsv_pack(size_t items, size_t b0, size_t b1) { return (uint64_t)items | ((uint64_t)b0 << 16) | ((uint64_t)b1 << 40); }

constexpr int kMaxBlobs = 2;
using Clock = std::chrono::steady_clock;

struct CoBatch {
    enum State { FREE, OPEN, CLOSED };
    State state = FREE;               // (under Coalescer::lock)
    uint64_t ticket = 0;              // flush order
    std::atomic<uint64_t> rsv{kRsvClosed};  // the reservation word (above); closed whenever the batch is not open
    size_t count = 0;                 // items / ragged bytes of the closed batch (written by the leader when it closes it)
    size_t blob_used[kMaxBlobs] = {0, 0};
    std::atomic<bool> full{false};    // flush without lingering
    std::atomic<uint64_t> copied{0};  // items whose inputs are in the staging
    std::atomic<uint32_t> wseq{0};    // the leader sleeps on it while copied != count
    std::atomic<bool> leader_waits{false};  // ... and says so: only then is a writer's wake a system call
    std::atomic<uint32_t> callers{0}; // calls that joined (statistics)
    std::atomic<uint64_t> returned{0};  // items whose results were taken: the caller that brings it to `count` recycles the batch
    Gate done;
    int rc = 0;
    std::string err;
    Clock::time_point opened;
    uint8_t *hin = nullptr, *hout = nullptr, *d = nullptr;
    uint8_t *hin_dev = nullptr, *hout_dev = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t ev = nullptr;  // blocking-sync event (the leader sleeps until the batch is done instead of polling the stream)
};

}  // namespace

struct Coalescer {
    int dev = 0;
    size_t max_items = 0, call_max = 0;
    unsigned max_wait_us = 0;
    int inflight_max = 2;
    int spin = 0;  // gate spins before sleeping (0 when callers may outnumber the CPUs)
    bool blocking = false;  // the leader sleeps on an interrupt-driven event instead of hipStreamSynchronize's polling

    SpinLock lock;
    std::atomic<uint32_t> seq{0};  // bumped whenever something a leader / a caller without a batch waits for has changed
    int inflight = 0;
    uint64_t next_ticket = 0, serving = 0;
    std::atomic<CoBatch *> open{nullptr};  // the batch new calls join (may be stale: the reservation word decides)
    std::vector<CoBatch *> batches;

    // layout, fixed by the first call (all calls through one table have the same arrays)
    std::mutex init_mu;
    std::atomic<int> ready{0};  // 0 not laid out, 1 ready, -1 failed
    int init_rc = 0;
    std::string init_err;
    std::vector<size_t> in_row, out_row, in_ofs, out_ofs;
    std::vector<char> in_secret, out_secret;
    size_t nblob = 0, blob_cap = 0, blob_ofs[kMaxBlobs] = {0, 0}, off_ofs[kMaxBlobs] = {0, 0};
    size_t hin_bytes = 0, hout_bytes = 0, d_out_base = 0, ws_ofs = 0, ws_cap = 0, d_bytes = 0;

    std::atomic<uint64_t> n_calls{0}, n_items{0}, n_launches{0};
};

namespace {

void bump(Coalescer *co) {
    co->seq.fetch_add(1);
    futex_op(&co->seq, FUTEX_WAKE, INT_MAX);  // few sleepers: leaders of unflushed batches, callers waiting for a free batch
}

int lay_out(Coalescer *co, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs,
            const std::function<size_t(size_t)> &ws_bytes) {
    if (blobs.size() > (size_t)kMaxBlobs) return CIRCL_HIP_EPARAM;
    HIP_TRY(hipSetDevice(physical_device(co->dev)));
    const size_t N = co->max_items;
    size_t o = 0;
    for (auto &in : ins) {
        co->in_row.push_back(in.row);
        co->in_secret.push_back(in.secret);
        co->in_ofs.push_back(o);
        o += up256(in.row * N + 16);
    }
    co->nblob = blobs.size();
    co->blob_cap = up256(N * 512 + (size_t(64) << 10));  // ragged rows (messages, contexts): 512 B per item on average + 64 KB
    for (size_t k = 0; k < co->nblob; k++) {
        co->blob_ofs[k] = o;
        o += co->blob_cap + 256;
        co->off_ofs[k] = o;
        o += up256((N + 1) * 8);
    }
    co->hin_bytes = std::max<size_t>(o, 256);
    o = 0;
    for (auto &out : outs) {
        co->out_row.push_back(out.row);
        co->out_secret.push_back(out.secret);
        co->out_ofs.push_back(o);
        o += up256(out.row * N + 16);
    }
    co->hout_bytes = std::max<size_t>(o, 256);
    co->d_out_base = co->hin_bytes;
    co->ws_ofs = co->hin_bytes + co->hout_bytes;
    co->ws_cap = up256(ws_bytes(N));
    co->d_bytes = co->ws_ofs + co->ws_cap;
    const int nb = co->inflight_max + 3;  // one open, inflight_max running, two being read out
    for (int i = 0; i < nb; i++) {
        CoBatch *b = new CoBatch;
        co->batches.push_back(b);
        HIP_TRY(pinned_alloc(reinterpret_cast<void **>(&b->hin), co->hin_bytes));
        HIP_TRY(pinned_alloc(reinterpret_cast<void **>(&b->hout), co->hout_bytes));
        memset(b->hin, 0, co->hin_bytes);
        memset(b->hout, 0, co->hout_bytes);
        b->hin_dev = pinned_device_ptr(b->hin);
        b->hout_dev = pinned_device_ptr(b->hout);
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&b->d), co->d_bytes));
        HIP_TRY(hipStreamCreateWithFlags(&b->st, hipStreamNonBlocking));
        if (co->blocking) HIP_TRY(hipEventCreateWithFlags(&b->ev, hipEventBlockingSync | hipEventDisableTiming));
    }
    return CIRCL_HIP_OK;
}

bool same_shape(const Coalescer *co, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs) {
    if (ins.size() != co->in_row.size() || outs.size() != co->out_row.size() || blobs.size() != co->nblob) return false;
    for (size_t k = 0; k < ins.size(); k++)
        if (ins[k].row != co->in_row[k] || ins[k].per_call) return false;
    for (size_t k = 0; k < outs.size(); k++)
        if (outs[k].row != co->out_row[k]) return false;
    return true;
}

// the leader's part: returns once the batch's results are in its page-locked output area (or b->rc says why not)
void flush(Coalescer *co, CoBatch *b, const std::function<size_t(size_t)> &ws_bytes, const PipeOpts &opts, const std::function<int(Chunk &)> &launch) {
    // ---- its turn, and room on the device ----
    const auto deadline = b->opened + std::chrono::microseconds(co->max_wait_us);
    co->lock.lock();
    for (;;) {
        const bool turn = co->serving == b->ticket && co->inflight < co->inflight_max;
        const bool ripe = b->full.load() || co->max_wait_us == 0 || Clock::now() >= deadline;
        if (turn && ripe) break;
        const uint32_t s = co->seq.load();
        co->lock.unlock();
        if (turn) {  // lingering for company: until the deadline, or until the batch fills up
            const auto left = std::chrono::duration_cast<std::chrono::nanoseconds>(deadline - Clock::now()).count();
            if (left > 0) {
                timespec ts{(time_t)(left / 1000000000), (long)(left % 1000000000)};
                futex_op(&co->seq, FUTEX_WAIT, s, &ts);
            }
        } else {
            futex_op(&co->seq, FUTEX_WAIT, s);
        }
        co->lock.lock();
    }
    co->inflight++;
    co->serving++;
    b->state = CoBatch::CLOSED;
    CoBatch *expect = b;
    co->open.compare_exchange_strong(expect, nullptr);
    const uint64_t r = b->rsv.fetch_or(kRsvClosed);  // no call joins from here on; what was reserved so far is the batch
    const size_t cnt = rsv_items(r);
    b->count = cnt;
    b->blob_used[0] = rsv_blob0(r);
    b->blob_used[1] = rsv_blob1(r);
    co->n_items.fetch_add(cnt, std::memory_order_relaxed);
    co->lock.unlock();
    bump(co);  // (the next batch's leader may now be first in line)
    // ---- the batch's callers have copied their rows in ----
    if (b->copied.load() != cnt) {
        b->leader_waits.store(true);
        for (;;) {
            const uint32_t s = b->wseq.load();
            if (b->copied.load() == cnt) break;
            futex_op(&b->wseq, FUTEX_WAIT, s);
        }
        b->leader_waits.store(false);
    }
    for (size_t k = 0; k < co->nblob; k++) reinterpret_cast<uint64_t *>(b->hin + co->off_ofs[k])[cnt] = b->blob_used[k];

    size_t moved = 0;
    for (size_t k = 0; k < co->in_row.size(); k++) moved += co->in_row[k] * cnt;
    for (size_t k = 0; k < co->out_row.size(); k++) moved += co->out_row[k] * cnt;
    for (size_t k = 0; k < co->nblob; k++) moved += b->blob_used[k] + 8 * (cnt + 1);
    const bool zc = moved <= zero_copy_bytes() && b->hin_dev && b->hout_dev;
    const size_t wsb = std::min(co->ws_cap, up256(ws_bytes(cnt)));
    auto run = [&]() -> int {
        HIP_TRY(hipSetDevice(physical_device(co->dev)));
        uint8_t *in_base = zc ? b->hin_dev : b->d, *out_base = zc ? b->hout_dev : b->d + co->d_out_base;
        Chunk c;
        c.cnt = cnt; c.st = b->st;
        c.ws = b->d + co->ws_ofs; c.ws_bytes = wsb;
        for (size_t k = 0; k < co->in_row.size(); k++) {
            c.in.push_back(in_base + co->in_ofs[k]);
            if (!zc && co->in_row[k]) HIP_TRY(hipMemcpyAsync(b->d + co->in_ofs[k], b->hin + co->in_ofs[k], co->in_row[k] * cnt, hipMemcpyHostToDevice, b->st));
        }
        for (size_t k = 0; k < co->nblob; k++) {
            c.blob.push_back(in_base + co->blob_ofs[k]);  // the batch's offsets count from the start of its own blob area
            c.off.push_back(reinterpret_cast<const uint64_t *>(in_base + co->off_ofs[k]));
            if (!zc) {
                if (b->blob_used[k]) HIP_TRY(hipMemcpyAsync(b->d + co->blob_ofs[k], b->hin + co->blob_ofs[k], b->blob_used[k], hipMemcpyHostToDevice, b->st));
                HIP_TRY(hipMemcpyAsync(b->d + co->off_ofs[k], b->hin + co->off_ofs[k], (cnt + 1) * 8, hipMemcpyHostToDevice, b->st));
            }
        }
        for (size_t k = 0; k < co->out_row.size(); k++) c.out.push_back(out_base + co->out_ofs[k]);
        if (int rc = launch(c)) return rc;
        for (size_t k = 0; k < co->out_row.size() && !zc; k++)
            if (co->out_row[k]) HIP_TRY(hipMemcpyAsync(b->hout + co->out_ofs[k], b->d + co->d_out_base + co->out_ofs[k], co->out_row[k] * cnt, hipMemcpyDeviceToHost, b->st));
        return CIRCL_HIP_OK;
    };
    g_err.clear();
    int rc = run();
    hipError_t se = hipSuccess;
    if (b->ev) {
        se = hipEventRecord(b->ev, b->st);
        if (se == hipSuccess) se = hipEventSynchronize(b->ev);
    } else {
        se = hipStreamSynchronize(b->st);
    }
    if (se != hipSuccess && rc == CIRCL_HIP_OK) {
        rc = CIRCL_HIP_EHIP;
        g_err = "coalesced batch: hipStreamSynchronize failed";
        (void)hipGetLastError();
    }
    b->rc = rc;
    b->err = g_err;
    co->n_launches.fetch_add(1, std::memory_order_relaxed);
    co->lock.lock();
    co->inflight--;
    co->lock.unlock();
    bump(co);         // the next leader goes first: its launch overlaps the wake-ups and copies of this batch's callers
    b->done.open();
    // ---- nothing secret stays in the device staging (enqueued behind the results; the batch's next use is on the same stream) ----
    if (opts.wipe_device) {
        if (!zc) {
            for (size_t k = 0; k < co->in_row.size(); k++)
                if (co->in_secret[k] && co->in_row[k]) (void)hipMemsetAsync(b->d + co->in_ofs[k], 0, co->in_row[k] * cnt, b->st);
            for (size_t k = 0; k < co->out_row.size(); k++)
                if (co->out_secret[k] && co->out_row[k]) (void)hipMemsetAsync(b->d + co->d_out_base + co->out_ofs[k], 0, co->out_row[k] * cnt, b->st);
        }
        const size_t sec = opts.ws_secret_bytes ? std::min(wsb, opts.ws_secret_bytes(cnt)) : wsb;
        if (sec) (void)hipMemsetAsync(b->d + co->ws_ofs, 0, sec, b->st);
        (void)hipGetLastError();
    }
}

// the last caller out: wipe the page-locked rows that held secrets, hand the batch back
void recycle(Coalescer *co, CoBatch *b) {
    for (size_t k = 0; k < co->in_row.size(); k++)
        if (co->in_secret[k]) memset(b->hin + co->in_ofs[k], 0, co->in_row[k] * b->count);
    for (size_t k = 0; k < co->out_row.size(); k++)
        if (co->out_secret[k]) memset(b->hout + co->out_ofs[k], 0, co->out_row[k] * b->count);
    co->n_calls.fetch_add(b->callers.load(), std::memory_order_relaxed);  // (the statistics: once per batch)
    co->lock.lock();
    b->state = CoBatch::FREE;
    co->lock.unlock();
    bump(co);
}

}  // namespace

Coalescer *coalescer_new(int dev, size_t max_items, unsigned max_wait_us) {
    Coalescer *co = new (std::nothrow) Coalescer;
    if (!co) return nullptr;
    co->dev = dev;
    co->max_items = std::min<size_t>(std::max<size_t>(max_items, 2), size_t(1) << 13);
    co->call_max = std::max<size_t>(1, co->max_items / 4);
    co->max_wait_us = std::min(max_wait_us, 100000u);
    co->inflight_max = env_int("CIRCL_HIP_COALESCE_INFLIGHT", 2, 1, 8);
    co->spin = env_int("CIRCL_HIP_COALESCE_SPIN", 0, 0, 1 << 20);
    co->blocking = env_int("CIRCL_HIP_COALESCE_BLOCKING", 0, 0, 1) != 0;
    return co;
}
void coalescer_free(Coalescer *co) {
    if (!co) return;
    // (the table's contract: no call is in flight when it is freed)
    if (!co->batches.empty() && hipSetDevice(physical_device(co->dev)) == hipSuccess) {
        for (CoBatch *b : co->batches) {
            if (b->st) { (void)hipStreamSynchronize(b->st); (void)hipStreamDestroy(b->st); }
            if (b->ev) (void)hipEventDestroy(b->ev);
            if (b->hin) { memset(b->hin, 0, co->hin_bytes); (void)pinned_free(b->hin); }
            if (b->hout) { memset(b->hout, 0, co->hout_bytes); (void)pinned_free(b->hout); }
            if (b->d) (void)hipFree(b->d);
        }
        (void)hipGetLastError();
    }
    for (CoBatch *b : co->batches) delete b;
    delete co;
}
size_t coalescer_call_max(const Coalescer *co) { return co ? co->call_max : 0; }
void coalescer_stats(const Coalescer *co, uint64_t *calls, uint64_t *items, uint64_t *launches) {
    if (calls) *calls = co ? co->n_calls.load() : 0;
    if (items) *items = co ? co->n_items.load() : 0;
    if (launches) *launches = co ? co->n_launches.load() : 0;
}

int coalesce_run(Coalescer *co, size_t n, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs,
                 const std::function<size_t(size_t)> &ws_bytes, const PipeOpts &opts, const std::function<int(Chunk &)> &launch) {
    if (n == 0) return CIRCL_HIP_OK;
    if (!co || n > co->call_max || blobs.size() > (size_t)kMaxBlobs) return kNotCoalesced;
    if (co->ready.load(std::memory_order_acquire) == 0) {
        std::lock_guard<std::mutex> lk(co->init_mu);
        if (co->ready.load() == 0) {
            g_err.clear();
            co->init_rc = lay_out(co, ins, blobs, outs, ws_bytes);
            co->init_err = g_err;
            (void)hipGetLastError();
            co->ready.store(co->init_rc == CIRCL_HIP_OK ? 1 : -1, std::memory_order_release);
        }
    }
    if (co->ready.load(std::memory_order_acquire) < 0) { g_err = co->init_err; return co->init_rc; }
    if (!same_shape(co, ins, blobs, outs)) return kNotCoalesced;
    size_t bb[kMaxBlobs] = {0, 0};
    for (size_t k = 0; k < blobs.size(); k++) {
        if (blobs[k].blob && !blobs[k].off) { g_err = "a blob without offsets"; return CIRCL_HIP_EPARAM; }
        bb[k] = blobs[k].blob ? (size_t)(blobs[k].off[n] - blobs[k].off[0]) : 0;
        if (bb[k] > co->blob_cap / 4) return kNotCoalesced;  // a long message: its own call
    }

    // ---- reserve rows: one compare-and-swap on the open batch's reservation word; the lock only to open a new batch ----
    CoBatch *b = nullptr;
    bool leader = false;
    size_t pos = 0, bpos[kMaxBlobs] = {0, 0};
    for (;;) {
        b = co->open.load(std::memory_order_acquire);
        bool joined = false, crowded = false;
        if (b) {
            uint64_t r = b->rsv.load(std::memory_order_relaxed);
            while (!(r & kRsvClosed)) {
                if (rsv_items(r) + n > co->max_items || rsv_blob0(r) + bb[0] > co->blob_cap || rsv_blob1(r) + bb[1] > co->blob_cap) { crowded = true; break; }
                if (b->rsv.compare_exchange_weak(r, r + rsv_pack(n, bb[0], bb[1]), std::memory_order_acq_rel, std::memory_order_relaxed)) {
                    pos = rsv_items(r);
                    bpos[0] = rsv_blob0(r);
                    bpos[1] = rsv_blob1(r);
                    joined = true;
                    if (pos + n >= co->max_items && !b->full.exchange(true)) bump(co);  // filled up: a lingering leader need not wait any longer
                    break;
                }
            }
        }
        if (joined) break;
        // no open batch, or this call does not fit into it: open the next one (under the lock: once per batch, not per call)
        co->lock.lock();
        CoBatch *cur = co->open.load();
        if (cur != b) { co->lock.unlock(); continue; }  // somebody else already did
        if (b && crowded && !b->full.exchange(true)) {  // its leader flushes it as soon as it may
            co->lock.unlock();
            bump(co);
            co->lock.lock();
            if (co->open.load() != b) { co->lock.unlock(); continue; }
        }
        CoBatch *fresh = nullptr;
        for (CoBatch *c : co->batches)
            if (c->state == CoBatch::FREE) { fresh = c; break; }
        if (!fresh) {  // every batch is busy: wait for one to come back
            const uint32_t sq = co->seq.load();
            co->lock.unlock();
            futex_op(&co->seq, FUTEX_WAIT, sq);
            continue;
        }
        fresh->state = CoBatch::OPEN;
        fresh->ticket = co->next_ticket++;
        fresh->full.store(false);
        fresh->copied.store(0);
        fresh->returned.store(0);
        fresh->callers.store(0);
        fresh->rc = 0;
        fresh->done.reset();
        fresh->opened = Clock::now();
        fresh->rsv.store(rsv_pack(n, bb[0], bb[1]), std::memory_order_release);  // open, with this call's rows at its head
        co->open.store(fresh, std::memory_order_release);
        co->lock.unlock();
        b = fresh;
        leader = true;
        if (n >= co->max_items) b->full.store(true);
        break;
    }
    b->callers.fetch_add(1, std::memory_order_relaxed);

    // ---- copy this call's rows in ----
    for (size_t k = 0; k < ins.size(); k++) {
        if (!ins[k].row) continue;
        uint8_t *dst = b->hin + co->in_ofs[k] + pos * ins[k].row;
        if (ins[k].p) memcpy(dst, ins[k].p, ins[k].row * n);
        else memset(dst, 0, ins[k].row * n);
    }
    for (size_t k = 0; k < blobs.size(); k++) {
        uint64_t *off = reinterpret_cast<uint64_t *>(b->hin + co->off_ofs[k]) + pos;
        if (blobs[k].blob) {
            if (bb[k]) memcpy(b->hin + co->blob_ofs[k] + bpos[k], blobs[k].blob + blobs[k].off[0], bb[k]);
            for (size_t i = 0; i < n; i++) off[i] = bpos[k] + (blobs[k].off[i] - blobs[k].off[0]);
        } else {
            for (size_t i = 0; i < n; i++) off[i] = bpos[k];  // absent: empty rows
        }
    }
    b->copied.fetch_add(n);
    b->wseq.fetch_add(1);
    if (b->leader_waits.load()) futex_op(&b->wseq, FUTEX_WAKE, 1);  // (seq_cst on both sides: a leader that missed the count sees the new wseq)

    if (leader) flush(co, b, ws_bytes, opts, launch);
    else b->done.wait(co->spin);

    // ---- results ----
    const int rc = b->rc;
    if (rc) g_err = b->err;
    else
        for (size_t k = 0; k < outs.size(); k++)
            if (outs[k].p && outs[k].row) memcpy(outs[k].p, b->hout + co->out_ofs[k] + pos * outs[k].row, outs[k].row * n);
    // (count: written by the leader before the gate opened -- and read BEFORE this call's rows are handed back: once they are, the last
    // caller may recycle the batch and the next generation's leader overwrite it)
    const size_t total = b->count;
    if (b->returned.fetch_add(n) + n == total) recycle(co, b);
    return rc;
}

// ---- process-wide coalescers of the non-table entry points ----
namespace {
constexpr int kCallParams = 4, kCallDevs = 64;
std::atomic<size_t> g_call_items{0};
std::atomic<uint32_t> g_call_wait{0};
std::atomic<Coalescer *> g_call[kCoOps][kCallParams][kCallDevs];
std::mutex g_call_mu;
}  // namespace
Coalescer *call_coalescer(int op, int param_slot, int dev) {
    const size_t items = g_call_items.load(std::memory_order_acquire);
    if (!items || op < 0 || op >= kCoOps || param_slot < 0 || param_slot >= kCallParams || dev < 0 || dev >= kCallDevs) return nullptr;
    Coalescer *co = g_call[op][param_slot][dev].load(std::memory_order_acquire);
    if (co) return co;
    std::lock_guard<std::mutex> lk(g_call_mu);
    co = g_call[op][param_slot][dev].load();
    if (!co) {
        co = coalescer_new(dev, items, g_call_wait.load());  // (never freed: calls may be in flight at any time)
        g_call[op][param_slot][dev].store(co, std::memory_order_release);
    }
    return co;
}
void call_coalescing_set(size_t max_items, uint32_t max_wait_us) {
    g_call_wait.store(max_wait_us);
    g_call_items.store(max_items, std::memory_order_release);
}

}  // namespace host
}  // namespace circl

// ---- C ABI -----------------------------------------------------------------------------------------------------------------
#include "keytable.h"
using namespace circl::host;

extern "C" {

int circl_hip_keytable_set_coalesce(circl_hip_keytable *t, size_t max_items, uint32_t max_wait_us) {
    if (!t || t->magic != kKeytableMagic) return CIRCL_HIP_EPARAM;
    if (t->device < 0) {  // a replicated table: every replica batches the small calls routed to it
        for (int d = 0; d < t->nreplica; d++)
            if (int rc = circl_hip_keytable_set_coalesce(t->replica[d], max_items, max_wait_us)) return rc;
        return CIRCL_HIP_OK;
    }
    if (t->coalescer) { coalescer_free(t->coalescer); t->coalescer = nullptr; }
    if (max_items == 0) return CIRCL_HIP_OK;
    t->coalescer = coalescer_new(t->device, max_items, max_wait_us);
    return t->coalescer ? CIRCL_HIP_OK : CIRCL_HIP_ENOMEM;
}
int circl_hip_set_coalesce(size_t max_items, uint32_t max_wait_us) {
    circl::host::call_coalescing_set(max_items, max_wait_us);
    return CIRCL_HIP_OK;
}
int circl_hip_keytable_coalesce_stats(const circl_hip_keytable *t, uint64_t *calls, uint64_t *items, uint64_t *launches) {
    if (calls) *calls = 0;
    if (items) *items = 0;
    if (launches) *launches = 0;
    if (!t || t->magic != kKeytableMagic) return CIRCL_HIP_EPARAM;
    const int nr = t->device < 0 ? t->nreplica : 1;
    for (int d = 0; d < nr; d++) {
        const circl_hip_keytable *r = t->device < 0 ? t->replica[d] : t;
        uint64_t c = 0, i = 0, l = 0;
        coalescer_stats(r->coalescer, &c, &i, &l);
        if (calls) *calls += c;
        if (items) *items += i;
        if (launches) *launches += l;
    }
    return CIRCL_HIP_OK;
}

}  // extern "C"
