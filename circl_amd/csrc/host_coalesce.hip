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

//
// ASYNCHRONOUS form (round 6; circl_hip_keytable_async_start, *_table_submit, circl_hip_poll / _wait / _keytable_eventfd): the blocking
// form above parks one OS thread per outstanding call (a goroutine in a blocking cgo call holds an M), and the futex sleep + wake of
// every caller was 7.6 of the 13 us of host CPU a coalesced call cost (profiles/r05_concurrent_final.txt).  Here NOBODY sleeps per call:
//   submitter   reserves rows exactly as above, copies its inputs in, leaves a record of where its results go, gets a TICKET, returns
//   dispatcher  ONE library thread per queue: closes the oldest open batch whenever the device has room, launches it, and -- while the
//               next batches run -- copies each finished batch's rows straight into the callers' output buffers, then publishes
//               `completed` (tickets are batch numbers: a ticket is done once completed >= ticket), wakes circl_hip_wait sleepers
//               and signals the queue's eventfd if it has one.  It sleeps (futex) only when the queue is empty.
//   host        polls tickets (one atomic load each), or blocks ONE thread in circl_hip_wait, or puts the eventfd in its epoll set.
#include <immintrin.h>
#include <linux/futex.h>
#include <sched.h>
#include <sys/eventfd.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <chrono>
#include <climits>
#include <deque>

namespace circl {
namespace host {

std::atomic<bool> g_stamps_on{false};
thread_local CallStamps g_stamps;
thread_local TailOffer g_tail_offer;

namespace {

inline uint64_t now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
#define STAMP(field)                                   \
    do {                                               \
        if (stamps) g_stamps.field = now_ns();         \
    } while (0)

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
// compare-and-swap on it; closing it (the leader / the dispatcher, once) is one fetch_or, which also tells the final counts.
constexpr uint64_t kRsvClosed = 1ull << 63;
inline size_t rsv_items(uint64_t r) { return (size_t)(r & 0xffff); }
inline size_t rsv_blob0(uint64_t r) { return (size_t)((r >> 16) & 0xffffff); }
inline size_t rsv_blob1(uint64_t r) { return (size_t)((r >> 40) & 0x7fffff); }
inline uint64_t rsv_pack(size_t items, size_t b0, size_t b1) { return (uint64_t)items | ((uint64_t)b0 << 16) | ((uint64_t)b1 << 40); }

constexpr int kMaxBlobs = 2, kMaxOuts = 4;
constexpr int kRcRing = 1024;  // batch return codes kept for circl_hip_poll (a ticket older than this many batches reads as done: see coalescer_state)
using Clock = std::chrono::steady_clock;

// asynchronous form: where the results of one submitted call go.  The record of a call is recs[its first row]: the submitter fills it,
// copies its rows in and then stamps `ready` with the batch's generation (release); the dispatcher walks the closed batch's records
// from row 0 (the next one starts n rows further) and waits for each stamp.  A submit therefore touches ONE shared word (the
// reservation) -- a per-batch counter of calls and one of copied rows cost four reactors more than the work itself
// (profiles/r06_async.txt).  A cache line each: neighbours belong to other threads.
struct alignas(64) CallRec {
    std::atomic<uint64_t> ready{0};
    uint32_t n = 0;
    uint8_t *out[kMaxOuts] = {nullptr, nullptr, nullptr, nullptr};
};

struct CoBatch {
    enum State { FREE, OPEN, CLOSED };
    State state = FREE;               // (under Coalescer::lock)
    uint64_t ticket = 0;              // flush order
    alignas(64) std::atomic<uint64_t> rsv{kRsvClosed};  // the reservation word (above); closed whenever the batch is not open.  A line of its own:
                                                         // every joining call writes it, the fields around it are read by others
    alignas(64) uint64_t agen = 0;    // asynchronous form: the generation the records of this use are stamped with (set when the batch opens)
    size_t count = 0;                 // items / ragged bytes of the closed batch (written when it is closed)
    size_t blob_used[kMaxBlobs] = {0, 0};
    std::atomic<bool> full{false};    // flush without lingering
    std::atomic<uint64_t> copied{0};  // items whose inputs are in the staging
    std::atomic<uint32_t> wseq{0};    // the leader sleeps on it while copied != count
    std::atomic<bool> leader_waits{false};  // ... and says so: only then is a writer's wake a system call
    std::atomic<uint32_t> callers{0}; // calls that joined (statistics; asynchronous form: index into recs)
    std::atomic<uint64_t> returned{0};  // items whose results were taken: the caller that brings it to `count` recycles the batch
    Gate done;
    int rc = 0;
    std::string err;
    Clock::time_point opened;
    uint8_t *hin = nullptr, *hout = nullptr, *d = nullptr;
    uint8_t *hin_dev = nullptr, *hout_dev = nullptr;
    hipStream_t st = nullptr;
    // completion flag: a dword at the end of the page-locked output area that the stream writes AFTER the batch's last operation
    // (CIRCL_HIP_COALESCE_DONE): whoever waits for the batch polls host memory instead of calling into the runtime
    std::atomic<uint32_t> *flag = nullptr;
    uint32_t *flag_dev = nullptr;
    unsigned *count_dev = nullptr;  // the workgroup counter of a launch that raises the flag itself (TailFlag): 256 bytes behind the workspace
    uint32_t gen = 0, uses = 0;
    bool wiped = false;       // the stream's finish kernel zeroes the device staging (completion mode 2): wipe_device has nothing to add
    bool zc = false;          // the launched batch ran zero-copy
    size_t wsb = 0;           // ... with this much workspace
    CallRec *recs = nullptr;  // [max_items], asynchronous form
    uint32_t ncalls = 0;      // ... counted by the dispatcher's walk
};

// The last thing on a batch's stream in completion mode 2: ONE workgroup zeroes what was secret in the device staging (the secret
// input and output rows -- only when the batch was copied rather than run zero-copy -- and the secret head of the workspace), then raises the completion flag in the page-locked output area.  One launch instead of a flag kernel and
// up to three hipMemsetAsync calls of 3-5 us of host time each (the dispatcher's loop is the serving path's bottleneck:
// profiles/r06_async.txt).  Ranges are 16-byte aligned (the staging offsets are multiples of 256); lengths are rounded up inside
// their areas by the caller.
struct WipeRange {
    uint8_t *p;
    size_t bytes;
};
constexpr int kWipeRanges = 6;
struct WipeRanges {
    WipeRange r[kWipeRanges];
};
__global__ void __launch_bounds__(256) coalesce_finish_kernel(WipeRanges w, uint32_t *flag, uint32_t value) {
#pragma unroll
    for (int k = 0; k < kWipeRanges; k++) {
        uint4 *q = reinterpret_cast<uint4 *>(w.r[k].p);
        const size_t n16 = w.r[k].bytes / 16;
        for (size_t i = threadIdx.x; i < n16; i += 256) q[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace

struct Coalescer {
    int dev = 0;
    size_t max_items = 0, call_max = 0;
    unsigned max_wait_us = 0;
    int inflight_max = 2;
    int done_mode = 2;      // how a batch's end is noticed -- 0: hipStreamSynchronize / hipStreamQuery; 1: a flag in page-locked memory written by
                            // hipStreamWriteValue32, polled; 2: the flag written by the stream's finish kernel (which also wipes), polled

    bool owner_counts = false;  // the owner (a key table: TableUse) already counts the calls inside: `active` stays untouched

    alignas(64) SpinLock lock;
    std::atomic<uint32_t> seq{0};  // bumped whenever something a leader / a caller without a batch waits for has changed
    int inflight = 0;
    uint64_t next_ticket = 0, serving = 0, next_agen = 1;
    alignas(64) std::atomic<CoBatch *> open{nullptr};  // the batch new calls join (may be stale: the reservation word decides); read by every call
    std::vector<CoBatch *> batches;
    std::atomic<int> seq_waiters{0};  // submitters asleep on `seq` for a free batch (the dispatcher's wake is a system call only then)
    alignas(64) std::atomic<uint32_t> active{0};  // calls inside coalesce_run / coalesce_submit (coalescer_idle)

    // layout, fixed by the first call (all calls through one table have the same arrays)
    std::mutex init_mu;
    std::atomic<int> ready{0};  // 0 not laid out, 1 ready
    Clock::time_point init_failed_at{};  // a failed layout is retried after a back-off, not remembered for ever
    int init_rc = 0;
    std::string init_err;
    std::vector<size_t> in_row, out_row, in_ofs, out_ofs;
    std::vector<char> in_secret, out_secret;
    size_t nblob = 0, blob_cap = 0, blob_ofs[kMaxBlobs] = {0, 0}, off_ofs[kMaxBlobs] = {0, 0};
    size_t hin_bytes = 0, hout_bytes = 0, flag_ofs = 0, d_out_base = 0, ws_ofs = 0, ws_cap = 0, d_bytes = 0;

    alignas(64) std::atomic<uint64_t> n_calls{0}, n_items{0}, n_launches{0};

    // ---- asynchronous form ----
    bool async = false;
    std::thread disp;
    std::atomic<bool> stop{false};
    std::function<int(Chunk &)> a_launch;
    std::function<size_t(size_t)> a_ws;
    PipeOpts a_opts;
    int efd = -1;
    unsigned spin_us = 20;                  // the dispatcher polls this long for new work before it sleeps
    std::deque<CoBatch *> unclosed;         // opened, not yet closed, oldest first (under lock)
    alignas(64) std::atomic<int> n_unclosed{0};  // written once per batch by a submitter, read by the dispatcher's loop
    std::atomic<uint32_t> disp_word{0};     // the dispatcher sleeps on it when the queue is empty
    std::atomic<int> disp_sleeping{0};
    // what pollers read (written once per finished batch, by the dispatcher): a line of their own
    alignas(64) std::atomic<uint64_t> completed{0};  // batches finished, in ticket order: ticket t is done once completed >= t
    std::atomic<uint64_t> n_failed{0};
    std::atomic<uint32_t> done_word{0};     // circl_hip_wait sleeps on it
    std::atomic<int> done_waiters{0};
    alignas(64) std::atomic<int> rc_ring[kRcRing];
};

namespace {

void bump(Coalescer *co) {
    co->seq.fetch_add(1);
    futex_op(&co->seq, FUTEX_WAKE, INT_MAX);  // few sleepers: leaders of unflushed batches, callers waiting for a free batch
}

void free_batches(Coalescer *co) {
    if (!co->batches.empty() && hipSetDevice(physical_device(co->dev)) == hipSuccess) {
        for (CoBatch *b : co->batches) {
            if (b->st) { (void)hipStreamSynchronize(b->st); (void)hipStreamDestroy(b->st); }
            if (b->hin) { memset(b->hin, 0, co->hin_bytes); (void)pinned_free(b->hin); }
            if (b->hout) { memset(b->hout, 0, co->hout_bytes); (void)pinned_free(b->hout); }
            if (b->d) (void)hipFree(b->d);
        }
        (void)hipGetLastError();
    }
    for (CoBatch *b : co->batches) {
        delete[] b->recs;
        delete b;
    }
    co->batches.clear();
    co->in_row.clear(); co->out_row.clear(); co->in_ofs.clear(); co->out_ofs.clear(); co->in_secret.clear(); co->out_secret.clear();
}

int lay_out(Coalescer *co, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs,
            const std::function<size_t(size_t)> &ws_bytes) {
    if (blobs.size() > (size_t)kMaxBlobs || outs.size() > (size_t)kMaxOuts) return CIRCL_HIP_EPARAM;
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
    co->flag_ofs = std::max<size_t>(o, 256);
    co->hout_bytes = co->flag_ofs + 256;  // (the completion flag has the last 256 bytes to itself)
    co->d_out_base = co->hin_bytes;
    co->ws_ofs = co->hin_bytes + co->hout_bytes;
    co->ws_cap = up256(ws_bytes(N));
    co->d_bytes = co->ws_ofs + co->ws_cap + 256;
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
        b->flag = reinterpret_cast<std::atomic<uint32_t> *>(b->hout + co->flag_ofs);
        b->flag_dev = b->hout_dev ? reinterpret_cast<uint32_t *>(b->hout_dev + co->flag_ofs) : nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&b->d), co->d_bytes));
        b->count_dev = reinterpret_cast<unsigned *>(b->d + co->d_bytes - 256);
        HIP_TRY(hipMemset(b->count_dev, 0, 256));
        HIP_TRY(hipStreamCreateWithFlags(&b->st, hipStreamNonBlocking));
        if (co->async) {
            b->recs = new (std::nothrow) CallRec[N];
            if (!b->recs) return CIRCL_HIP_ENOMEM;
        }
    }
    return CIRCL_HIP_OK;
}

// first use: lay the batches out.  A failure (a transient out-of-memory, say) is NOT sticky: what was allocated goes back, the caller
// takes the un-coalesced path (kNotCoalesced), and the next call after a one-second back-off tries again.
int ensure_layout(Coalescer *co, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs,
                  const std::function<size_t(size_t)> &ws_bytes) {
    if (co->ready.load(std::memory_order_acquire) == 1) return CIRCL_HIP_OK;
    std::lock_guard<std::mutex> lk(co->init_mu);
    if (co->ready.load() == 1) return CIRCL_HIP_OK;
    if (co->init_rc != CIRCL_HIP_OK && Clock::now() - co->init_failed_at < std::chrono::seconds(1)) return kNotCoalesced;
    g_err.clear();
    co->init_rc = lay_out(co, ins, blobs, outs, ws_bytes);
    (void)hipGetLastError();
    if (co->init_rc != CIRCL_HIP_OK) {
        co->init_err = g_err;
        co->init_failed_at = Clock::now();
        free_batches(co);
        return co->async ? co->init_rc : kNotCoalesced;
    }
    co->ready.store(1, std::memory_order_release);
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

// ---- the parts of a flush (shared by the blocking leader and the asynchronous dispatcher) ----

// under co->lock: no call joins from here on; what was reserved so far is the batch
void close_batch(Coalescer *co, CoBatch *b) {
    co->inflight++;
    co->serving++;
    b->state = CoBatch::CLOSED;
    CoBatch *expect = b;
    co->open.compare_exchange_strong(expect, nullptr);
    const uint64_t r = b->rsv.fetch_or(kRsvClosed);
    const size_t cnt = rsv_items(r);
    b->count = cnt;
    b->blob_used[0] = rsv_blob0(r);
    b->blob_used[1] = rsv_blob1(r);
    co->n_items.fetch_add(cnt, std::memory_order_relaxed);
}

// the batch's callers have copied their rows in
void await_copies(CoBatch *b, bool may_sleep) {
    const size_t cnt = b->count;
    if (!may_sleep) {  // the dispatcher: walk the records; a writer not yet done is in the middle of a memcpy (or, rarely, descheduled)
        // (every record is a line last written by another core: most calls hold one item, so the lines of the next rows are asked for
        // ahead of the walk -- serially they were ~80 ns each, a third of a 128-call batch's turn-around)
        uint32_t calls = 0;
        for (size_t a = 0; a < std::min<size_t>(cnt, 12); a++) __builtin_prefetch(&b->recs[a], 0, 3);
        for (size_t pos = 0; pos < cnt; calls++) {
            if (pos + 12 < cnt) __builtin_prefetch(&b->recs[pos + 12], 0, 3);
            const CallRec &r = b->recs[pos];
            for (int i = 0; r.ready.load(std::memory_order_acquire) != b->agen; i++) {
                if (i < 4096) _mm_pause();
                else sched_yield();
            }
            pos += r.n;
        }
        b->ncalls = calls;
        return;
    }
    if (b->copied.load() == cnt) return;
    b->leader_waits.store(true);
    for (;;) {
        const uint32_t s = b->wseq.load();
        if (b->copied.load() == cnt) break;
        futex_op(&b->wseq, FUTEX_WAIT, s);
    }
    b->leader_waits.store(false);
}

// copies in (or zero-copy), the launch, copies out, the completion marker: everything of the batch is on its stream afterwards
int enqueue(Coalescer *co, CoBatch *b, const std::function<size_t(size_t)> &ws_bytes, const PipeOpts &opts, const std::function<int(Chunk &)> &launch) {
    const size_t cnt = b->count;
    b->wiped = false;
    for (size_t k = 0; k < co->nblob; k++) reinterpret_cast<uint64_t *>(b->hin + co->off_ofs[k])[cnt] = b->blob_used[k];
    size_t moved = 0;
    for (size_t k = 0; k < co->in_row.size(); k++) moved += co->in_row[k] * cnt;
    for (size_t k = 0; k < co->out_row.size(); k++) moved += co->out_row[k] * cnt;
    for (size_t k = 0; k < co->nblob; k++) moved += b->blob_used[k] + 8 * (cnt + 1);
    const bool zc = moved <= zero_copy_bytes() && b->hin_dev && b->hout_dev;
    b->zc = zc;
    b->wsb = std::min(co->ws_cap, up256(ws_bytes(cnt)));
    HIP_TRY(hipSetDevice(physical_device(co->dev)));
    uint8_t *in_base = zc ? b->hin_dev : b->d, *out_base = zc ? b->hout_dev : b->d + co->d_out_base;
    Chunk c;
    c.cnt = cnt; c.st = b->st;
    c.ws = b->d + co->ws_ofs; c.ws_bytes = b->wsb;
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
    // a zero-copy batch in completion mode 2 offers its flag to the launch itself (host_common.h TailOffer)
    const bool offer = zc && co->done_mode == 2 && b->flag_dev && opts.tail_flag_ok;
    g_tail_offer = offer ? TailOffer{b->flag_dev, b->count_dev, b->gen + 1, false} : TailOffer{};
    const int lrc = launch(c);
    const bool tail_taken = g_tail_offer.taken;
    g_tail_offer = TailOffer{};
    if (tail_taken) b->gen++;  // (whatever the launch reports afterwards: a kernel that took the offer will write this value -- the next use must not mistake it for its own)
    if (lrc) return lrc;
    if (tail_taken) {  // the launch raises the flag; it ran on page-locked rows and left nothing secret in the workspace
        b->wiped = true;
        return CIRCL_HIP_OK;
    }
    for (size_t k = 0; k < co->out_row.size() && !zc; k++)
        if (co->out_row[k]) HIP_TRY(hipMemcpyAsync(b->hout + co->out_ofs[k], b->d + co->d_out_base + co->out_ofs[k], co->out_row[k] * cnt, hipMemcpyDeviceToHost, b->st));
    if (co->done_mode && b->flag_dev) {
        b->gen++;
        if (co->done_mode == 1) HIP_TRY(hipStreamWriteValue32(b->st, b->flag_dev, b->gen, 0));
        else {
            WipeRanges w{};
            if (opts.wipe_device) {  // (what wipe_device() would otherwise enqueue as separate memsets)
                int k = 0;
                bool all = true;
                auto add = [&](uint8_t *p, size_t bytes) {
                    if (!bytes) return;
                    if (k < kWipeRanges) w.r[k++] = {p, (bytes + 15) / 16 * 16};  // (rounded up inside the array's own 256-byte slack)
                    else all = false;
                };
                if (!zc) {
                    for (size_t i = 0; i < co->in_row.size(); i++)
                        if (co->in_secret[i]) add(b->d + co->in_ofs[i], co->in_row[i] * cnt);
                    for (size_t i = 0; i < co->out_row.size(); i++)
                        if (co->out_secret[i]) add(b->d + co->d_out_base + co->out_ofs[i], co->out_row[i] * cnt);
                }
                add(b->d + co->ws_ofs, opts.ws_secret_bytes ? std::min(b->wsb, opts.ws_secret_bytes(cnt)) : b->wsb);
                b->wiped = all;  // (more secret arrays than ranges: wipe_device() does it the old way afterwards)
            }
            hipLaunchKernelGGL(coalesce_finish_kernel, dim3(1), dim3(256), 0, b->st, w, b->flag_dev, b->gen);
            HIP_TRY(hipGetLastError());
        }
    }
    return CIRCL_HIP_OK;
}

inline bool flagged(const Coalescer *co, const CoBatch *b) { return co->done_mode && b->flag_dev; }
// has the stream finished the batch?  (never blocks)
bool batch_done(Coalescer *co, CoBatch *b) {
    if (flagged(co, b)) return b->flag->load(std::memory_order_acquire) == b->gen;
    const hipError_t e = hipStreamQuery(b->st);
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return false; }
    return true;  // done -- or failed, which await_done reports
}
// waits for it; returns the batch's final status
int await_done(Coalescer *co, CoBatch *b, int rc) {
    hipError_t se = hipSuccess;
    if (flagged(co, b) && rc == CIRCL_HIP_OK) {
        // poll the flag the stream writes behind the batch: no call into the runtime on the way (profiles/r06_one_call.txt).  Bounded: a
        // stream that faulted never writes it, so after ~2 ms the runtime is asked
        const uint64_t t0 = now_ns();
        for (uint32_t i = 0;; i++) {
            if (b->flag->load(std::memory_order_acquire) == b->gen) break;
            _mm_pause();
            if ((i & 1023) == 1023 && now_ns() - t0 > 2000000ull) { se = hipStreamSynchronize(b->st); break; }
        }
        if ((++b->uses & 63) == 0) (void)hipStreamQuery(b->st);  // lets the runtime retire the stream's finished commands now and then
    } else {
        se = hipStreamSynchronize(b->st);
    }
    if (se != hipSuccess && rc == CIRCL_HIP_OK) {
        rc = CIRCL_HIP_EHIP;
        g_err = "coalesced batch: the stream failed";
        (void)hipGetLastError();
    }
    return rc;
}

// nothing secret stays in the device staging (enqueued behind the results; the batch's next use is on the same stream)
void wipe_device(Coalescer *co, CoBatch *b, const PipeOpts &opts) {
    if (!opts.wipe_device || b->wiped) return;
    const size_t cnt = b->count;
    if (!b->zc) {
        for (size_t k = 0; k < co->in_row.size(); k++)
            if (co->in_secret[k] && co->in_row[k]) (void)hipMemsetAsync(b->d + co->in_ofs[k], 0, co->in_row[k] * cnt, b->st);
        for (size_t k = 0; k < co->out_row.size(); k++)
            if (co->out_secret[k] && co->out_row[k]) (void)hipMemsetAsync(b->d + co->d_out_base + co->out_ofs[k], 0, co->out_row[k] * cnt, b->st);
    }
    const size_t sec = opts.ws_secret_bytes ? std::min(b->wsb, opts.ws_secret_bytes(cnt)) : b->wsb;
    if (sec) (void)hipMemsetAsync(b->d + co->ws_ofs, 0, sec, b->st);
    (void)hipGetLastError();
}
void wipe_host_rows(Coalescer *co, CoBatch *b) {
    for (size_t k = 0; k < co->in_row.size(); k++)
        if (co->in_secret[k]) memset(b->hin + co->in_ofs[k], 0, co->in_row[k] * b->count);
    for (size_t k = 0; k < co->out_row.size(); k++)
        if (co->out_secret[k]) memset(b->hout + co->out_ofs[k], 0, co->out_row[k] * b->count);
}

// the blocking leader's part: returns once the batch's results are in its page-locked output area (or b->rc says why not)
void flush(Coalescer *co, CoBatch *b, const std::function<size_t(size_t)> &ws_bytes, const PipeOpts &opts, const std::function<int(Chunk &)> &launch) {
    const bool stamps = g_stamps_on.load(std::memory_order_relaxed);
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
    close_batch(co, b);
    co->lock.unlock();
    bump(co);  // (the next batch's leader may now be first in line)
    STAMP(closed);
    await_copies(b, true);
    STAMP(copies_in);
    g_err.clear();
    int rc = enqueue(co, b, ws_bytes, opts, launch);
    STAMP(launched);
    rc = await_done(co, b, rc);
    STAMP(done);
    b->rc = rc;
    b->err = g_err;
    co->n_launches.fetch_add(1, std::memory_order_relaxed);
    co->lock.lock();
    co->inflight--;
    co->lock.unlock();
    bump(co);         // the next leader goes first: its launch overlaps the wake-ups and copies of this batch's callers
    b->done.open();
    wipe_device(co, b, opts);
}

// the last caller out: wipe the page-locked rows that held secrets, hand the batch back
void recycle(Coalescer *co, CoBatch *b) {
    wipe_host_rows(co, b);
    co->n_calls.fetch_add(b->callers.load(), std::memory_order_relaxed);  // (the statistics: once per batch)
    co->lock.lock();
    b->state = CoBatch::FREE;
    co->lock.unlock();
    bump(co);
}

void wake_dispatcher(Coalescer *co) {
    if (co->disp_sleeping.load()) {  // (seq_cst against the dispatcher's store / re-check: see async_main)
        co->disp_word.fetch_add(1);
        futex_op(&co->disp_word, FUTEX_WAKE, 1);
    }
}

// ---- reserve rows: one compare-and-swap on the open batch's reservation word; the lock only to open a new batch ----
// Returns the batch (rows [pos, pos + n), blob bytes from bpos[]), sets `opened` when this call opened it.  may_wait = false
// (asynchronous submitters): nullptr instead of sleeping when every batch is busy.
CoBatch *reserve(Coalescer *co, size_t n, const size_t (&bb)[kMaxBlobs], size_t &pos, size_t (&bpos)[kMaxBlobs], bool &opened, bool may_wait) {
    opened = false;
    for (;;) {
        CoBatch *b = co->open.load(std::memory_order_acquire);
        bool crowded = false;
        if (b) {
            uint64_t r = b->rsv.load(std::memory_order_relaxed);
            while (!(r & kRsvClosed)) {
                if (rsv_items(r) + n > co->max_items || rsv_blob0(r) + bb[0] > co->blob_cap || rsv_blob1(r) + bb[1] > co->blob_cap) { crowded = true; break; }
                if (b->rsv.compare_exchange_weak(r, r + rsv_pack(n, bb[0], bb[1]), std::memory_order_acq_rel, std::memory_order_relaxed)) {
                    pos = rsv_items(r);
                    bpos[0] = rsv_blob0(r);
                    bpos[1] = rsv_blob1(r);
                    if (pos + n >= co->max_items && !b->full.exchange(true)) {  // filled up: whoever flushes it need not wait any longer
                        if (co->async) wake_dispatcher(co);
                        else bump(co);
                    }
                    return b;
                }
            }
        }
        // no open batch, or this call does not fit into it: open the next one (under the lock: once per batch, not per call)
        co->lock.lock();
        CoBatch *cur = co->open.load();
        if (cur != b) { co->lock.unlock(); continue; }  // somebody else already did
        if (b && crowded && !b->full.exchange(true)) {  // it is flushed as soon as it may be
            co->lock.unlock();
            if (co->async) wake_dispatcher(co);
            else bump(co);
            co->lock.lock();
            if (co->open.load() != b) { co->lock.unlock(); continue; }
        }
        CoBatch *fresh = nullptr;
        for (CoBatch *c : co->batches)
            if (c->state == CoBatch::FREE) { fresh = c; break; }
        if (!fresh) {  // every batch is busy
            const uint32_t sq = co->seq.load();
            co->lock.unlock();
            if (!may_wait) return nullptr;
            co->seq_waiters.fetch_add(1);
            futex_op(&co->seq, FUTEX_WAIT, sq);
            co->seq_waiters.fetch_sub(1);
            continue;
        }
        fresh->state = CoBatch::OPEN;
        fresh->ticket = co->next_ticket++;
        fresh->full.store(n >= co->max_items);
        fresh->copied.store(0);
        fresh->returned.store(0);
        fresh->callers.store(0);
        fresh->rc = 0;
        fresh->done.reset();
        fresh->opened = Clock::now();
        fresh->agen = co->next_agen++;
        fresh->rsv.store(rsv_pack(n, bb[0], bb[1]), std::memory_order_release);  // open, with this call's rows at its head
        co->open.store(fresh, std::memory_order_release);
        if (co->async) {
            co->unclosed.push_back(fresh);
            co->n_unclosed.fetch_add(1);
        }
        co->lock.unlock();
        pos = 0;
        bpos[0] = bpos[1] = 0;
        opened = true;
        return fresh;
    }
}

// copy this call's rows into the batch's page-locked staging
void copy_in(Coalescer *co, CoBatch *b, size_t n, size_t pos, const size_t (&bb)[kMaxBlobs], const size_t (&bpos)[kMaxBlobs], const std::vector<HIn> &ins,
             const std::vector<HBlob> &blobs) {
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
}

// what every entry checks first: the call can join this coalescer at all
int admissible(Coalescer *co, size_t n, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs, size_t (&bb)[kMaxBlobs]) {
    if (!same_shape(co, ins, blobs, outs)) return kNotCoalesced;
    for (size_t k = 0; k < ins.size(); k++)  // a NULL pointer is rows of zeros only where the entry point said so (an absent key_idx / rnd)
        if (!ins[k].p && ins[k].row && !ins[k].optional) { g_err = "a required input pointer is NULL"; return CIRCL_HIP_EPARAM; }
    for (size_t k = 0; k < blobs.size(); k++) {
        if (blobs[k].blob && !blobs[k].off) { g_err = "a blob without offsets"; return CIRCL_HIP_EPARAM; }
        bb[k] = blobs[k].blob ? (size_t)(blobs[k].off[n] - blobs[k].off[0]) : 0;
        if (bb[k] > co->blob_cap / 4) return kNotCoalesced;  // a long message: its own call
    }
    return CIRCL_HIP_OK;
}

struct ActiveCall {
    Coalescer *co;  // nullptr: the owner counts (a key table's TableUse is held around every call into its coalescer)
    explicit ActiveCall(Coalescer *c) : co(c->owner_counts ? nullptr : c) { if (co) co->active.fetch_add(1); }
    ~ActiveCall() { if (co) co->active.fetch_sub(1); }
};

// ---- the dispatcher of the asynchronous form ----
void complete(Coalescer *co, CoBatch *b, int rc) {
    const uint32_t ncalls = b->ncalls;
    for (size_t pos = 0; pos < b->count;) {
        const CallRec &r = b->recs[pos];
        for (size_t k = 0; k < co->out_row.size(); k++) {
            if (!r.out[k] || !co->out_row[k]) continue;
            if (rc == CIRCL_HIP_OK) memcpy(r.out[k], b->hout + co->out_ofs[k] + pos * co->out_row[k], co->out_row[k] * r.n);
            else memset(r.out[k], 0, co->out_row[k] * r.n);  // a failed batch hands out nothing (and circl_hip_poll says why)
        }
        pos += r.n;
    }
    co->n_calls.fetch_add(ncalls, std::memory_order_relaxed);
    co->n_launches.fetch_add(1, std::memory_order_relaxed);
    const uint64_t ticket = b->ticket;
    co->rc_ring[ticket % kRcRing].store(rc, std::memory_order_relaxed);
    if (rc != CIRCL_HIP_OK) co->n_failed.fetch_add(1);
    // publish first (results are in the callers' buffers: a poller that sees the ticket done reads its rows), tidy up afterwards
    co->completed.store(ticket + 1, std::memory_order_release);
    co->done_word.fetch_add(1);
    if (co->done_waiters.load() > 0) futex_op(&co->done_word, FUTEX_WAKE, INT_MAX);
    if (co->efd >= 0) {
        const uint64_t one = 1;
        (void)!write(co->efd, &one, sizeof one);
    }
    wipe_device(co, b, co->a_opts);  // (nothing to do when the stream's finish kernel already did it)
    wipe_host_rows(co, b);
    co->lock.lock();
    b->state = CoBatch::FREE;
    co->inflight--;
    co->lock.unlock();
    co->seq.fetch_add(1);  // (blocking submitters waiting for a free batch: they re-read `seq` before they sleep)
    if (co->seq_waiters.load() > 0) futex_op(&co->seq, FUTEX_WAKE, INT_MAX);
}

void async_main(Coalescer *co) {
    (void)hipSetDevice(physical_device(co->dev));
    pin_to(dev_info(co->dev).cpus);  // (the device's NUMA node: the staging it reads and writes lives there)
    // ONE thread closes, launches AND copies finished batches out.  Splitting it -- a launcher and a completer side by side -- was built and
    // measured (profiles/r06_async_ab.txt): no gain at two batches in flight (-5...-10 %), level at three, for twice the polling CPU: the
    // queue is a closed loop bound by a batch's own latency (launch + kernel + flag ~ 30 us, copy-out 10-20), not by this loop's turn-around.
    std::deque<CoBatch *> fifo;  // launched, oldest first
    uint64_t idle_since = 0;
    for (;;) {
        bool progressed = false;
        // ---- launch: the oldest open batch, whenever the device has room (so a batch holds what arrived while its predecessors ran) ----
        if ((int)fifo.size() < co->inflight_max && co->n_unclosed.load(std::memory_order_acquire) > 0) {
            CoBatch *b = nullptr;
            co->lock.lock();
            if (!co->unclosed.empty()) {
                CoBatch *c = co->unclosed.front();
                const bool ripe = co->max_wait_us == 0 || c->full.load() || co->stop.load() ||
                                  Clock::now() >= c->opened + std::chrono::microseconds(co->max_wait_us);
                if (ripe) {
                    co->unclosed.pop_front();
                    co->n_unclosed.fetch_sub(1);
                    close_batch(co, c);
                    b = c;
                }
            }
            co->lock.unlock();
            if (b) {
                await_copies(b, false);
                g_err.clear();
                b->rc = enqueue(co, b, co->a_ws, co->a_opts, co->a_launch);
                fifo.push_back(b);
                progressed = true;
            }
        }
        // ---- complete: the oldest launched batch, while the younger ones run ----
        if (!fifo.empty()) {
            CoBatch *b = fifo.front();
            if (b->rc != CIRCL_HIP_OK || batch_done(co, b)) {
                const int rc = await_done(co, b, b->rc);
                complete(co, b, rc);
                fifo.pop_front();
                progressed = true;
            }
        }
        if (progressed) { idle_since = 0; continue; }
        if (!fifo.empty()) { _mm_pause(); continue; }  // the device is working on something: poll (tens of microseconds)
        if (co->stop.load() && co->n_unclosed.load() == 0) break;
        // ---- nothing in flight: poll briefly for new work, then sleep until a submitter opens a batch (or a lingering one is due) ----
        const uint64_t t = now_ns();
        if (!idle_since) idle_since = t;
        if (t - idle_since < (uint64_t)co->spin_us * 1000ull) { _mm_pause(); continue; }
        co->disp_sleeping.store(1);
        const uint32_t w = co->disp_word.load();
        bool sleep = !co->stop.load(), timed = false;
        timespec ts{0, 0};
        if (co->n_unclosed.load() > 0) {  // only a batch that lingers for company can be here: sleep until it is due
            co->lock.lock();
            if (!co->unclosed.empty()) {
                CoBatch *c = co->unclosed.front();
                const auto left = std::chrono::duration_cast<std::chrono::nanoseconds>(c->opened + std::chrono::microseconds(co->max_wait_us) - Clock::now()).count();
                if (left <= 0 || c->full.load() || co->max_wait_us == 0) sleep = false;
                else { ts.tv_sec = (time_t)(left / 1000000000); ts.tv_nsec = (long)(left % 1000000000); timed = true; }
            }
            co->lock.unlock();
        }
        if (sleep) futex_op(&co->disp_word, FUTEX_WAIT, w, timed ? &ts : nullptr);
        co->disp_sleeping.store(0);
        idle_since = 0;
    }
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
    co->done_mode = env_int("CIRCL_HIP_COALESCE_DONE", 2, 0, 2);  // (measured: profiles/r06_one_call.txt -- one caller 31 -> 24 us, 64 callers +11 %)
    co->spin_us = (unsigned)env_int("CIRCL_HIP_ASYNC_SPIN_US", 20, 0, 100000);
    for (auto &r : co->rc_ring) r.store(0, std::memory_order_relaxed);
    return co;
}
// nothing of a caller is inside, no batch is open or running
bool coalescer_idle(Coalescer *co) {
    if (!co) return true;
    if (co->active.load() != 0) return false;
    co->lock.lock();
    bool idle = true;
    for (CoBatch *b : co->batches)
        if (b->state != CoBatch::FREE) idle = false;
    co->lock.unlock();
    return idle;
}
void coalescer_free(Coalescer *co) {
    if (!co) return;
    if (co->async) {  // the dispatcher flushes what is open, finishes what runs, and leaves
        co->stop.store(true);
        co->disp_sleeping.store(1);  // (force the wake: the flag is only a hint)
        wake_dispatcher(co);
        if (co->disp.joinable()) co->disp.join();
        if (co->efd >= 0) close(co->efd);
    }
    // (the owner made sure no call is inside: circl_hip_keytable_set_coalesce / _free freeze the table and check coalescer_idle first)
    free_batches(co);
    delete co;
}
size_t coalescer_call_max(const Coalescer *co) { return co ? co->call_max : 0; }
void coalescer_stats(const Coalescer *co, uint64_t *calls, uint64_t *items, uint64_t *launches) {
    if (calls) *calls = co ? co->n_calls.load() : 0;
    if (items) *items = co ? co->n_items.load() : 0;
    if (launches) *launches = co ? co->n_launches.load() : 0;
}
bool coalescer_is_async(const Coalescer *co) { return co && co->async; }
int coalescer_eventfd(const Coalescer *co) { return co ? co->efd : -1; }

int coalescer_async_start(Coalescer *co, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs,
                          const std::function<size_t(size_t)> &ws_bytes, const PipeOpts &opts, const std::function<int(Chunk &)> &launch, bool want_eventfd) {
    if (!co || co->async || co->ready.load() != 0) return CIRCL_HIP_EPARAM;
    co->async = true;
    co->a_launch = launch;
    co->a_ws = ws_bytes;
    co->a_opts = opts;
    const int rc = ensure_layout(co, ins, blobs, outs, ws_bytes);
    if (rc != CIRCL_HIP_OK) { g_err = co->init_err; co->async = false; return rc; }
    if (want_eventfd) co->efd = eventfd(0, EFD_NONBLOCK | EFD_CLOEXEC);
    co->disp = std::thread(async_main, co);
    return CIRCL_HIP_OK;
}

int coalesce_submit(Coalescer *co, size_t n, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs, uint64_t *seq,
                    bool may_wait) {
    if (seq) *seq = 0;
    if (!co || !co->async || !seq) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;  // (ticket 0 is always done)
    if (n > co->call_max) { g_err = "more items than a submitted call may hold (max_items / 4)"; return CIRCL_HIP_EPARAM; }
    ActiveCall guard(co);
    size_t bb[kMaxBlobs] = {0, 0};
    if (int rc = admissible(co, n, ins, blobs, outs, bb)) {
        if (rc == kNotCoalesced) { g_err = "this call cannot join the table's queue (a message beyond a quarter of the blob area, or other arrays than the queue's)"; return CIRCL_HIP_EPARAM; }
        return rc;
    }
    size_t pos = 0, bpos[kMaxBlobs] = {0, 0};
    bool opened = false;
    CoBatch *b = reserve(co, n, bb, pos, bpos, opened, may_wait);
    if (!b) return CIRCL_HIP_EAGAIN;
    // (ticket and generation are stable: the batch cannot be recycled before this call's record is stamped)
    const uint64_t ticket = b->ticket + 1, gen = b->agen;
    CallRec &rec = b->recs[pos];
    rec.n = (uint32_t)n;
    for (size_t k = 0; k < (size_t)kMaxOuts; k++) rec.out[k] = k < outs.size() ? outs[k].p : nullptr;
    copy_in(co, b, n, pos, bb, bpos, ins, blobs);
    rec.ready.store(gen, std::memory_order_release);  // the record and the rows are the dispatcher's to read from here on
    if (opened) wake_dispatcher(co);
    *seq = ticket;
    return CIRCL_HIP_OK;
}
int coalescer_state(const Coalescer *co, uint64_t seq) {
    if (!co || !co->async) return CIRCL_HIP_EPARAM;
    if (seq == 0) return 1;
    const uint64_t done = co->completed.load(std::memory_order_acquire);
    if (seq > done) return seq <= done + (uint64_t)co->batches.size() + 1 ? 0 : CIRCL_HIP_EPARAM;  // (a ticket this queue never issued)
    if (done - seq < (uint64_t)kRcRing) {
        const int rc = co->rc_ring[(seq - 1) % kRcRing].load(std::memory_order_relaxed);
        return rc == CIRCL_HIP_OK ? 1 : rc;
    }
    return co->n_failed.load() ? CIRCL_HIP_EHIP : 1;  // older than the ring: done; failed if ANY batch of this queue ever failed (fail closed)
}
int coalescer_wait(Coalescer *co, uint64_t seq, int64_t timeout_us) {
    if (!co || !co->async) return CIRCL_HIP_EPARAM;
    const uint64_t t_end = timeout_us < 0 ? ~0ull : now_ns() + (uint64_t)timeout_us * 1000ull;
    for (int spins = 0;; spins++) {
        const int s = coalescer_state(co, seq);
        if (s != 0) return s;
        if (spins < 64) { _mm_pause(); continue; }
        const uint64_t t = now_ns();
        if (t >= t_end) return 0;
        co->done_waiters.fetch_add(1);
        const uint32_t w = co->done_word.load();
        if (coalescer_state(co, seq) == 0) {
            if (timeout_us < 0) futex_op(&co->done_word, FUTEX_WAIT, w);
            else {
                const uint64_t left = t_end - t;
                timespec ts{(time_t)(left / 1000000000ull), (long)(left % 1000000000ull)};
                futex_op(&co->done_word, FUTEX_WAIT, w, &ts);
            }
        }
        co->done_waiters.fetch_sub(1);
    }
}

int coalesce_run(Coalescer *co, size_t n, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs,
                 const std::function<size_t(size_t)> &ws_bytes, const PipeOpts &opts, const std::function<int(Chunk &)> &launch) {
    if (n == 0) return CIRCL_HIP_OK;
    if (!co || n > co->call_max || blobs.size() > (size_t)kMaxBlobs || outs.size() > (size_t)kMaxOuts) return kNotCoalesced;
    if (co->async) {  // a blocking call through a table whose queue is asynchronous: submit, then wait for the ticket
        uint64_t seq = 0;
        const int rc = coalesce_submit(co, n, ins, blobs, outs, &seq, true);
        if (rc != CIRCL_HIP_OK) return rc;
        const int s = coalescer_wait(co, seq, -1);
        return s == 1 ? CIRCL_HIP_OK : s;
    }
    const bool stamps = g_stamps_on.load(std::memory_order_relaxed);
    if (stamps) g_stamps = CallStamps{};
    STAMP(enter);
    ActiveCall guard(co);
    if (int rc = ensure_layout(co, ins, blobs, outs, ws_bytes)) return rc;
    size_t bb[kMaxBlobs] = {0, 0};
    if (int rc = admissible(co, n, ins, blobs, outs, bb)) return rc;

    size_t pos = 0, bpos[kMaxBlobs] = {0, 0};
    bool leader = false;
    CoBatch *b = reserve(co, n, bb, pos, bpos, leader, true);
    b->callers.fetch_add(1, std::memory_order_relaxed);
    STAMP(reserved);
    copy_in(co, b, n, pos, bb, bpos, ins, blobs);
    b->copied.fetch_add(n);
    b->wseq.fetch_add(1);
    if (b->leader_waits.load()) futex_op(&b->wseq, FUTEX_WAKE, 1);  // (seq_cst on both sides: a leader that missed the count sees the new wseq)
    STAMP(copied_in);

    if (leader) flush(co, b, ws_bytes, opts, launch);
    else b->done.wait(0);  // (polling the gate before sleeping was measured in round 5 and lost: callers may outnumber the CPUs)

    // ---- results ----
    const int rc = b->rc;
    if (rc) g_err = b->err;
    else
        for (size_t k = 0; k < outs.size(); k++)
            if (outs[k].p && outs[k].row) memcpy(outs[k].p, b->hout + co->out_ofs[k] + pos * outs[k].row, outs[k].row * n);
    STAMP(copied_out);
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
        co = coalescer_new(dev, items, g_call_wait.load());  // (never freed: a call may hold the pointer at any time; switched off = not handed out)
        g_call[op][param_slot][dev].store(co, std::memory_order_release);
    }
    return co;
}
int call_coalescing_set(size_t max_items, uint32_t max_wait_us) {
    g_call_wait.store(max_wait_us);
    g_call_items.store(max_items, std::memory_order_release);
    if (max_items != 0) return CIRCL_HIP_OK;
    // switched off: DRAIN -- return once no call is inside any of the process-wide coalescers and none of their batches is open or
    // running (calls that arrive from now on take the un-coalesced path; one that read the old setting a moment ago is waited for)
    const uint64_t t_end = now_ns() + 5000000000ull;
    for (int op = 0; op < kCoOps; op++)
        for (int p = 0; p < kCallParams; p++)
            for (int d = 0; d < kCallDevs; d++) {
                Coalescer *co = g_call[op][p][d].load(std::memory_order_acquire);
                while (co && !coalescer_idle(co)) {
                    if (now_ns() > t_end) { g_err = "circl_hip_set_coalesce(0): calls still in flight after 5 s"; return CIRCL_HIP_EBUSY; }
                    sched_yield();
                }
            }
    return CIRCL_HIP_OK;
}

}  // namespace host
}  // namespace circl

// ---- C ABI -----------------------------------------------------------------------------------------------------------------
#include "keytable.h"
using namespace circl::host;

namespace {
// the tables a setter has to freeze: the table itself, or every replica of a replicated one
std::vector<circl_hip_keytable *> parts_of(circl_hip_keytable *t) {
    std::vector<circl_hip_keytable *> v;
    if (t->device < 0) for (int d = 0; d < t->nreplica; d++) v.push_back(t->replica[d]);
    else v.push_back(t);
    return v;
}
// Freezes every part (calls that arrive now skip the coalescer) and checks that nothing is inside one and no batch is open or running.
// On success the parts STAY frozen (the caller swaps the coalescers and thaws); on failure they are thawed and EBUSY is returned.
int freeze_parts(circl_hip_keytable *t, std::vector<circl_hip_keytable *> &parts) {
    parts = parts_of(t);
    if (t->device < 0) t->frozen.store(true);
    for (auto *r : parts) r->frozen.store(true);
    // (an asynchronous queue may hold submitted calls nobody is inside the library for: coalescer_free finishes them -- its dispatcher
    // drains before it leaves -- so only CALLERS INSIDE make such a table busy)
    bool busy = t->device < 0 && t->users.load() != 0;
    for (auto *r : parts) busy = busy || r->users.load() != 0 || (r->coalescer && !coalescer_is_async(r->coalescer) && !coalescer_idle(r->coalescer));
    if (busy) {
        for (auto *r : parts) r->frozen.store(false);
        if (t->device < 0) t->frozen.store(false);
        g_err = "the table has calls in flight";
        return CIRCL_HIP_EBUSY;
    }
    return CIRCL_HIP_OK;
}
void thaw_parts(circl_hip_keytable *t, const std::vector<circl_hip_keytable *> &parts) {
    for (auto *r : parts) r->frozen.store(false);
    if (t->device < 0) t->frozen.store(false);
}
}  // namespace

namespace circl {
namespace host {
// circl_hip_keytable_free's part: quiesce (bounded), then release the coalescers.  false: still busy after the bound (the caller leaks
// the table rather than free memory under running calls).
bool keytable_quiesce(circl_hip_keytable *t) {
    std::vector<circl_hip_keytable *> parts;
    const uint64_t t_end = now_ns() + 2000000000ull;
    // (an asynchronous queue with tickets outstanding finishes them in coalescer_free: its dispatcher drains before it exits -- only
    // CALLERS INSIDE the library are waited for here)
    for (;;) {
        parts = parts_of(t);
        if (t->device < 0) t->frozen.store(true);
        for (auto *r : parts) r->frozen.store(true);
        bool busy = t->device < 0 && t->users.load() != 0;
        for (auto *r : parts) busy = busy || r->users.load() != 0 || (r->coalescer && !coalescer_is_async(r->coalescer) && !coalescer_idle(r->coalescer));
        if (!busy) return true;
        if (now_ns() > t_end) return false;
        sched_yield();
    }
}
}  // namespace host
}  // namespace circl

extern "C" {

int circl_hip_keytable_set_coalesce(circl_hip_keytable *t, size_t max_items, uint32_t max_wait_us) {
    if (!t || t->magic != kKeytableMagic) return CIRCL_HIP_EPARAM;
    std::vector<circl_hip_keytable *> parts;
    if (int rc = freeze_parts(t, parts)) return rc;  // CIRCL_HIP_EBUSY while calls are in flight: nothing is freed under a caller
    int rc = CIRCL_HIP_OK;
    for (auto *r : parts) {
        if (r->coalescer) { coalescer_free(r->coalescer); r->coalescer = nullptr; }
        if (max_items == 0) continue;
        r->coalescer = coalescer_new(r->device, max_items, max_wait_us);
        if (!r->coalescer) rc = CIRCL_HIP_ENOMEM;
        else r->coalescer->owner_counts = true;  // (every call into it holds a TableUse: keytable.h)
    }
    thaw_parts(t, parts);
    return rc;
}
int circl_hip_set_coalesce(size_t max_items, uint32_t max_wait_us) { return circl::host::call_coalescing_set(max_items, max_wait_us); }
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

// ---- the asynchronous form ----
int circl_hip_keytable_async_start(circl_hip_keytable *t, size_t max_items, uint32_t max_wait_us, int want_eventfd) {
    if (!t || t->magic != kKeytableMagic || max_items == 0) return CIRCL_HIP_EPARAM;
    if (!(t->family == 1 || t->family == 3 || (t->family == 2 && !t->private_keys))) { g_err = "asynchronous queues serve ML-KEM tables, hybrid KEM tables and ML-DSA public-key tables"; return CIRCL_HIP_EPARAM; }
    std::vector<circl_hip_keytable *> parts;
    if (int rc = freeze_parts(t, parts)) return rc;
    int rc = CIRCL_HIP_OK;
    for (auto *r : parts) {
        if (r->coalescer) { coalescer_free(r->coalescer); r->coalescer = nullptr; }
        Coalescer *co = coalescer_new(r->device, max_items, max_wait_us);
        if (!co) { rc = CIRCL_HIP_ENOMEM; break; }
        co->owner_counts = true;
        rc = r->family == 1 ? kem_table_async_start(r, co, want_eventfd != 0)
                            : r->family == 3 ? hyb_table_async_start(r, co, want_eventfd != 0) : dsa_table_async_start(r, co, want_eventfd != 0);
        if (rc != CIRCL_HIP_OK) { coalescer_free(co); break; }
        r->coalescer = co;
    }
    if (rc != CIRCL_HIP_OK)
        for (auto *r : parts)
            if (r->coalescer) { coalescer_free(r->coalescer); r->coalescer = nullptr; }
    thaw_parts(t, parts);
    return rc;
}
int circl_hip_keytable_async_stop(circl_hip_keytable *t) { return circl_hip_keytable_set_coalesce(t, 0, 0); }
int circl_hip_keytable_eventfd(const circl_hip_keytable *t, int replica) {
    if (!t || t->magic != kKeytableMagic) return -1;
    const circl_hip_keytable *r = t->device < 0 ? (replica >= 0 && replica < t->nreplica ? t->replica[replica] : nullptr) : (replica == 0 ? t : nullptr);
    return r ? coalescer_eventfd(r->coalescer) : -1;
}
static Coalescer *ticket_queue(const circl_hip_keytable *t, uint64_t ticket, uint64_t *seq) {
    if (!t || t->magic != kKeytableMagic) return nullptr;
    const int rep = (int)(ticket >> 56);
    *seq = ticket & ((1ull << 56) - 1);
    const circl_hip_keytable *r = t->device < 0 ? (rep < t->nreplica ? t->replica[rep] : nullptr) : (rep == 0 ? t : nullptr);
    return r && coalescer_is_async(r->coalescer) ? r->coalescer : nullptr;
}
// poll and wait are calls INSIDE the table like any other (a setter that freed the queue under them would leave them reading freed memory):
// they count themselves (TableUse) and, should a setter have the table frozen at that moment, report "pending" -- the setter sees them,
// answers CIRCL_HIP_EBUSY and thaws; the next poll reads the queue again.
int circl_hip_poll(const circl_hip_keytable *t, const uint64_t *tickets, size_t n, int8_t *state) {
    if (n && (!tickets || !state)) return CIRCL_HIP_EPARAM;
    if (!t || t->magic != kKeytableMagic) {
        for (size_t i = 0; i < n; i++) state[i] = (int8_t)CIRCL_HIP_EPARAM;
        return (int)n;
    }
    TableUse use(t);
    if (t->frozen.load()) {
        for (size_t i = 0; i < n; i++) state[i] = 0;
        return 0;
    }
    int done = 0;
    for (size_t i = 0; i < n; i++) {
        uint64_t seq = 0;
        Coalescer *co = ticket_queue(t, tickets[i], &seq);
        const int s = co ? coalescer_state(co, seq) : CIRCL_HIP_EPARAM;
        state[i] = (int8_t)s;
        done += s != 0;
    }
    return done;
}
int circl_hip_wait(const circl_hip_keytable *t, uint64_t ticket, int64_t timeout_us) {
    if (!t || t->magic != kKeytableMagic) return CIRCL_HIP_EPARAM;
    TableUse use(t);
    if (t->frozen.load()) return 0;  // (a setter is looking at the table right now: it will find this call and back off)
    uint64_t seq = 0;
    Coalescer *co = ticket_queue(t, ticket, &seq);
    if (!co) return CIRCL_HIP_EPARAM;
    return coalescer_wait(co, seq, timeout_us);
}

int circl_hip_profile_call_stamps(int enable, uint64_t *out8) {
    if (enable >= 0) g_stamps_on.store(enable != 0);
    if (out8) {
        const CallStamps &s = g_stamps;
        out8[0] = s.enter; out8[1] = s.reserved; out8[2] = s.copied_in; out8[3] = s.closed;
        out8[4] = s.copies_in; out8[5] = s.launched; out8[6] = s.done; out8[7] = s.copied_out;
    }
    return CIRCL_HIP_OK;
}

}  // extern "C"

// ---- circl_hip_queue: the asynchronous form for keys that come WITH the call -------------------------------------------------------------
// A queue is a coalescer with a dispatcher (above) that belongs to no table: one operation, one parameter set, one device.  Lifetime as for
// tables: every call counts itself inside; close refuses (CIRCL_HIP_EBUSY) while one is, finishes what was submitted, then frees.
struct circl_hip_queue {
    uint32_t magic;
    int op, param, device;
    Coalescer *co;
    QueueShape sh;
    std::atomic<bool> closing{false};
    mutable UseCount users;
};
namespace {
constexpr uint32_t kQueueMagic = 0x51554531u;  // "QUE1"
struct QueueUse {
    const circl_hip_queue *q;
    int s;
    explicit QueueUse(const circl_hip_queue *qq) : q(qq), s(use_slot()) { q->users.slot[s].n.fetch_add(1); }
    ~QueueUse() { q->users.slot[s].n.fetch_sub(1); }
};
inline bool queue_ok(const circl_hip_queue *q) { return q && q->magic == kQueueMagic; }
}  // namespace

extern "C" {

int circl_hip_queue_open(int op, int param, int device, size_t max_items, int want_eventfd, circl_hip_queue **out) {
    if (out) *out = nullptr;
    if (!out || op < CIRCL_HIP_QUEUE_MLKEM_ENCAPS || op > CIRCL_HIP_QUEUE_HYBRID_DECAPS || max_items == 0) return CIRCL_HIP_EPARAM;
    if (device < 0 || device >= ndev()) return CIRCL_HIP_ENODEV;  // (a queue lives on ONE device: CIRCL_HIP_ALL_DEVICES is not accepted)
    circl_hip_queue *q = new (std::nothrow) circl_hip_queue();
    if (!q) return CIRCL_HIP_ENOMEM;
    q->magic = kQueueMagic; q->op = op; q->param = param; q->device = device;
    q->co = coalescer_new(device, max_items, 0);
    if (!q->co) { delete q; return CIRCL_HIP_ENOMEM; }
    q->co->owner_counts = true;  // (every call into it holds a QueueUse)
    const bool decaps = op == CIRCL_HIP_QUEUE_MLKEM_DECAPS || op == CIRCL_HIP_QUEUE_HYBRID_DECAPS;
    const int rc = op <= CIRCL_HIP_QUEUE_MLKEM_DECAPS ? kem_call_queue_start(decaps, param, q->co, want_eventfd != 0, &q->sh)
                                                      : hyb_call_queue_start(decaps, param, q->co, want_eventfd != 0, &q->sh);
    if (rc != CIRCL_HIP_OK) {
        coalescer_free(q->co);
        delete q;
        return rc;
    }
    *out = q;
    return CIRCL_HIP_OK;
}
int circl_hip_queue_close(circl_hip_queue *q) {
    if (!queue_ok(q)) return CIRCL_HIP_EPARAM;
    q->closing.store(true);  // (seq_cst against the callers' count: either they see it, or it sees them)
    if (q->users.load() != 0) {
        q->closing.store(false);
        g_err = "the queue has calls in flight";
        return CIRCL_HIP_EBUSY;
    }
    coalescer_free(q->co);  // finishes every submitted call first (the dispatcher drains before it leaves)
    q->magic = 0;
    delete q;
    return CIRCL_HIP_OK;
}
int circl_hip_queue_eventfd(const circl_hip_queue *q) { return queue_ok(q) ? coalescer_eventfd(q->co) : -1; }
int circl_hip_queue_stats(const circl_hip_queue *q, uint64_t *calls, uint64_t *items, uint64_t *launches) {
    if (!queue_ok(q)) return CIRCL_HIP_EPARAM;
    coalescer_stats(q->co, calls, items, launches);
    return CIRCL_HIP_OK;
}
int circl_hip_queue_submit(circl_hip_queue *q, const uint8_t *key, const uint8_t *in, uint8_t *out0, uint8_t *ss, uint8_t *status, size_t n, uint64_t *ticket) {
    if (ticket) *ticket = 0;
    if (!queue_ok(q) || !ticket) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    const bool encaps = q->sh.out0 != 0;
    if (!key || !in || !ss || (encaps && !out0)) { g_err = "a required pointer is NULL"; return CIRCL_HIP_EPARAM; }
    QueueUse use(q);
    if (q->closing.load()) return CIRCL_HIP_EPARAM;
    std::vector<HOut> outs;
    if (encaps) outs.push_back({out0, q->sh.out0});
    outs.push_back({ss, q->sh.ss, true});
    outs.push_back({status, 1});
    uint64_t seq = 0;
    const int rc = coalesce_submit(q->co, n, {{key, q->sh.key, q->sh.key_secret}, {in, q->sh.in, q->sh.in_secret}}, {}, outs, &seq, false);
    if (rc == CIRCL_HIP_OK) *ticket = seq;
    return rc;
}
int circl_hip_queue_poll(const circl_hip_queue *q, const uint64_t *tickets, size_t n, int8_t *state) {
    if (n && (!tickets || !state)) return CIRCL_HIP_EPARAM;
    if (!queue_ok(q)) {
        for (size_t i = 0; i < n; i++) state[i] = (int8_t)CIRCL_HIP_EPARAM;
        return (int)n;
    }
    QueueUse use(q);
    int done = 0;
    for (size_t i = 0; i < n; i++) {
        const int s = q->closing.load() ? 0 : coalescer_state(q->co, tickets[i]);
        state[i] = (int8_t)s;
        done += s != 0;
    }
    return done;
}
int circl_hip_queue_wait(const circl_hip_queue *q, uint64_t ticket, int64_t timeout_us) {
    if (!queue_ok(q)) return CIRCL_HIP_EPARAM;
    QueueUse use(q);
    if (q->closing.load()) return 0;
    return coalescer_wait(q->co, ticket, timeout_us);
}

}  // extern "C"

