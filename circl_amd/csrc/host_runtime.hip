// host_runtime.hip -- device table, profiling, byte-mover thread pools, staging slots and the host-buffer pipeline
// (declared in host_common.h).  No compute happens on the CPU here: the threads only copy bytes between the caller's
// memory and page-locked staging buffers.
#include "host_common.h"
#include "keytable.h"

#include <immintrin.h>
#include <pthread.h>
#include <sched.h>

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <sstream>

namespace circl {
namespace host {

thread_local std::string g_err;

// ---- devices ----------------------------------------------------------------------------------
int env_int(const char *name, int dflt, int lo, int hi) {
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    const int v = atoi(e);
    return v < lo || v > hi ? dflt : v;
}

// Page-locked staging memory.  ThreadSanitizer builds (tests/test_gpu_sanitizers.py) take it from mmap + hipHostRegister instead of
// hipHostMalloc: the ROCm runtime maps its own host memory behind TSan's back, so a staging buffer that lands on the address
// range of a finished thread's unmapped stack inherited that thread's access history, and the first write to it was reported as
// a race with the dead thread (seen with the no-worker configuration, one run in three).  TSan sees mmap and clears the range.
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define CIRCL_TSAN 1
#endif
#endif
#ifdef CIRCL_TSAN
}  // namespace host
}  // namespace circl
#include <sys/mman.h>
#include <map>
namespace circl {
namespace host {
static std::mutex g_pinned_mu;
static std::map<void *, size_t> g_pinned_sizes;
hipError_t pinned_alloc(void **p, size_t bytes) {
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return hipErrorOutOfMemory;
    const hipError_t e = hipHostRegister(m, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { munmap(m, bytes); return e; }
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    g_pinned_sizes[m] = bytes;
    *p = m;
    return hipSuccess;
}
hipError_t pinned_free(void *p) {
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        bytes = g_pinned_sizes[p];
        g_pinned_sizes.erase(p);
    }
    const hipError_t e = hipHostUnregister(p);
    munmap(p, bytes);
    return e;
}
#else
hipError_t pinned_alloc(void **p, size_t bytes) { return hipHostMalloc(p, bytes, hipHostMallocDefault); }  // default placement: the device's NUMA node
hipError_t pinned_free(void *p) { return hipHostFree(p); }
#endif
uint8_t *pinned_device_ptr(void *p) {
    void *dp = nullptr;
    if (!p || hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return static_cast<uint8_t *>(dp);
}
size_t zero_copy_bytes() {
    // measured (profiles/r05_zerocopy.txt): against enqueued copies a resident-key call of 64 items takes 41 instead of 64 us, of 256
    // items 64 instead of 90; the two meet between 2 and 4 MB (tools/route_check.py, profiles/r06_routes.txt: at 2 MB zero-copy still wins by
    // 14-16 %, at 4 MB an encapsulation is level and a decapsulation, which reads its ciphertext twice, loses 38 %)
    static const size_t v = (size_t)env_int("CIRCL_HIP_ZEROCOPY_KB", 2048, 0, 1 << 20) << 10;
    return v;
}

namespace {

struct DeviceTable {
    int n = 0;       // LOGICAL devices: what every index of the ABI, every per-device pool and shard() count in
    int nphys = 0;   // HIP devices behind them; logical device d runs on HIP device d % nphys
    std::vector<DeviceInfo> info;
    int max_cus = 256;
};
DeviceTable *g_devs = nullptr;
std::once_flag g_devs_once;

std::vector<int> parse_cpulist(const std::string &s) {  // "0-15,128-143"
    std::vector<int> out;
    std::stringstream ss(s);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
        if (tok.empty()) continue;
        const size_t dash = tok.find('-');
        const int lo = atoi(tok.c_str()), hi = dash == std::string::npos ? lo : atoi(tok.c_str() + dash + 1);
        for (int c = lo; c <= hi && c < CPU_SETSIZE; c++) out.push_back(c);
    }
    return out;
}

void init_devices() {
    auto *t = new DeviceTable;  // never freed: worker threads may outlive static destruction
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { n = 0; (void)hipGetLastError(); }
    t->nphys = n;
    // CIRCL_HIP_LOGICAL_DEVICES=L presents L devices whatever the box has (logical d -> HIP device d % n): a node with
    // one GPU then runs the all-devices branch of shard(), L staging pools, L sets of streams and L mover pools -- every
    // piece of per-device host state of an L-GPU node -- and an 8-GPU node can be driven as 16 half-batches.
    // (never fewer than the HIP devices: device-resident callers index per-device state with their current HIP device)
    const int logical = n > 0 ? std::max(n, env_int("CIRCL_HIP_LOGICAL_DEVICES", n, 1, 64)) : 0;
    t->n = logical;
    t->info.resize(n > 0 ? n : 0);
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    const bool have_aff = sched_getaffinity(0, sizeof allowed, &allowed) == 0;
    int mx = 0;
    for (int d = 0; d < n; d++) {
        DeviceInfo &di = t->info[d];
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && cus > 0) di.cus = cus;
        mx = std::max(mx, di.cus);
        char bdf[64] = {0};
        if (hipDeviceGetPCIBusId(bdf, sizeof bdf, d) == hipSuccess) {
            for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
            std::ifstream f(std::string("/sys/bus/pci/devices/") + bdf + "/numa_node");
            int node = -1;
            if (f >> node && node >= 0) {
                di.numa = node;
                std::ifstream g("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
                std::string list;
                if (std::getline(g, list))
                    for (int c : parse_cpulist(list))
                        if (!have_aff || CPU_ISSET(c, &allowed)) di.cpus.push_back(c);
            }
        }
    }
    if (mx > 0) t->max_cus = mx;
    for (int d = n; d < logical; d++) t->info.push_back(t->info[d % n]);
    g_devs = t;
}
const DeviceTable &devs() {
    std::call_once(g_devs_once, init_devices);
    return *g_devs;
}

}  // namespace

void pin_to(const std::vector<int> &cpus) {
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}

int ndev() { return devs().n; }
int physical_device(int dev) {
    const DeviceTable &t = devs();
    return t.nphys > 0 && dev >= 0 ? dev % t.nphys : 0;
}
const DeviceInfo &dev_info(int dev) {
    static const DeviceInfo fallback;
    const DeviceTable &t = devs();
    return dev >= 0 && dev < t.n ? t.info[dev] : fallback;
}
int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
    return d;
}
int max_cu_count() { return devs().max_cus; }

int usable_cpus() {
    static const int v = [] {
        cpu_set_t set;
        CPU_ZERO(&set);
        int n = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
        if (n < 1) n = 1;
        std::ifstream f("/sys/fs/cgroup/cpu.max");  // cgroup v2: "<quota> <period>" or "max <period>"
        std::string q;
        long period = 0;
        if (f >> q >> period) {
            if (q != "max" && period > 0) n = std::min<long>(n, std::max<long>(1, (atol(q.c_str()) + period - 1) / period));
        } else {
            std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
            long quota = -1;
            if ((fq >> quota) && (fp >> period) && quota > 0 && period > 0) n = std::min<long>(n, std::max<long>(1, (quota + period - 1) / period));
        }
        return n;
    }();
    return v;
}

// ---- occupancy cache, keyed on (device, kernel) --------------------------------------------------
unsigned resident_blocks_cached(int dev, const void *key, const std::function<int()> &query) {
    struct Entry { int dev; const void *key; unsigned blocks; };
    static std::mutex mu;
    static std::vector<Entry> cache;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &e : cache)
            if (e.dev == dev && e.key == key) return e.blocks;
    }
    int occ = query();  // outside the lock: the runtime call may be slow
    if (occ > kMaxBlocksPerCU) occ = kMaxBlocksPerCU;
    const unsigned v = (unsigned)(dev_info(dev).cus * occ);
    std::lock_guard<std::mutex> lk(mu);
    for (auto &e : cache)
        if (e.dev == dev && e.key == key) return e.blocks;
    cache.push_back({dev, key, v});
    return v;
}

// ---- kernel-level profiling ---------------------------------------------------------------
namespace {
struct ProfRec { int kernel; hipEvent_t a, b; };
std::mutex g_prof_mu;
std::atomic<bool> g_prof_on{false};
std::vector<ProfRec> g_prof_pending;
double g_prof_ms[CIRCL_HIP_KERNEL_COUNT];
uint64_t g_prof_n[CIRCL_HIP_KERNEL_COUNT];
}  // namespace
bool prof_on() { return g_prof_on.load(std::memory_order_relaxed); }
void prof_push(int kernel, hipEvent_t a, hipEvent_t b) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_pending.push_back({kernel, a, b});
}

// ---- byte-mover thread pools ------------------------------------------------------------------
namespace {

struct Batch {
    size_t n = 0;
    const std::function<void(size_t)> *fn = nullptr;
    std::atomic<size_t> next{0}, done{0};
    std::mutex mu;
    std::condition_variable cv;
};
struct Pool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Batch>> q;
    int nthreads = 0;
};
std::mutex g_pools_mu;
std::vector<Pool *> g_pools;  // one per device, created lazily, never destroyed

void worker_main(Pool *p, std::vector<int> cpus) {
    pin_to(cpus);
    for (;;) {
        std::shared_ptr<Batch> b;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return !p->q.empty(); });
            b = p->q.front();
        }
        const size_t i = b->next.fetch_add(1);
        if (i >= b->n) {  // exhausted: retire it so that the next batch becomes visible
            std::lock_guard<std::mutex> lk(p->mu);
            if (!p->q.empty() && p->q.front() == b) p->q.pop_front();
            continue;
        }
        (*b->fn)(i);
        if (b->done.fetch_add(1) + 1 == b->n) {
            std::lock_guard<std::mutex> lk(b->mu);
            b->cv.notify_all();
        }
    }
}

Pool *pool_of(int dev) {
    const int nd = std::max(ndev(), 1);
    if (dev < 0 || dev >= nd) dev = 0;
    std::lock_guard<std::mutex> lk(g_pools_mu);
    if (g_pools.empty()) g_pools.assign(nd, nullptr);
    if (!g_pools[dev]) {
        Pool *p = new Pool;
        // the calling thread works too, so a pool of T threads gives T + 1 movers; the CPUs are shared by all devices.  The copies are
        // memory-bound (profiles/r05_logical8.txt, 2^20 ML-KEM-768 encapsulations from byte-misaligned pageable arrays): 2 movers 3.42e7/s
        // with 3.6 CPUs busy, 4 3.82e7 / 4.8, 8 3.76e7 / 6.2, 16 3.78e7 / 12.0 -- but with 4 a caller whose arrays live on the other
        // socket fell to 3.27e7/s (bench.py's process, one run), so 8: ~50 CPUs for an 8-GPU node instead of ~100
        const int dflt = std::min(8, std::max(1, usable_cpus() / nd));
        p->nthreads = env_int("CIRCL_HIP_HOST_THREADS", dflt, 0, 256);
        const std::vector<int> cpus = dev_info(dev).cpus;
        for (int t = 0; t < p->nthreads; t++) std::thread(worker_main, p, cpus).detach();
        g_pools[dev] = p;
    }
    return g_pools[dev];
}

// Streaming copy: non-temporal stores keep the destination out of the caches (no read-for-ownership traffic, and the
// DMA engine / the caller reads it from memory anyway).  dst / src arbitrary alignment.
__attribute__((target("avx2"))) void copy_stream_avx2(uint8_t *dst, const uint8_t *src, size_t n) {
    const size_t head = std::min(n, (size_t)((32 - (reinterpret_cast<uintptr_t>(dst) & 31)) & 31));
    if (head) { memcpy(dst, src, head); dst += head; src += head; n -= head; }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 96), d);
    }
    _mm_sfence();
    if (i < n) memcpy(dst + i, src + i, n - i);
}
void copy_bytes(void *dst, const void *src, size_t n) {
    static const bool nt = __builtin_cpu_supports("avx2");  // non-temporal copies for the staging areas
    if (nt && n >= 4096) copy_stream_avx2(static_cast<uint8_t *>(dst), static_cast<const uint8_t *>(src), n);
    else memcpy(dst, src, n);
}

}  // namespace

void pool_run(int dev, size_t n, const std::function<void(size_t)> &fn) {
    if (n == 0) return;
    Pool *p = pool_of(dev);
    if (n == 1 || p->nthreads == 0) {
        for (size_t i = 0; i < n; i++) fn(i);
        return;
    }
    auto b = std::make_shared<Batch>();
    b->n = n;
    b->fn = &fn;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->q.push_back(b);
    }
    p->cv.notify_all();
    for (;;) {  // the caller is a mover too
        const size_t i = b->next.fetch_add(1);
        if (i >= n) break;
        fn(i);
        b->done.fetch_add(1);
    }
    {
        std::unique_lock<std::mutex> lk(b->mu);
        b->cv.wait(lk, [&] { return b->done.load() >= n; });
    }
    std::lock_guard<std::mutex> lk(p->mu);
    for (auto it = p->q.begin(); it != p->q.end(); ++it)
        if (*it == b) { p->q.erase(it); break; }
}

void parallel_copy(int dev, const std::vector<CopyJob> &jobs) {
    constexpr size_t PIECE = size_t(1) << 20;
    size_t total = 0;
    for (auto &j : jobs) total += j.bytes;
    if (total == 0) return;
    auto one = [](const CopyJob &j, size_t lo, size_t len) {
        if (j.src) copy_bytes(static_cast<uint8_t *>(j.dst) + lo, static_cast<const uint8_t *>(j.src) + lo, len);
        else memset(static_cast<uint8_t *>(j.dst) + lo, 0, len);
    };
    if (total <= (size_t(1) << 18)) {  // not worth waking anybody
        for (auto &j : jobs)
            if (j.bytes) one(j, 0, j.bytes);
        return;
    }
    struct Piece { const CopyJob *j; size_t lo, len; };
    std::vector<Piece> pieces;
    pieces.reserve(total / PIECE + jobs.size());
    for (auto &j : jobs)
        for (size_t lo = 0; lo < j.bytes; lo += PIECE) pieces.push_back({&j, lo, std::min(PIECE, j.bytes - lo)});
    pool_run(dev, pieces.size(), [&](size_t i) { one(*pieces[i].j, pieces[i].lo, pieces[i].len); });
}

// ---- staging slots ----------------------------------------------------------------------------
namespace {
struct SlotPool {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Slot *> free_slots;
    int created = 0;
    // One stream per DIRECTION per device, shared by all slots and callers: a stream that alternates H2D and D2H copies
    // gets 60-68 GB/s for both directions together on this platform, a dedicated H2D stream next to a dedicated D2H
    // stream 81-95 GB/s (tools/pcie_probe.hip) -- the SDMA engines are bound per queue.
    // The kernels of all chunks of a call go to ONE compute stream (taken round-robin from two per device, so that two
    // concurrent callers do not share one): a chunk's kernels fill the GPU anyway, and the runtime multiplexes streams
    // onto a handful of hardware queues -- with a stream per slot, compute streams ended up behind the copy streams'
    // barrier packets and the PCIe rate depended erratically on the pipeline depth (tools/host_path.py sweeps).
    std::once_flag copy_once;
    hipStream_t h2d = nullptr, d2h = nullptr, compute[2] = {nullptr, nullptr};
    std::atomic<unsigned> next_compute{0};
    std::once_flag aux_once;
    hipStream_t aux[2] = {nullptr, nullptr};
};
std::mutex g_slot_pools_mu;
std::vector<SlotPool *> g_slot_pools;
SlotPool *slot_pool_of(int dev) {
    std::lock_guard<std::mutex> lk(g_slot_pools_mu);
    if (g_slot_pools.empty()) g_slot_pools.assign(std::max(ndev(), 1), nullptr);
    if (!g_slot_pools[dev]) g_slot_pools[dev] = new SlotPool;
    return g_slot_pools[dev];
}
}  // namespace
int pipeline_streams(int dev, hipStream_t *h2d, hipStream_t *d2h, hipStream_t *compute) {
    SlotPool *p = slot_pool_of(dev);
    std::call_once(p->copy_once, [&] {
        if (hipStreamCreateWithFlags(&p->h2d, hipStreamNonBlocking) != hipSuccess) p->h2d = nullptr;
        if (hipStreamCreateWithFlags(&p->d2h, hipStreamNonBlocking) != hipSuccess) p->d2h = nullptr;
        for (auto &c : p->compute)
            if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) c = nullptr;
    });
    if (!p->h2d || !p->d2h || !p->compute[0] || !p->compute[1]) { g_err = "stream creation failed"; (void)hipGetLastError(); return CIRCL_HIP_EHIP; }
    *h2d = p->h2d;
    *d2h = p->d2h;
    *compute = p->compute[p->next_compute.fetch_add(1) & 1];
    return CIRCL_HIP_OK;
}
// two more non-blocking streams per device for device-resident calls that fork internally (batch signing runs its two halves
// side by side: their latency-bound late rounds fill each other's gaps)
int aux_streams(int dev, hipStream_t (&s)[2]) {
    SlotPool *p = slot_pool_of(dev);
    std::call_once(p->aux_once, [&] {
        for (auto &a : p->aux)
            if (hipStreamCreateWithFlags(&a, hipStreamNonBlocking) != hipSuccess) a = nullptr;
    });
    if (!p->aux[0] || !p->aux[1]) { g_err = "stream creation failed"; (void)hipGetLastError(); return CIRCL_HIP_EHIP; }
    s[0] = p->aux[0];
    s[1] = p->aux[1];
    return CIRCL_HIP_OK;
}
namespace {
int max_slots() {
    static const int v = env_int("CIRCL_HIP_HOST_SLOTS", 12, 1, 64);
    return v;
}
}  // namespace

namespace {
// what the staging pools of all devices hold right now (circl_hip_host_pool_stats: sizing a node's host memory)
std::atomic<long long> g_pool_pinned{0}, g_pool_device{0};
std::atomic<int> g_pool_slots{0};
}  // namespace
int Slot::ensure(size_t d_bytes, size_t hin_bytes, size_t hout_bytes) {
    if (d_bytes > d_cap) {
        if (d) HIP_TRY(hipFree(d));  // (a slot is idle whenever it is resized: its last chunk was retired)
        g_pool_device -= (long long)d_cap;
        d = nullptr; d_cap = 0;
        const size_t want = up256(d_bytes + d_bytes / 8);  // a little head-room: ragged chunks differ in size
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), want));
        d_cap = want;
        g_pool_device += (long long)want;
    }
    if (hin_bytes > hin_cap) {
        if (hin) HIP_TRY(pinned_free(hin));
        g_pool_pinned -= (long long)hin_cap;
        hin = nullptr; hin_cap = 0;
        const size_t want = up256(hin_bytes + hin_bytes / 8);
        HIP_TRY(pinned_alloc(reinterpret_cast<void **>(&hin), want));
        hin_cap = want;
        g_pool_pinned += (long long)want;
        hin_dev = pinned_device_ptr(hin);
    }
    if (hout_bytes > hout_cap) {
        if (hout) HIP_TRY(pinned_free(hout));
        g_pool_pinned -= (long long)hout_cap;
        hout = nullptr; hout_cap = 0;
        const size_t want = up256(hout_bytes + hout_bytes / 8);
        HIP_TRY(pinned_alloc(reinterpret_cast<void **>(&hout), want));
        hout_cap = want;
        g_pool_pinned += (long long)want;
        hout_dev = pinned_device_ptr(hout);
    }
    return CIRCL_HIP_OK;
}

Slot *slot_acquire(int dev, bool block) {
    SlotPool *p = slot_pool_of(dev);
    {
        std::unique_lock<std::mutex> lk(p->mu);
        for (;;) {
            if (!p->free_slots.empty()) {
                Slot *s = p->free_slots.back();
                p->free_slots.pop_back();
                return s;
            }
            if (p->created < max_slots()) { p->created++; break; }
            if (!block) return nullptr;
            p->cv.wait(lk);
        }
    }
    Slot *s = new Slot;
    s->dev = dev;
    g_pool_slots++;
    if (hipEventCreateWithFlags(&s->done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_k, hipEventDisableTiming) != hipSuccess) {
        g_err = "slot_acquire: event creation failed";
        (void)hipGetLastError();
        for (hipEvent_t e : {s->done, s->ev_in, s->ev_k})
            if (e) (void)hipEventDestroy(e);
        delete s;
        g_pool_slots--;
        std::lock_guard<std::mutex> lk(p->mu);
        p->created--;
        p->cv.notify_one();
        return nullptr;
    }
    return s;
}
void slot_release(Slot *s) {
    if (!s) return;
    SlotPool *p = slot_pool_of(s->dev);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->free_slots.push_back(s);
    }
    p->cv.notify_one();
}

bool is_pinned_host(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

size_t host_chunk_items(size_t dflt) {
    static const int lg = env_int("CIRCL_HIP_HOST_CHUNK", 0, 8, 24);  // log2 of the chunk size, 0 = per-operation default
    return lg ? size_t(1) << lg : dflt;
}

// ---- the pipeline -------------------------------------------------------------------------------
namespace {
struct Seg { size_t dofs = 0, hofs = 0, bytes = 0; bool staged = false; };  // device offset, staging offset, payload bytes
struct InFlight {
    Slot *slot = nullptr;
    size_t lo = 0, cnt = 0;
    std::vector<Seg> out;      // per HOut
    std::vector<Seg> secret_in;  // staged secret inputs to wipe
};
}  // namespace

int run_pipeline(int dev, size_t n, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs,
                 const std::function<size_t(size_t)> &ws_bytes, const PipeOpts &opts, const std::function<int(Chunk &)> &launch) {
    if (n == 0) return CIRCL_HIP_OK;
    if (dev < 0 || dev >= ndev()) return CIRCL_HIP_ENODEV;
    for (auto &in : ins)
        if (!in.p && in.row) { g_err = "a required input pointer is NULL"; return CIRCL_HIP_EPARAM; }
    for (auto &b : blobs)
        if (b.blob && !b.off) { g_err = "a blob without offsets"; return CIRCL_HIP_EPARAM; }
    HIP_TRY(hipSetDevice(physical_device(dev)));
    const size_t chunk = std::max<size_t>(1, std::min(n, opts.chunk_items));
    const size_t depth = (size_t)std::max(1, opts.depth);  // chunks in flight per call
    std::vector<char> in_pinned(ins.size()), out_pinned(outs.size()), blob_pinned(blobs.size());
    for (size_t k = 0; k < ins.size(); k++) in_pinned[k] = is_pinned_host(ins[k].p);
    for (size_t k = 0; k < outs.size(); k++) out_pinned[k] = outs[k].p && is_pinned_host(outs[k].p);
    for (size_t k = 0; k < blobs.size(); k++) blob_pinned[k] = blobs[k].blob && is_pinned_host(blobs[k].blob);

    hipStream_t h2d = nullptr, d2h = nullptr, st = nullptr;
    if (int rc = pipeline_streams(dev, &h2d, &d2h, &st)) return rc;
    bool tiny = false;
    {
        // A call that is ONE small chunk gains nothing from separate copy streams and pays for them: two cross-stream event hops
        // and two SDMA start-ups are ~20 us of a one-item call's 140 (measured through the Python binding: n = 1 139 -> 120 us, 64 items
        // 166 -> 147, 1 024 items 316 -> 299).  Its copies go on the compute stream (the events below are then recorded and awaited on
        // one stream, which costs nothing).  The largest such call moves 4 MB.
        constexpr size_t inline_bytes = size_t(4) << 20;
        size_t moved = 0;
        for (auto &in : ins) moved += in.row * (in.per_call ? 1 : n);
        for (auto &o : outs) moved += o.row * n;
        for (auto &b : blobs)
            if (b.blob) moved += (size_t)(b.off[n] - b.off[0]) + 8 * (n + 1);
        if (n <= chunk && moved <= inline_bytes) h2d = d2h = st;
        tiny = n <= chunk && moved <= zero_copy_bytes();
    }
    // ... and such a call moves all its staged inputs with ONE copy and all its outputs with ONE copy: the page-locked staging areas
    // then mirror the device staging's layout (same offsets, padding included), and the cross-stream events are not needed.  A one-item
    // call spends most of its time in HIP API calls (~3 us each): 2 + 3 copies, 4 event calls and 3-4 memsets become 2 copies and
    // 3 memsets (profiles/r04_host_small.txt).
    const bool one_stream = h2d == st && d2h == st;
    bool merge_in = one_stream, merge_out = one_stream;
    for (size_t k = 0; k < ins.size(); k++) merge_in = merge_in && !in_pinned[k];
    for (size_t k = 0; k < blobs.size(); k++) merge_in = merge_in && (!blobs[k].blob || !blob_pinned[k]);
    for (size_t k = 0; k < outs.size(); k++) merge_out = merge_out && (!outs[k].p || !out_pinned[k]);
    // ... and a TINY such call (<= zero_copy_bytes(), 64 KB) enqueues no copy at all: its kernels read the page-locked input staging and
    // write the page-locked output staging over PCIe themselves (the areas are device-mapped; their contents are visible to the host
    // once the stream has drained, like any kernel output).  What is left of a one-item call is the launch and the wait
    // (profiles/r05_zerocopy.txt).
    const bool zero_copy = tiny && merge_in && merge_out;
    std::deque<InFlight> inflight;
    // error paths must not recycle a slot (or return to the caller) with copies or kernels still in flight
    struct Drain {
        std::deque<InFlight> &q;
        hipStream_t h2d, d2h, st;
        const std::vector<HOut> &outs;
        bool wipe_device, merge_out;
        ~Drain() {
            if (q.empty()) return;
            (void)hipStreamSynchronize(h2d);
            (void)hipStreamSynchronize(st);
            (void)hipStreamSynchronize(d2h);
            // a failed call leaves nothing secret behind either: the staging copies of secret inputs / outputs of every chunk
            // still in flight, and (for calls that wipe the device staging) the whole device slot
            for (auto &f : q) {
                for (auto &s : f.secret_in) memset(f.slot->hin + s.hofs, 0, s.bytes);
                for (size_t k = 0; k < outs.size() && k < f.out.size(); k++)
                    if (outs[k].secret && (f.out[k].staged || merge_out) && f.slot->hout && f.out[k].hofs + f.out[k].bytes <= f.slot->hout_cap)
                        memset(f.slot->hout + f.out[k].hofs, 0, f.out[k].bytes);
                if (wipe_device && f.slot->d) (void)hipMemset(f.slot->d, 0, f.slot->d_cap);
                slot_release(f.slot);
            }
            (void)hipGetLastError();
        }
    } drain{inflight, h2d, d2h, st, outs, opts.wipe_device, merge_out};

    // stage-out of a finished chunk = copy jobs (staging -> caller memory), then wipe jobs (secret staging areas)
    auto out_jobs = [&](InFlight &f, std::vector<CopyJob> &jobs) {
        for (size_t k = 0; k < outs.size(); k++)
            if (outs[k].p && f.out[k].staged) jobs.push_back({outs[k].p + f.lo * outs[k].row, f.slot->hout + f.out[k].hofs, f.out[k].bytes});
    };
    auto wipe_jobs = [&](InFlight &f, std::vector<CopyJob> &jobs) {
        // (a merged / zero-copy call brings EVERY output segment to the page-locked area, wanted by the caller or not)
        for (size_t k = 0; k < outs.size(); k++)
            if (outs[k].secret && (f.out[k].staged || merge_out) && f.out[k].bytes) jobs.push_back({f.slot->hout + f.out[k].hofs, nullptr, f.out[k].bytes});
        for (auto &s : f.secret_in) jobs.push_back({f.slot->hin + s.hofs, nullptr, s.bytes});
    };
    auto retire = [&](InFlight &f) -> int {
        HIP_TRY(hipEventSynchronize(f.slot->done));
        std::vector<CopyJob> jobs;
        out_jobs(f, jobs);
        parallel_copy(dev, jobs);
        jobs.clear();
        wipe_jobs(f, jobs);
        parallel_copy(dev, jobs);
        return CIRCL_HIP_OK;
    };
    // Before chunk c is enqueued, the inputs of chunk c - AHEAD must have arrived: at most AHEAD chunks' worth of copies is
    // ever queued on the H2D stream, however many slots the call holds.  Measured (tools/host_path.py, 2^20 ML-KEM-768
    // encapsulations): without the bound the rate falls with every extra chunk in flight (4.0e7/s at depth 3, 2.6e7 at 4,
    // 2.0e7 at 6, page-locked caller buffers); with AHEAD = 1 it is 4.0e7/s (48.6 + 44.8 GB/s, the bidirectional PCIe
    // ceiling of this box) at any depth.
    constexpr int ahead = 1;

    for (size_t lo = 0; lo < n; lo += chunk) {
        const size_t cnt = std::min(chunk, n - lo);
        // ---- layout of this chunk ----
        size_t dofs = 0, hin_ofs = 0, hout_ofs = 0;
        auto take_d = [&](size_t bytes) { const size_t o = dofs; dofs += up256(bytes + 16); return o; };  // + slack: kernels may read whole dwords
        auto take_hin = [&](size_t bytes) { const size_t o = hin_ofs; hin_ofs += (bytes + 63) & ~size_t(63); return o; };
        auto take_hout = [&](size_t bytes) { const size_t o = hout_ofs; hout_ofs += (bytes + 63) & ~size_t(63); return o; };
        std::vector<Seg> sin(ins.size()), sblob(blobs.size()), soff(blobs.size());
        InFlight f;
        f.lo = lo; f.cnt = cnt;
        f.out.resize(outs.size());
        for (size_t k = 0; k < ins.size(); k++) {
            Seg &s = sin[k];
            s.bytes = ins[k].row * (ins[k].per_call ? 1 : cnt);
            s.dofs = take_d(s.bytes);
            s.staged = !in_pinned[k];
            if (merge_in) { s.hofs = s.dofs; hin_ofs = dofs; }
            else if (s.staged) s.hofs = take_hin(s.bytes);
        }
        for (size_t k = 0; k < blobs.size(); k++) {
            if (!blobs[k].blob) continue;
            sblob[k].bytes = (size_t)(blobs[k].off[lo + cnt] - blobs[k].off[lo]);
            sblob[k].dofs = take_d(sblob[k].bytes);
            sblob[k].staged = !blob_pinned[k];
            if (merge_in) { sblob[k].hofs = sblob[k].dofs; hin_ofs = dofs; }
            else if (sblob[k].staged) sblob[k].hofs = take_hin(sblob[k].bytes);
            soff[k].bytes = (cnt + 1) * 8;
            soff[k].dofs = take_d(soff[k].bytes);
            soff[k].staged = true;
            if (merge_in) { soff[k].hofs = soff[k].dofs; hin_ofs = dofs; }
            else soff[k].hofs = take_hin(soff[k].bytes);
        }
        const size_t in_end = dofs, out_begin = dofs;
        for (size_t k = 0; k < outs.size(); k++) {
            Seg &s = f.out[k];
            s.bytes = outs[k].row * cnt;
            s.dofs = take_d(s.bytes);
            s.staged = outs[k].p && !out_pinned[k];
            if (merge_out) { s.hofs = s.dofs - out_begin; hout_ofs = dofs - out_begin; }
            else if (s.staged) s.hofs = take_hout(s.bytes);
        }
        const size_t out_end = dofs;
        const size_t wsb = ws_bytes(cnt);
        const size_t ws_ofs = dofs;
        dofs += up256(wsb);

        // A call never WAITS for a slot while it holds one (several concurrent callers doing that on a small pool would
        // deadlock): if the pool is exhausted it first drains its own oldest chunk; only a call that holds nothing blocks.
        Slot *slot = nullptr;
        for (;;) {
            g_err.clear();
            slot = slot_acquire(dev, inflight.empty());
            if (slot) break;
            if (inflight.empty()) return CIRCL_HIP_EHIP;  // creation failed (g_err says why)
            if (!g_err.empty()) return CIRCL_HIP_EHIP;
            const int rc = retire(inflight.front());
            if (rc) return rc;
            slot_release(inflight.front().slot);
            inflight.pop_front();
        }
        f.slot = slot;
        inflight.push_back(f);  // from here on the Drain guard owns the slot
        InFlight &cur = inflight.back();
        int rc = slot->ensure(dofs, hin_ofs, hout_ofs);
        if (rc) return rc;
        const bool zc = zero_copy && (!in_end || slot->hin_dev) && (out_end == out_begin || slot->hout_dev);

        // ---- stage in -- together with the stage-out of the oldest chunk when the call holds `depth` slots: one batch for
        // the byte movers instead of two half-empty ones ----
        std::vector<CopyJob> jobs;
        for (size_t k = 0; k < ins.size(); k++)
            if (sin[k].staged && sin[k].bytes) {
                jobs.push_back({slot->hin + sin[k].hofs, ins[k].p + (ins[k].per_call ? 0 : lo * ins[k].row), sin[k].bytes});
                if (ins[k].secret) cur.secret_in.push_back(sin[k]);
            }
        for (size_t k = 0; k < blobs.size(); k++) {
            if (!blobs[k].blob) continue;
            if (sblob[k].staged && sblob[k].bytes) jobs.push_back({slot->hin + sblob[k].hofs, blobs[k].blob + blobs[k].off[lo], sblob[k].bytes});
            jobs.push_back({slot->hin + soff[k].hofs, blobs[k].off + lo, soff[k].bytes});
        }
        const bool retire_front = inflight.size() > depth;  // (the new chunk is already in the deque)
        if (retire_front) {
            HIP_TRY(hipEventSynchronize(inflight.front().slot->done));
            out_jobs(inflight.front(), jobs);
        }
        parallel_copy(dev, jobs);
        if (retire_front) {
            jobs.clear();
            wipe_jobs(inflight.front(), jobs);
            parallel_copy(dev, jobs);
            slot_release(inflight.front().slot);
            inflight.pop_front();
        }
        if (ahead > 0 && inflight.size() > (size_t)ahead) HIP_TRY(hipEventSynchronize(inflight[inflight.size() - 1 - ahead].slot->ev_in));

        // ---- enqueue: H2D on the device's H2D stream, kernels on the call's compute stream, D2H on the device's D2H stream ----
        Chunk c;
        c.cnt = cnt; c.st = st;
        c.ws = slot->d + ws_ofs; c.ws_bytes = up256(wsb);
        for (size_t k = 0; k < ins.size(); k++) {
            uint8_t *dp = zc ? slot->hin_dev + sin[k].hofs : slot->d + sin[k].dofs;
            c.in.push_back(dp);
            if (!sin[k].bytes || merge_in) continue;
            const void *src = sin[k].staged ? (const void *)(slot->hin + sin[k].hofs) : (const void *)(ins[k].p + (ins[k].per_call ? 0 : lo * ins[k].row));
            HIP_TRY(hipMemcpyAsync(dp, src, sin[k].bytes, hipMemcpyHostToDevice, h2d));
        }
        if (merge_in && in_end && !zc) HIP_TRY(hipMemcpyAsync(slot->d, slot->hin, in_end, hipMemcpyHostToDevice, h2d));  // every input, blob and offset array at once
        for (size_t k = 0; k < blobs.size(); k++) {
            if (!blobs[k].blob) { c.blob.push_back(nullptr); c.off.push_back(nullptr); continue; }
            uint8_t *dp = zc ? slot->hin_dev + sblob[k].hofs : slot->d + sblob[k].dofs;
            if (sblob[k].bytes && !merge_in) {
                const void *src = sblob[k].staged ? (const void *)(slot->hin + sblob[k].hofs) : (const void *)(blobs[k].blob + blobs[k].off[lo]);
                HIP_TRY(hipMemcpyAsync(dp, src, sblob[k].bytes, hipMemcpyHostToDevice, h2d));
            }
            if (!merge_in) HIP_TRY(hipMemcpyAsync(slot->d + soff[k].dofs, slot->hin + soff[k].hofs, soff[k].bytes, hipMemcpyHostToDevice, h2d));
            c.blob.push_back(dp - blobs[k].off[lo]);  // the kernels index it with the caller's absolute offsets
            c.off.push_back(reinterpret_cast<const uint64_t *>(zc ? slot->hin_dev + soff[k].hofs : slot->d + soff[k].dofs));
        }
        for (size_t k = 0; k < outs.size(); k++) c.out.push_back(zc ? slot->hout_dev + cur.out[k].hofs : slot->d + cur.out[k].dofs);
        HIP_TRY(hipEventRecord(slot->ev_in, h2d));  // (the look-ahead bound of the next chunk waits on it)
        if (h2d != st) HIP_TRY(hipStreamWaitEvent(st, slot->ev_in, 0));
        rc = launch(c);
        if (rc) return rc;
        if (d2h != st) {
            HIP_TRY(hipEventRecord(slot->ev_k, st));
            HIP_TRY(hipStreamWaitEvent(d2h, slot->ev_k, 0));
        }
        if (merge_out && out_end > out_begin && !zc) HIP_TRY(hipMemcpyAsync(slot->hout, slot->d + out_begin, out_end - out_begin, hipMemcpyDeviceToHost, d2h));
        for (size_t k = 0; k < outs.size(); k++) {
            if (!outs[k].p || !cur.out[k].bytes || merge_out) continue;
            void *dst = cur.out[k].staged ? (void *)(slot->hout + cur.out[k].hofs) : (void *)(outs[k].p + lo * outs[k].row);
            HIP_TRY(hipMemcpyAsync(dst, slot->d + cur.out[k].dofs, cur.out[k].bytes, hipMemcpyDeviceToHost, d2h));
        }
        if (opts.wipe_device) {  // keys, seeds and intermediates do not outlive the chunk
            if (!opts.ws_secret_bytes) {
                HIP_TRY(hipMemsetAsync(slot->d, 0, dofs, d2h));
            } else {
                for (size_t k = 0; k < ins.size() && !zc; k++)  // (a zero-copy call left nothing in the device staging but its workspace)
                    if (ins[k].secret && sin[k].bytes) HIP_TRY(hipMemsetAsync(slot->d + sin[k].dofs, 0, sin[k].bytes, d2h));
                for (size_t k = 0; k < outs.size() && !zc; k++)
                    if (outs[k].secret && cur.out[k].bytes) HIP_TRY(hipMemsetAsync(slot->d + cur.out[k].dofs, 0, cur.out[k].bytes, d2h));
                const size_t sec = std::min(up256(wsb), opts.ws_secret_bytes(cnt));
                if (sec) HIP_TRY(hipMemsetAsync(slot->d + ws_ofs, 0, sec, d2h));
            }
        }
        HIP_TRY(hipEventRecord(slot->done, d2h));
    }
    while (!inflight.empty()) {
        const int rc = retire(inflight.front());
        if (rc) return rc;
        slot_release(inflight.front().slot);
        inflight.pop_front();
    }
    return CIRCL_HIP_OK;
}

int upload_secret(void *d_dst, const void *h_src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return CIRCL_HIP_OK;
    if (is_pinned_host(h_src)) {  // the caller's own page-locked memory: DMA-ed as is, the caller's to clear
        HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        return CIRCL_HIP_OK;
    }
    void *pin = nullptr;
    HIP_TRY(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    memcpy(pin, h_src, bytes);
    const hipError_t e1 = hipMemcpyAsync(d_dst, pin, bytes, hipMemcpyHostToDevice, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    volatile uint8_t *z = static_cast<volatile uint8_t *>(pin);
    for (size_t i = 0; i < bytes; i++) z[i] = 0;
    (void)hipHostFree(pin);
    HIP_TRY(e1);
    HIP_TRY(e2);
    return CIRCL_HIP_OK;
}

// ---- key tables on several devices (keytable.h) ---------------------------------------------------
const circl_hip_keytable *keytable_here(const circl_hip_keytable *t) {
    if (!t || t->magic != kKeytableMagic) return nullptr;
    if (t->device >= 0) return t;
    int phys = 0;
    if (hipGetDevice(&phys) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    for (int d = 0; d < t->nreplica; d++)
        if (physical_device(d) == phys) return t->replica[d];
    return nullptr;
}
int keytable_replicate(int device, const std::function<int(int dev, circl_hip_keytable **one)> &make, circl_hip_keytable **out) {
    *out = nullptr;
    const int nd = ndev();
    if (nd <= 0) return CIRCL_HIP_ENODEV;
    if (device >= 0) return device < nd ? make(device, out) : CIRCL_HIP_ENODEV;
    if (device != CIRCL_HIP_ALL_DEVICES) return CIRCL_HIP_EPARAM;
    circl_hip_keytable *top = new (std::nothrow) circl_hip_keytable();
    if (!top) return CIRCL_HIP_ENOMEM;
    top->magic = kKeytableMagic;
    top->device = CIRCL_HIP_ALL_DEVICES;
    top->replica = new (std::nothrow) circl_hip_keytable *[nd]();
    if (!top->replica) { delete top; return CIRCL_HIP_ENOMEM; }
    for (int d = 0; d < nd; d++) {
        const int rc = make(d, &top->replica[d]);
        if (rc != CIRCL_HIP_OK) { circl_hip_keytable_free(top); return rc; }
        top->nreplica = d + 1;
    }
    const circl_hip_keytable *r0 = top->replica[0];
    top->family = r0->family; top->param = r0->param; top->private_keys = r0->private_keys; top->nkeys = r0->nkeys; top->row = r0->row;
    top->scheme = r0->scheme;
    *out = top;
    return CIRCL_HIP_OK;
}

int next_replica(int nreplica) {
    static std::atomic<unsigned> rr{0};
    return (int)(rr.fetch_add(1, std::memory_order_relaxed) % (unsigned)std::max(nreplica, 1));
}

// ---- shard ----------------------------------------------------------------------------------------
int shard(size_t n, int device, const std::function<int(int dev, size_t lo, size_t cnt)> &fn, size_t one_device_max) {
    const int nd = ndev();
    if (nd <= 0) return CIRCL_HIP_ENODEV;
    if (device >= 0) return device < nd ? fn(device, size_t(0), n) : CIRCL_HIP_ENODEV;
    if (device != CIRCL_HIP_ALL_DEVICES) return CIRCL_HIP_EPARAM;
    if (nd == 1) return fn(0, size_t(0), n);
    // a SMALL call goes to ONE device, taken round-robin (keytable.h table_shard does the same): a thread and a launch per device
    // for a handful of items cost more than they return, and the contiguous split sent every one-item call to the last device
    if (n <= one_device_max) return fn(next_replica(nd), size_t(0), n);
    std::vector<int> rcs(nd, 0);
    std::vector<std::string> errs(nd);
    std::vector<std::thread> th;
    for (int d = 0; d < nd; d++) {
        const size_t lo = n * d / nd, hi = n * (d + 1) / nd;
        th.emplace_back([&, d, lo, hi] {
            pin_to(dev_info(d).cpus);  // this thread drives device d: stay on its NUMA node
            rcs[d] = hi > lo ? fn(d, lo, hi - lo) : CIRCL_HIP_OK;
            errs[d] = g_err;
        });
    }
    for (auto &t : th) t.join();
    for (int d = 0; d < nd; d++)
        if (rcs[d]) {
            g_err = errs[d];
            return rcs[d];
        }
    return CIRCL_HIP_OK;
}

}  // namespace host
}  // namespace circl

// ---- library management, profiling and pinned-memory helpers of the C ABI ----------------------------
using namespace circl::host;

extern "C" {

int circl_hip_init(void) {
    const int n = ndev();
    return n > 0 ? n : CIRCL_HIP_ENODEV;
}
int circl_hip_device_count(void) { return std::max(ndev(), 0); }
const char *circl_hip_last_error(void) { return g_err.c_str(); }
const char *circl_hip_version(void) { return "circl-hip 0.3 (gfx950)"; }
int circl_hip_physical_device(int device) { return device >= 0 && device < ndev() ? physical_device(device) : CIRCL_HIP_ENODEV; }

int circl_hip_device_info(int device, int *cus, int *numa_node) {
    if (device < 0 || device >= ndev()) return CIRCL_HIP_ENODEV;
    const DeviceInfo &di = dev_info(device);
    if (cus) *cus = di.cus;
    if (numa_node) *numa_node = di.numa;
    return CIRCL_HIP_OK;
}

int circl_hip_profile_enable(int on) {
    g_prof_on.store(on != 0);
    return CIRCL_HIP_OK;
}
int circl_hip_profile_read(int kernel, double *total_ms, uint64_t *launches) {
    if (kernel < 0 || kernel >= CIRCL_HIP_KERNEL_COUNT) return CIRCL_HIP_EPARAM;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof_pending) {
        float ms = 0;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            g_prof_ms[r.kernel] += ms;
            g_prof_n[r.kernel] += 1;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof_pending.clear();
    if (total_ms) *total_ms = g_prof_ms[kernel];
    if (launches) *launches = g_prof_n[kernel];
    g_prof_ms[kernel] = 0;
    g_prof_n[kernel] = 0;
    return CIRCL_HIP_OK;
}

// keytable.h: a table of private keys is wiped before its memory goes back (the expanded rows are public, the key rows are not)
static void keytable_release(circl_hip_keytable *t) {
    t->magic = 0;
    for (int d = 0; d < t->nreplica; d++)
        if (t->replica[d]) keytable_release(t->replica[d]);
    delete[] t->replica;
    if (t->inner) keytable_release(t->inner);
    if (t->coalescer) { circl::host::coalescer_free(t->coalescer); t->coalescer = nullptr; }  // (an asynchronous queue finishes its tickets first)
    if (ndev() > 0 && t->device >= 0 && t->device < ndev() && hipSetDevice(circl::host::physical_device(t->device)) == hipSuccess) {
        if (t->d_keys) {
            if (t->private_keys) (void)hipMemset(t->d_keys, 0, t->keys_bytes);
            (void)hipFree(t->d_keys);
        }
        if (t->d_table) {
            if (t->private_keys && t->family == 2) (void)hipMemset(t->d_table, 0, t->table_bytes);  // ML-DSA: the transformed secrets
            (void)hipFree(t->d_table);
        }
        if (t->d_x) {
            if (t->private_keys) (void)hipMemset(t->d_x, 0, t->x_bytes);
            (void)hipFree(t->d_x);
        }
    }
    (void)hipGetLastError();
    delete t;
}
// Never frees under a caller: waits (up to two seconds) for the calls inside the table to return; if they do not, the table is marked
// dead -- later calls get CIRCL_HIP_EPARAM -- and its memory is LEAKED rather than pulled from under them.
void circl_hip_keytable_free(circl_hip_keytable *t) {
    if (!t || t->magic != kKeytableMagic) return;
    if (!circl::host::keytable_quiesce(t)) {
        t->magic = 0;
        g_err = "circl_hip_keytable_free: calls still inside the table after 2 s; the table was marked dead and leaked";
        return;
    }
    keytable_release(t);
}
int circl_hip_keytable_close(circl_hip_keytable *t) {
    if (!t || t->magic != kKeytableMagic) return CIRCL_HIP_EPARAM;
    if (int rc = circl_hip_keytable_async_stop(t)) return rc;  // CIRCL_HIP_EBUSY while a call is inside (coalescers and queues otherwise go first)
    if (t->users.load() != 0) { g_err = "the table has calls in flight"; return CIRCL_HIP_EBUSY; }
    circl_hip_keytable_free(t);
    return CIRCL_HIP_OK;
}
int circl_hip_keytable_device(const circl_hip_keytable *t) { return (t && t->magic == kKeytableMagic) ? t->device : CIRCL_HIP_ENODEV; }
size_t circl_hip_keytable_nkeys(const circl_hip_keytable *t) { return (t && t->magic == kKeytableMagic) ? t->nkeys : 0; }
const circl_hip_keytable *circl_hip_keytable_on_device(const circl_hip_keytable *t, int device) { return circl::host::keytable_on(t, device); }

int circl_hip_host_pool_stats(int *slots, uint64_t *pinned_bytes, uint64_t *device_bytes) {
    if (slots) *slots = g_pool_slots.load();
    if (pinned_bytes) *pinned_bytes = (uint64_t)std::max<long long>(0, g_pool_pinned.load());
    if (device_bytes) *device_bytes = (uint64_t)std::max<long long>(0, g_pool_device.load());
    return CIRCL_HIP_OK;
}

void *circl_hip_alloc_host(size_t bytes) {
    void *p = nullptr;
    if (ndev() <= 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void circl_hip_free_host(void *p) {
    if (p) (void)hipHostFree(p);
}

}  // extern "C"
