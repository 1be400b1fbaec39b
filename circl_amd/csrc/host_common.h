// host_common.h -- host-side plumbing shared by the translation units of libcirclhip.so.
//
//   per-device state     CU count, NUMA node, occupancy cache keyed on (device, kernel), all race-free
//   kernel profiling     HIP-event brackets on the launch stream (circl_hip_profile_*)
//   host-buffer pipeline a per-device pool of staging slots (page-locked host staging + device staging + events),
//                        filled and drained by a small per-device thread pool pinned to the GPU's NUMA node, so that a
//                        caller with ordinary pageable memory (a Go []byte) reaches the PCIe-bound rate; concurrent
//                        callers take different slots and overlap
//   shard()              contiguous split of a batch over the visible devices, one host thread each, no collective
//
// There is deliberately no CPU compute path anywhere in this library: the thread pool only moves bytes.
#pragma once
#include "../../include/circl_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace circl {
namespace host {

extern thread_local std::string g_err;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            char b_[256];                                                                          \
            snprintf(b_, sizeof b_, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            ::circl::host::g_err = b_;                                                             \
            return e_ == hipErrorOutOfMemory ? CIRCL_HIP_ENOMEM : CIRCL_HIP_EHIP;                   \
        }                                                                                          \
    } while (0)

inline size_t up256(size_t x) { return (x + 255) & ~size_t(255); }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- devices ----------------------------------------------------------------------------------
int ndev();                      // LOGICAL devices (0 if none), fixed at first call: the HIP devices of the process unless
                                 // CIRCL_HIP_LOGICAL_DEVICES says otherwise (host_runtime.hip); every `dev` below is logical
int physical_device(int dev);    // the HIP device a logical device runs on (dev % HIP device count)
struct DeviceInfo {
    int cus = 256;               // compute units
    int numa = -1;               // NUMA node of the PCIe function, -1 unknown
    std::vector<int> cpus;       // CPUs of that node this process may run on (empty: no pinning)
};
const DeviceInfo &dev_info(int dev);
void pin_to(const std::vector<int> &cpus);  // the calling thread may run on these CPUs only (nothing happens for an empty list)
int current_device();            // hipGetDevice, 0 on failure
inline int cu_count() { return dev_info(current_device()).cus; }  // of the CURRENT device (launch geometry)
int max_cu_count();              // over all visible devices (workspace sizing: valid whichever device runs the call)
int env_int(const char *name, int dflt, int lo, int hi);  // an integer tuning knob from the environment, clamped
int usable_cpus();               // affinity mask capped by the cgroup CPU quota

// ---- persistent-launch geometry for the scratch-based kernels -----------------------------------
#ifndef CIRCL_MAX_BLOCKS_PER_CU
#define CIRCL_MAX_BLOCKS_PER_CU 16  // 4 single-wave workgroups per SIMD
#endif
constexpr int kMaxBlocksPerCU = CIRCL_MAX_BLOCKS_PER_CU;
// upper bound on resident single-wave workgroups on any device: sizes the scratch part of a workspace
inline size_t max_resident_blocks() { return (size_t)max_cu_count() * kMaxBlocksPerCU; }
unsigned resident_blocks_cached(int dev, const void *key, const std::function<int()> &query);
// resident single-wave workgroups of `kern` on the current device (occupancy query cached per (device, kernel))
template <class Kern> unsigned resident_blocks(Kern kern, int lds_bytes) {
    const int dev = current_device();
    return resident_blocks_cached(dev, reinterpret_cast<const void *>(kern), [&]() -> int {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64, (size_t)lds_bytes) != hipSuccess || occ < 1) occ = 4;
        return occ;
    });
}

// ---- kernel-level profiling ---------------------------------------------------------------
bool prof_on();
void prof_push(int kernel, hipEvent_t a, hipEvent_t b);
// RAII bracket around one or more kernel launches on `st`
struct ProfScope {
    int kernel = -1;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    ProfScope(int k, hipStream_t s) : st(s) {
        if (!prof_on()) return;
        if (hipEventCreate(&a) != hipSuccess) return;
        if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return; }
        kernel = k;
        (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (kernel < 0) return;
        (void)hipEventRecord(b, st);
        prof_push(kernel, a, b);
    }
};

// ---- host thread pool (byte movers) ---------------------------------------------------------
// run(n, fn): fn(0) .. fn(n-1) spread over the device's worker threads and the calling thread; returns when all are done.
void pool_run(int dev, size_t n, const std::function<void(size_t)> &fn);
struct CopyJob { void *dst; const void *src; size_t bytes; };  // src == nullptr: zero-fill dst
void parallel_copy(int dev, const std::vector<CopyJob> &jobs);

// ---- staging slots ----------------------------------------------------------------------------
struct Slot {
    int dev = -1;
    hipEvent_t done = nullptr, ev_in = nullptr, ev_k = nullptr;  // outputs are home / inputs have arrived / kernels are done
    uint8_t *d = nullptr;     size_t d_cap = 0;      // device staging: inputs, outputs, workspace of one chunk
    uint8_t *hin = nullptr;   size_t hin_cap = 0;    // page-locked staging, host -> device
    uint8_t *hout = nullptr;  size_t hout_cap = 0;   // page-locked staging, device -> host
    uint8_t *hin_dev = nullptr, *hout_dev = nullptr; // the DEVICE addresses of the two page-locked areas (zero-copy calls: the kernels
                                                     // of a tiny call read / write them over PCIe themselves, no copy is enqueued)
    int ensure(size_t d_bytes, size_t hin_bytes, size_t hout_bytes);
};
// page-locked, device-mapped host memory (TSan builds: mmap + hipHostRegister, host_runtime.hip) and its address on the device
hipError_t pinned_alloc(void **p, size_t bytes);
hipError_t pinned_free(void *p);
uint8_t *pinned_device_ptr(void *p);  // nullptr if the runtime cannot map it
// A call (or a coalesced batch) that moves at most this many bytes does not enqueue copies at all: its kernels get the device
// addresses of the page-locked staging areas.  CIRCL_HIP_ZEROCOPY_KB (0 = never), default 1 MB (profiles/r05_zerocopy.txt).
size_t zero_copy_bytes();
// block = true: waits while every slot of the device is in use; false: returns nullptr at once in that case.
// nullptr with a non-empty g_err = creating a slot failed.
Slot *slot_acquire(int dev, bool block = true);
void slot_release(Slot *s);
bool is_pinned_host(const void *p);  // page-locked (hipHostMalloc / hipHostRegister) memory: DMA-able as is
// Host -> device copy of SECRET bytes (private keys of a key table): through a page-locked buffer of the library's own that is
// zeroed afterwards -- a hipMemcpyAsync straight from pageable memory bounces through the runtime's staging, which nobody wipes.
// Synchronises `st`.
int upload_secret(void *d_dst, const void *h_src, size_t bytes, hipStream_t st);

// ---- the pipeline -------------------------------------------------------------------------------
struct HIn {                 // fixed-size rows: item i is row i
    const uint8_t *p;
    size_t row;
    bool secret = false;     // wipe the staging copy afterwards
    bool per_call = false;   // ONE row for the whole call (a shared key): re-staged with every chunk
    bool optional = false;   // coalesce_run / coalesce_submit: p == nullptr means rows of ZEROS (an absent key_idx / rnd); without it a
                             // NULL pointer is CIRCL_HIP_EPARAM there, as it is in run_pipeline
};
struct HBlob {               // ragged rows: item i is blob[off[i] .. off[i+1]); blob == nullptr: absent (kernels get nullptr)
    const uint8_t *blob;
    const uint64_t *off;
};
struct HOut {
    uint8_t *p;              // nullptr: the device buffer exists but nothing is copied back
    size_t row;
    bool secret = false;
};
struct Chunk {               // what `launch` gets: device pointers of one chunk
    std::vector<uint8_t *> in;            // one per HIn
    std::vector<const uint8_t *> blob;    // one per HBlob, REBASED so that the caller's absolute offsets index it (nullptr if absent)
    std::vector<const uint64_t *> off;    // one per HBlob: the chunk's cnt + 1 offsets (nullptr if absent)
    std::vector<uint8_t *> out;           // one per HOut
    uint8_t *ws;
    size_t ws_bytes;
    size_t cnt;
    hipStream_t st;
};
struct PipeOpts {
    size_t chunk_items = size_t(1) << 15;
    int depth = 6;            // chunks in flight per call (staging slots held)
    bool wipe_device = false; // zero the device staging of every chunk once its results are out
    // With wipe_device: how much of a chunk's WORKSPACE is secret (from its start).  Unset = all of it.  Set (ML-KEM:
    // the per-item slots m', r', K', J; the matrix scratch behind them is public) = only the secret inputs, the secret
    // outputs and that prefix are zeroed, a few hundred KB instead of the 134 MB scratch.
    std::function<size_t(size_t)> ws_secret_bytes;
    // coalesced batches only: the launch is ONE call of an entry point whose one-launch route may raise the batch's completion flag
    // itself (TailOffer below).  Never set where the launch runs further kernels behind such a call (the hybrids call the ML-KEM table
    // entry points and then the X25519 ladder and the combiner: the flag would go up before they ran).
    bool tail_flag_ok = false;
};
// Runs items [0, n) on device `dev`: per chunk  stage-in (host threads) -> H2D -> launch -> D2H -> stage-out (host threads),
// with `depth` chunks in flight on separate streams.  ws_bytes(cnt) = workspace the launch needs for cnt items.
int run_pipeline(int dev, size_t n, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs,
                 const std::function<size_t(size_t)> &ws_bytes, const PipeOpts &opts, const std::function<int(Chunk &)> &launch);
size_t host_chunk_items(size_t dflt);  // CIRCL_HIP_HOST_CHUNK overrides the default chunk size (tuning aid)

// ---- cross-caller coalescing of small calls through ONE resident key table (host_coalesce.hip) -----------------------------
// The reference's consumers call kem.Scheme / sign.Scheme one operation at a time from many goroutines (kem/hybrid/hybrid.go:95-99,
// hpke/algs.go:283-285, kem/mlkem/mlkem768/kyber.go:347-386).  A Coalescer merges such concurrent small calls into one launch:
// callers reserve rows of an open batch (one compare-and-swap) and copy their inputs into its page-locked staging; the caller that opened the
// batch flushes it as soon as the device has room for another batch (so a batch collects exactly the calls that arrive while the previous ones
// run: no timer at low load, large batches at high load), or after max_wait_us if that is set; everybody copies their own rows out.
// Bytes are those of the un-coalesced call: the kernels are the same ones, an item does not know its neighbours.
struct Coalescer;
Coalescer *coalescer_new(int dev, size_t max_items, unsigned max_wait_us);
void coalescer_free(Coalescer *co);   // an asynchronous queue finishes what was submitted first (its dispatcher drains, then exits)
bool coalescer_idle(Coalescer *co);   // no call inside, no batch open or running (what a setter checks before it frees one)
size_t coalescer_call_max(const Coalescer *co);  // calls of more items than this do not join batches
void coalescer_stats(const Coalescer *co, uint64_t *calls, uint64_t *items, uint64_t *launches);
// Same contract as run_pipeline for a call of n <= coalescer_call_max() items.  A NULL input pointer with a non-zero row = rows
// of zeros ONLY for an input marked `optional` (an absent key_idx / rnd); otherwise CIRCL_HIP_EPARAM, as in run_pipeline.  Returns
// kNotCoalesced (> 0) when the call cannot join (too many blob bytes, a shape that differs from the coalescer's first call, a layout
// that could not be allocated right now): the caller then runs it through run_pipeline.
constexpr int kNotCoalesced = 1;
int coalesce_run(Coalescer *co, size_t n, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs,
                 const std::function<size_t(size_t)> &ws_bytes, const PipeOpts &opts, const std::function<int(Chunk &)> &launch);

// The ASYNCHRONOUS form (host_coalesce.hip): coalescer_async_start fixes the queue's arrays, workspace rule and launch (they outlive
// every call) and starts its dispatcher thread; coalesce_submit copies a call's rows in, records where its results go and returns a
// ticket sequence number at once (CIRCL_HIP_EAGAIN when every batch is busy and may_wait is false); coalescer_state: 1 done, 0 pending,
// < 0 the batch failed; coalescer_wait blocks the ONE thread that calls it until the ticket is done or timeout_us passed (< 0: no limit).
// coalesce_run on such a coalescer is submit + wait.
int coalescer_async_start(Coalescer *co, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs,
                          const std::function<size_t(size_t)> &ws_bytes, const PipeOpts &opts, const std::function<int(Chunk &)> &launch, bool want_eventfd);
int coalesce_submit(Coalescer *co, size_t n, const std::vector<HIn> &ins, const std::vector<HBlob> &blobs, const std::vector<HOut> &outs, uint64_t *seq,
                    bool may_wait);
int coalescer_state(const Coalescer *co, uint64_t seq);
int coalescer_wait(Coalescer *co, uint64_t seq, int64_t timeout_us);
bool coalescer_is_async(const Coalescer *co);
int coalescer_eventfd(const Coalescer *co);
// The completion flag of a coalesced batch written by the batch's OWN launch (keccak_dev.h TailFlag): enqueue() offers it here -- only for a
// batch that runs zero-copy, so that nothing of the device staging is left to wipe -- around the table's launch call; a one-launch route
// whose kernel keeps nothing secret in the workspace takes it (take_tail_flag) and passes it to its kernel.  Not taken: the coalescer
// enqueues its finish kernel as before.  Thread-local: the offer is made and consumed on the launching thread, within one call.
struct TailOffer {
    uint32_t *flag = nullptr;
    unsigned *count = nullptr;
    uint32_t value = 0;
    bool taken = false;
};
extern thread_local TailOffer g_tail_offer;
inline bool take_tail_flag(uint32_t **flag, unsigned **count, uint32_t *value) {
    if (!g_tail_offer.flag || g_tail_offer.taken) return false;
    g_tail_offer.taken = true;
    *flag = g_tail_offer.flag; *count = g_tail_offer.count; *value = g_tail_offer.value;
    return true;
}
// circl_hip_queue (include/circl_hip.h): the asynchronous form of the entry points that take their keys WITH the call.  The api unit that owns
// an operation fixes the queue's arrays and launch on `co` and says how long the rows are (api_mlkem.hip, api_hybrid.hip).
struct QueueShape {
    size_t key = 0, in = 0, out0 = 0, ss = 0;  // bytes per row: the key that comes with the item, its input, the ciphertext (encapsulation only), the secret
    bool key_secret = false, in_secret = false;
};
int kem_call_queue_start(bool decaps, int param, Coalescer *co, bool want_eventfd, QueueShape *shape);
int hyb_call_queue_start(bool decaps, int scheme, Coalescer *co, bool want_eventfd, QueueShape *shape);
// circl_hip_profile_call_stamps: CLOCK_MONOTONIC nanoseconds of the stages of the calling thread's last BLOCKING coalesced call
struct CallStamps { uint64_t enter = 0, reserved = 0, copied_in = 0, closed = 0, copies_in = 0, launched = 0, done = 0, copied_out = 0; };
extern std::atomic<bool> g_stamps_on;
extern thread_local CallStamps g_stamps;

// ... and process-wide, for the entry points that take their keys with every call (circl_hip_set_coalesce): a TLS server encapsulates
// to a DIFFERENT, ephemeral key in every handshake -- there is no table to attach a coalescer to.  One coalescer per (entry point,
// parameter set, device), created at its first small call.  nullptr: switched off (the default) or an unknown slot.
enum CallOp : int { kCoKemEncaps = 0, kCoKemDecaps, kCoDsaVerify, kCoDsaVerifyInternal, kCoHybEncaps, kCoHybDecaps, kCoOps };
Coalescer *call_coalescer(int op, int param_slot, int dev);  // param_slot: 0..3 (the entry point's own numbering of its parameter sets)
int call_coalescing_set(size_t max_items, uint32_t max_wait_us);  // 0: off AND drained (returns once no call is inside a process-wide coalescer)
// (these entry points only join a batch with every input present and leave the complaint about a NULL pointer to run_pipeline)
inline bool all_inputs_present(const std::vector<HIn> &ins) {
    for (auto &in : ins)
        if (!in.p) return false;
    return true;
}

// Contiguous split of [0,n) over the visible devices, one host thread each (pinned to the device's NUMA node), no collective.
// device = CIRCL_HIP_ALL_DEVICES: a call of at most `one_device_max` items goes to ONE device, round-robin (a thread and a launch per device
// for a handful of items cost more than they return); a larger one is split contiguously, one host thread per device.  The default suits
// operations of tens of microseconds per launch (ML-KEM, ML-DSA verification); the ones whose small batches are LATENCY-bound for
// hundreds of microseconds -- ML-DSA signing's rounds, the X25519 ladder of the hybrids, key generation -- pass kHeavyOneDeviceMax, so
// that a few hundred of them already use every device.
constexpr size_t kOneDeviceMax = 1024, kHeavyOneDeviceMax = 64;
int shard(size_t n, int device, const std::function<int(int dev, size_t lo, size_t cnt)> &fn, size_t one_device_max = kOneDeviceMax);


// host_runtime.hip: two library-owned non-blocking streams of device `dev` for calls that fork internally
int aux_streams(int dev, hipStream_t (&s)[2]);
int pipeline_streams(int dev, hipStream_t *h2d, hipStream_t *d2h, hipStream_t *compute);  // the device's shared copy / compute streams

// api_mlkem.hip: the smallest ML-KEM workspace for n items (scratch routes only) and the secret prefix of any ML-KEM workspace
size_t mlkem_ws_min_bytes(size_t n);
size_t mlkem_ws_secret_bytes(size_t n);

// api_x25519.hip: both X25519 ladders of a hybrid KEM operation (base point and peer point, same scalar) in one launch
int x25519_pair_dev(const uint8_t *d_scalar, const uint8_t *d_point, uint8_t *d_out_base, uint8_t *d_out_shared, uint8_t *d_ok, size_t n,
                    hipStream_t st);

}  // namespace host
}  // namespace circl
