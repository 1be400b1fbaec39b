// api_mldsa.hip -- ML-DSA (FIPS 204) and round-3 Dilithium entry points of the C ABI (include/circl_hip.h).
//
// No CPU path: every compute entry point launches the HIP kernels of mldsa_kernels.h / mldsa_sign_batched.h or fails
// with CIRCL_HIP_ENODEV.
#include "host_common.h"
#include "mldsa_kernels.h"
#include "mldsa_sign_batched.h"

using namespace circl::host;

namespace {

using circl::mlkem::KM_ITEM;
using circl::mlkem::KM_KEYED;
using circl::mlkem::KM_SHARED;

bool is_r3(int param) { return param == 2 || param == 3 || param == 5; }

// ---- device-resident ML-DSA verify ------------------------------------------------------------

// ML-DSA verify / keygen workspace: per-item intermediates, the ticket counter, and one 48 KB scratch slice
// (the sampled matrix rows) per resident workgroup of the persistent kernel.
int dsa_blocks_per_cu() {
    static const int v = [] {
        const char *e = getenv("CIRCL_HIP_DSA_BLOCKS_PER_CU");  // tuning aid
        const int x = e ? atoi(e) : 0;
        return x >= 1 && x <= kMaxBlocksPerCU ? x : kMaxBlocksPerCU;
    }();
    return v;
}
template <int MODE> size_t mldsa_groups(size_t n) { return (n + circl::mldsa::DG<MODE>::IT - 1) / circl::mldsa::DG<MODE>::IT; }
// scratch slices the workspace provides: enough for any visible device
template <int MODE> size_t mldsa_scratch_blocks(size_t n) {
    return std::min<size_t>(mldsa_groups<MODE>(n), (size_t)max_cu_count() * dsa_blocks_per_cu());
}
template <int MODE> size_t mldsa_item_ws_bytes(size_t n) {
    using G = circl::mldsa::DG<MODE>;
    return up256(n * G::MUW1) + up256(n * circl::mldsa::kBallStateBytes) + up256(n);
}
template <int MODE> size_t mldsa_ws_bytes(size_t n) {
    return mldsa_item_ws_bytes<MODE>(n) + 256 + mldsa_scratch_blocks<MODE>(n) * circl::mldsa::DG<MODE>::SCRATCH_BYTES + 256;  // + tr of a shared key
}
template <class Kern> unsigned dsa_resident_blocks(Kern kern, int lds_bytes) {
    const unsigned occ = resident_blocks(kern, lds_bytes);  // CUs of the current device * min(occupancy, kMaxBlocksPerCU)
    return std::min<unsigned>(occ, (unsigned)(cu_count() * dsa_blocks_per_cu()));
}
// key-table cache behind the verify workspace: packed A rows (whole groups of IT entries) and a 64-byte tr slot per entry
template <int MODE> size_t mldsa_table_bytes(size_t nkeys) {
    using G = circl::mldsa::DG<MODE>;
    const size_t padded = (nkeys + G::IT - 1) / G::IT * G::IT;
    return up256(padded * G::STREAMS * circl::mldsa::kPackedRowDwords * 4) + up256(nkeys * 64);
}

// KM_ITEM: every item its own public key.  KM_SHARED: n signatures under ONE public key (the reference's parsed-key case,
// where A and tr are cached in the PublicKey object, internal/dilithium.go:114-126): tr once per launch, ExpandA once per
// resident workgroup; 9 lane-permutations per item remain (mu, SampleInBall, c').  KM_KEYED: a table of nkeys public keys
// and an index per item: tr and ExpandA once per TABLE ENTRY, then the shared-key work per item.
template <int MODE, int KM>
int mldsa_verify_dev_impl(const uint8_t *pk, size_t nkeys, const uint32_t *key_idx, const uint8_t *sig, const uint8_t *msg_blob,
                          const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off, int internal, uint8_t *ok, size_t n,
                          void *ws, size_t ws_bytes, hipStream_t st) {
    using G = circl::mldsa::DG<MODE>;
    using namespace circl::mldsa;
    if (n == 0) return CIRCL_HIP_OK;
    if (KM == KM_KEYED && nkeys == 0) return CIRCL_HIP_EPARAM;
    const size_t need = mldsa_ws_bytes<MODE>(n) + (KM == KM_KEYED ? mldsa_table_bytes<MODE>(nkeys) : 0);
    if (ws_bytes < need || !aligned16(ws) || !aligned16(pk) || (reinterpret_cast<uintptr_t>(key_idx) & 3)) return CIRCL_HIP_EWORKSPACE;
    uint8_t *muw1 = static_cast<uint8_t *>(ws);
    uint8_t *ball = muw1 + up256(n * G::MUW1);
    uint8_t *fail = ball + up256(n * kBallStateBytes);
    unsigned *work = reinterpret_cast<unsigned *>(muw1 + mldsa_item_ws_bytes<MODE>(n));
    uint8_t *scratch = reinterpret_cast<uint8_t *>(work) + 256;
    uint8_t *tr = scratch + mldsa_scratch_blocks<MODE>(n) * G::SCRATCH_BYTES;  // shared key: 64 bytes behind the scratch slices
    uint32_t *key_rows = nullptr;
    const uint8_t *tr_arg = nullptr;
    HIP_TRY(hipMemsetAsync(work, 0, 256, st));
    const unsigned hb = (unsigned)((n + 255) / 256);
    if (KM == KM_KEYED) {
        const size_t padded = (nkeys + G::IT - 1) / G::IT * G::IT;
        key_rows = reinterpret_cast<uint32_t *>(muw1 + mldsa_ws_bytes<MODE>(n));
        uint8_t *key_tr = reinterpret_cast<uint8_t *>(key_rows) + up256(padded * G::STREAMS * kPackedRowDwords * 4);
        tr_arg = key_tr;
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_KEYTABLE, st);
        hipLaunchKernelGGL(mldsa_tr_table_kernel<MODE>, dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, st, pk, key_tr, nkeys);
        hipLaunchKernelGGL(mldsa_expand_keys_kernel<MODE>, dim3((unsigned)(padded / G::IT)), dim3(64), G::LDS_FIFO, st, pk, key_rows, nkeys);
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_HASH, st);
        if (KM == KM_SHARED) {
            hipLaunchKernelGGL(mldsa_tr_kernel<MODE>, dim3(1), dim3(64), 0, st, pk, tr);
            tr_arg = tr;
        }
        hipLaunchKernelGGL(mldsa_prep_kernel<MODE>, dim3(hb), dim3(256), 0, st, pk, sig, msg_blob, msg_off, ctx_blob, ctx_off, internal, muw1, ball,
                           fail, n, tr_arg, KM == KM_KEYED ? key_idx : (const uint32_t *)nullptr);
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_VERIFY, st);
        auto kern = mldsa_verify_kernel<MODE, 0, KM>;
        const unsigned vb = std::min<unsigned>((unsigned)mldsa_scratch_blocks<MODE>(n), dsa_resident_blocks(kern, G::LDS_V_TOTAL));
        hipLaunchKernelGGL(kern, dim3(vb), dim3(64), G::LDS_V_TOTAL, st, pk, sig, muw1, (const uint8_t *)ball, fail, scratch, work, n, key_idx,
                           (const uint32_t *)key_rows);
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_HASH, st);
        hipLaunchKernelGGL(mldsa_final_kernel<MODE>, dim3(hb), dim3(256), 0, st, sig, (const uint8_t *)muw1, (const uint8_t *)fail, ok, n);
    }
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

template <int MODE>
int mldsa_keygen_dev_impl(const uint8_t *seed32, uint8_t *pk, uint8_t *sk, size_t n, void *ws, size_t ws_bytes, hipStream_t st) {
    using Kg = circl::mldsa::KG<MODE>;
    using namespace circl::mldsa;
    if (n == 0) return CIRCL_HIP_OK;
    if (ws_bytes < mldsa_ws_bytes<MODE>(n) || !aligned16(ws) || !aligned16(seed32) || !aligned16(pk) || !aligned16(sk))
        return CIRCL_HIP_EWORKSPACE;
    uint8_t *es = static_cast<uint8_t *>(ws);
    unsigned *work = reinterpret_cast<unsigned *>(es + mldsa_item_ws_bytes<MODE>(n));
    uint8_t *scratch = reinterpret_cast<uint8_t *>(work) + 256;
    HIP_TRY(hipMemsetAsync(work, 0, 256, st));
    const unsigned hb = (unsigned)((n + 255) / 256);
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_HASH, st);
        hipLaunchKernelGGL(mldsa_keygen_seed_kernel<MODE>, dim3(hb), dim3(256), 0, st, seed32, es, n);
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_KEYGEN, st);
        auto kern = mldsa_keygen_kernel<MODE>;
        const unsigned kb = std::min<unsigned>((unsigned)mldsa_scratch_blocks<MODE>(n), dsa_resident_blocks(kern, Kg::LDS_TOTAL));
        hipLaunchKernelGGL(kern, dim3(kb), dim3(64), Kg::LDS_TOTAL, st, (const uint8_t *)es, pk, sk, scratch, work, n);
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_HASH, st);
        hipLaunchKernelGGL(mldsa_keygen_finish_kernel<MODE>, dim3(hb), dim3(256), 0, st, (const uint8_t *)pk, sk, n);
    }
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

#define DSA_SWITCH(param, CALL)      \
    switch (param) {                 \
    case 44: return CALL(44);        \
    case 65: return CALL(65);        \
    case 87: return CALL(87);        \
    case 2: return CALL(2);          \
    case 3: return CALL(3);          \
    case 5: return CALL(5);          \
    }

template <int KM>
int mldsa_verify_dev_any(int param, const uint8_t *pk, size_t nkeys, const uint32_t *key_idx, const uint8_t *sig, const uint8_t *msg_blob,
                         const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off, int internal, uint8_t *ok, size_t n, void *ws,
                         size_t wsb, hipStream_t st) {
    if (ndev() <= 0) return CIRCL_HIP_ENODEV;
#define CALL(M) mldsa_verify_dev_impl<M, KM>(pk, nkeys, key_idx, sig, msg_blob, msg_off, ctx_blob, ctx_off, internal, ok, n, ws, wsb, st)
    DSA_SWITCH(param, CALL)
#undef CALL
    return CIRCL_HIP_EPARAM;
}

size_t mldsa_ws_any(int param, size_t n) {
#define CALL(M) mldsa_ws_bytes<M>(n)
    DSA_SWITCH(param, CALL)
#undef CALL
    return 0;
}
size_t mldsa_table_any(int param, size_t nkeys) {
#define CALL(M) mldsa_table_bytes<M>(nkeys)
    DSA_SWITCH(param, CALL)
#undef CALL
    return 0;
}

// host-side context rules (mldsa65/dilithium.go:63-65, :116-118; round 3: sign.ErrContextNotSupported)
enum CtxRule { CTX_OK = 0, CTX_TOO_LONG = 1, CTX_UNSUPPORTED = 2 };
int check_contexts(int param, const uint8_t *ctx_blob, const uint64_t *ctx_off, size_t n) {
    if (!ctx_blob || !ctx_off) return CTX_OK;
    const bool r3 = is_r3(param);
    for (size_t i = 0; i < n; i++) {
        const uint64_t len = ctx_off[i + 1] - ctx_off[i];
        if (r3 && len) return CTX_UNSUPPORTED;
        if (len > 255) return CTX_TOO_LONG;
    }
    return CTX_OK;
}

PipeOpts dsa_opts(size_t dflt_chunk, bool secret, int depth = 3) {
    PipeOpts o;
    o.chunk_items = host_chunk_items(dflt_chunk);
    o.wipe_device = secret;
    o.depth = depth;
    return o;
}

// Host-buffer verify on one device: pk / sig rows and the message / context blobs of a chunk are staged together; the
// kernels keep using the caller's absolute offsets through rebased blob pointers.
template <int KM>
int mldsa_verify_host_one(int param, int dev, const uint8_t *pk, size_t nkeys, const uint32_t *key_idx, const uint8_t *sig, const uint8_t *msg_blob,
                          const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off, int internal, uint8_t *ok, size_t n) {
    const size_t PK = circl_hip_mldsa_pk_size(param), SIG = circl_hip_mldsa_sig_size(param);
    std::vector<HIn> ins;
    if (KM == KM_ITEM) ins.push_back({pk, PK});
    else ins.push_back({pk, PK * (KM == KM_KEYED ? nkeys : 1), false, true});
    ins.push_back({sig, SIG});
    if (KM == KM_KEYED) ins.push_back({reinterpret_cast<const uint8_t *>(key_idx), 4});
    return run_pipeline(dev, n, ins, {{msg_blob, msg_off}, {ctx_blob, ctx_blob ? ctx_off : nullptr}}, {{ok, 1}},
                        [&](size_t c) { return mldsa_ws_any(param, c) + (KM == KM_KEYED ? mldsa_table_any(param, nkeys) : 0); },
                        dsa_opts(size_t(1) << 13, false), [&](Chunk &c) {
                            return mldsa_verify_dev_any<KM>(param, c.in[0], nkeys, KM == KM_KEYED ? reinterpret_cast<const uint32_t *>(c.in[2]) : nullptr,
                                                            c.in[1], c.blob[0], c.off[0], c.blob[1], c.off[1], internal, c.out[0], c.cnt, c.ws, c.ws_bytes, c.st);
                        });
}

// ---- ML-DSA sign ------------------------------------------------------------------------------
constexpr int kSignBlocksPerCU = 8;

constexpr size_t kSignBatchedMin = 16;  // below this the single persistent kernel has less launch overhead

template <int MODE> size_t mldsa_sign_ws_core(size_t n) {
    using S = circl::mldsa::SG<MODE>;
    using B = circl::mldsa::SB<MODE>;
    const size_t persistent = up256(128 * n) + 256 + (size_t)max_cu_count() * kSignBlocksPerCU * S::SCRATCH_BYTES;
    const size_t tail_units = (size_t)max_cu_count() * kSignBlocksPerCU;  // speculative tail: best[] and one parked signature per unit
    const size_t batched = up256(n * B::PER_ITEM) + 256 + up256(4 * n) * 4 + 256 + up256(4 * tail_units) + tail_units * S::SPEC_STRIDE;
    return n < kSignBatchedMin ? persistent : persistent + batched;  // the batched path finishes its tail persistently
}
// ... followed by one byte per item: the "context refused" flags of mldsa_sign_prep_kernel
template <int MODE> size_t mldsa_sign_ws_bytes(size_t n) { return mldsa_sign_ws_core<MODE>(n) + up256(n); }

// Page-locked read-back slots for the per-round counts: a small pool, so that concurrent signing calls never share a slot.
struct PinnedCounts {
    std::mutex mu;
    std::vector<std::pair<int, uint32_t *>> free_slots;  // (device, slot)
    uint32_t *acquire(int dev) {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < free_slots.size(); i++)
                if (free_slots[i].first == dev) {
                    uint32_t *p = free_slots[i].second;
                    free_slots.erase(free_slots.begin() + i);
                    return p;
                }
        }
        uint32_t *p = nullptr;
        if (hipHostMalloc(reinterpret_cast<void **>(&p), 256, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return p;
    }
    void release(int dev, uint32_t *p) {
        std::lock_guard<std::mutex> lk(mu);
        free_slots.emplace_back(dev, p);
    }
};
PinnedCounts g_pinned_counts;

// Phase-split signing: rounds over the list of unsigned items (mldsa_sign_batched.h).  Synchronises the
// stream once per round to read the number of items that are still unsigned.
template <int MODE>
int mldsa_sign_batched(const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob,
                       const uint64_t *ctx_off, const uint8_t *rnd, int internal, uint8_t *sig, size_t n, void *ws, hipStream_t st,
                       bool shared, uint8_t *dead) {
    using namespace circl::mldsa;
    using B = SB<MODE>;
    constexpr int K = DP<MODE>::K, L = DP<MODE>::L;
    const int cus = cu_count();
    uint8_t *p = static_cast<uint8_t *>(ws);
    SignState S;
    S.shared = shared ? 1u : 0u;
    S.mr = p; p += up256(128 * n);
    S.A = reinterpret_cast<uint32_t *>(p); p += n * B::A_BYTES;
    S.sec = reinterpret_cast<uint32_t *>(p); p += n * B::SEC_BYTES;
    S.y = reinterpret_cast<uint32_t *>(p); p += n * B::Y_BYTES;
    S.w0 = reinterpret_cast<uint32_t *>(p); p += n * B::W0_BYTES;
    S.w1 = p; p += n * B::W1_BYTES;
    S.muw1 = p; p += n * B::MUW1_BYTES;
    S.cb = p; p += n * B::CB_BYTES;
    p = static_cast<uint8_t *>(ws) + up256(n * B::PER_ITEM) + 256;  // (mr was rounded up separately)
    S.attempts = reinterpret_cast<uint32_t *>(p); p += up256(4 * n);
    S.list[0] = reinterpret_cast<uint32_t *>(p); p += up256(4 * n);
    S.list[1] = reinterpret_cast<uint32_t *>(p); p += up256(4 * n);
    S.best = reinterpret_cast<uint32_t *>(p); p += up256(4 * n);
    S.count = reinterpret_cast<uint32_t *>(p); p += 256;
    unsigned *tail_work = reinterpret_cast<unsigned *>(p);          // persistent-kernel ticket counter
    uint8_t *tail_scratch = p + 256;
    const size_t tail_units = (size_t)cus * kSignBlocksPerCU;
    const size_t tail_units_ws = (size_t)max_cu_count() * kSignBlocksPerCU;  // what the workspace was sized for
    uint32_t *tail_best = reinterpret_cast<uint32_t *>(tail_scratch + tail_units_ws * SG<MODE>::SCRATCH_BYTES);
    uint8_t *tail_spec = reinterpret_cast<uint8_t *>(tail_best) + up256(4 * tail_units_ws);
    static const uint32_t tail_mult = [] {  // tuning aid: CIRCL_HIP_SIGN_TAIL = leftover items per CU handed to the persistent kernel
        const char *e = getenv("CIRCL_HIP_SIGN_TAIL");
        const int x = e ? atoi(e) : 0;
        return (uint32_t)(x >= 1 && x <= 1024 ? x : 2);  // measured optimum (tools/sign_tail_sweep.sh): 2 leftover items per CU
    }();
    const uint32_t tail_threshold = (uint32_t)cus * tail_mult;
    // Speculative rounds: once at most spec_target entries are left, a round costs its five dependent launches whatever
    // the count, so every item gets k = spec_target / items (<= 8) consecutive attempts per round.
    static const uint32_t spec_per_cu = [] {  // tuning aid: CIRCL_HIP_SIGN_SPEC = list entries per CU below which rounds speculate (0 = never)
        const char *e = getenv("CIRCL_HIP_SIGN_SPEC");
        const int x = e ? atoi(e) : -1;
        return (uint32_t)(x >= 0 && x <= 4096 ? x : 128);  // plateau 96 .. 256 at 2^16 items; 0 costs 15 % (ML-DSA-65)
    }();
    const uint32_t spec_target = (uint32_t)std::min<size_t>((size_t)cus * spec_per_cu, n);
    if (n >= (size_t(1) << circl::mldsa::kEntryShift)) return CIRCL_HIP_EPARAM;
    const unsigned nb256 = (unsigned)((n + 255) / 256);
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_HASH, st);
        hipLaunchKernelGGL(mldsa_sign_prep_kernel<MODE>, dim3(nb256), dim3(256), 0, st, sk, msg_blob, msg_off, ctx_blob, ctx_off, rnd, internal,
                           S.mr, n, shared ? 1 : 0, dead);
    }
    const uint32_t counts0[2] = {(uint32_t)n, 0};
    HIP_TRY(hipMemcpyAsync(S.count, counts0, 8, hipMemcpyHostToDevice, st));
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_SIGN, st);
        const size_t nkeys = shared ? 1 : n;
        hipLaunchKernelGGL(sign_expand_a_kernel<MODE>, dim3((unsigned)((nkeys * K * L + 255) / 256)), dim3(256), 0, st, sk, S, nkeys);
        hipLaunchKernelGGL(sign_secrets_kernel<MODE>, dim3((unsigned)n), dim3(64), 0, st, sk, S, n);
    }
    // The host needs the number of list entries only to size the next round's grids, and the kernels bound themselves
    // with the device-side count.  While the rounds are throughput-bound it runs one round ahead: round r is launched with
    // the count read back after round r - 2 (an over-estimate, the list only shrinks) while round r - 1 still executes.
    // In the latency-bound regime (and for the hand-over to the tail kernel) it waits for the exact count every round.
    const int dev = current_device();
    // error paths must not hand the pinned slot or the events back while copies into the slot (or the kernels) are still
    // in flight on the stream: a concurrent signer could pick the recycled slot up and read a stale count
    struct Guard {
        int dev;
        hipStream_t st;
        uint32_t *slot = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        ~Guard() {
            (void)hipStreamSynchronize(st);
            for (auto e : ev)
                if (e) (void)hipEventDestroy(e);
            if (slot) g_pinned_counts.release(dev, slot);
        }
    } guard{dev, st};
    guard.slot = g_pinned_counts.acquire(dev);
    if (!guard.slot) { g_err = "hipHostMalloc failed"; return CIRCL_HIP_EHIP; }
    volatile uint32_t *h_count = guard.slot;  // [0], [1]: counts after even / odd rounds
    HIP_TRY(hipEventCreateWithFlags(&guard.ev[0], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&guard.ev[1], hipEventDisableTiming));
    hipEvent_t *ev = guard.ev;
    int cur = 0;
    uint32_t upper = (uint32_t)n;   // bound on the current list's length (entries)
    unsigned k_cur = 1;             // entries per item in the current list
    bool exact = true;              // upper is the exact length
    int pending = 0;                // read-backs in flight: rounds (round - pending) .. (round - 1)
    for (int round = 0; upper > 0; round++) {
        if (round > 4096) { g_err = "mldsa sign: rejection loop did not terminate"; return CIRCL_HIP_EHIP; }
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_SIGN, st);
        const bool late = upper / k_cur <= std::max(spec_target, tail_threshold + tail_threshold / 2);
        if (late && !exact) {  // latency-bound rounds, or close to the hand-over: work with the exact count
            HIP_TRY(hipStreamSynchronize(st));
            upper = h_count[(round - 1) & 1];
            exact = true;
            pending = 0;
            if (upper == 0) break;
        }
        const uint32_t items = (upper + k_cur - 1) / k_cur;  // exact when `exact`
        if (exact && items <= tail_threshold) {
            // few items left: every leftover item gets its own wavefront(s), which run that item's remaining rejection
            // iterations to the end (continuing its nonce sequence).  The tail's duration is the unluckiest item's ~30
            // sequential attempts, with most of the chip idle: when the resident slots allow, 2, 4 or 8 wavefronts share
            // an item and try its attempts in parallel (first success wins).
            if (k_cur > 1) {  // the tail wants one entry per item: keep the first of each
                HIP_TRY(hipMemsetAsync(S.count + (cur ^ 1), 0, 4, st));
                hipLaunchKernelGGL(sign_compact_kernel, dim3((upper + 255) / 256), dim3(256), 0, st, S, cur, 0u, 1u);
                cur ^= 1;
            }
            const unsigned spec_w = (size_t)items * 8 <= tail_units ? 8u : (size_t)items * 4 <= tail_units ? 4u : (size_t)items * 2 <= tail_units ? 2u : 1u;
            HIP_TRY(hipMemsetAsync(tail_work, 0, 256, st));
            if (spec_w > 1) HIP_TRY(hipMemsetAsync(tail_best, 0xff, 4 * (size_t)items, st));
            hipLaunchKernelGGL(mldsa_sign_kernel<MODE>, dim3((unsigned)std::min<size_t>((size_t)items * spec_w, tail_units)), dim3(64),
                               SG<MODE>::LDS_TOTAL, st, sk, (const uint8_t *)S.mr, sig, tail_scratch, tail_work, (const uint32_t *)S.list[cur],
                               (const uint32_t *)S.attempts, (size_t)items, spec_w, tail_best, tail_spec, shared ? 1 : 0);
            if (spec_w > 1)
                hipLaunchKernelGGL(sign_tail_commit_kernel<MODE>, dim3(items), dim3(64), 0, st, (const uint32_t *)S.list[cur],
                                   (const uint32_t *)S.attempts, (const uint32_t *)tail_best, (const uint8_t *)tail_spec, sig, spec_w);
            break;
        }
        // attempts per item in the NEXT list: the survivors of this round are at most `items`
        unsigned k_next = 1;
        if (exact && spec_target > 0 && items <= spec_target)
            k_next = (unsigned)std::min<uint32_t>(circl::mldsa::kMaxSpec, std::max<uint32_t>(1u, spec_target / items));
        hipLaunchKernelGGL(sign_mask_kernel<MODE>, dim3((unsigned)(((size_t)upper * L + 255) / 256)), dim3(256), 0, st, S, cur);
        hipLaunchKernelGGL(sign_w_kernel<MODE>, dim3(upper), dim3(64), 0, st, S, cur);
        hipLaunchKernelGGL(sign_challenge_kernel<MODE>, dim3((upper + 255) / 256), dim3(256), 0, st, S, cur);
        hipLaunchKernelGGL(sign_finish_kernel<MODE>, dim3(upper), dim3(64), 0, st, S, cur, sig, k_cur);
        if (k_cur > 1) hipLaunchKernelGGL(sign_commit_kernel<MODE>, dim3(upper), dim3(64), 0, st, S, cur, sig);
        HIP_TRY(hipMemsetAsync(S.count + (cur ^ 1), 0, 4, st));
        hipLaunchKernelGGL(sign_compact_kernel, dim3((upper + 255) / 256), dim3(256), 0, st, S, cur, k_cur, k_next);
        cur ^= 1;
        HIP_TRY(hipMemcpyAsync(const_cast<uint32_t *>(&h_count[round & 1]), S.count + cur, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(ev[round & 1], st));
        pending++;
        if (k_cur == 1 && k_next == 1) {
            if (pending == 2) {  // the count after round - 1 has long arrived: it bounds the list of round + 1
                HIP_TRY(hipEventSynchronize(ev[(round - 1) & 1]));
                upper = h_count[(round - 1) & 1];
                exact = false;
                pending = 1;
            } else {
                exact = false;  // (right after an exact count, `upper` stays the bound for one more round)
            }
        } else {
            // the entry count changes with k: no stale bound is valid for the new list, so wait for this round's count
            HIP_TRY(hipStreamSynchronize(st));
            upper = h_count[round & 1];
            exact = true;
            pending = 0;
        }
        k_cur = k_next;
    }
    hipLaunchKernelGGL(mldsa_sign_zero_dead_kernel<MODE>, dim3(nb256), dim3(256), 0, st, sig, (const uint8_t *)dead, n);
    // The workspace held rho'', the NTT-domain secrets, the accepted attempts' y next to c~ (z - y = c s1) and parked
    // signatures: nothing key-equivalent stays behind in the caller's workspace.
    HIP_TRY(hipMemsetAsync(S.mr, 0, up256(128 * n), st));
    HIP_TRY(hipMemsetAsync(S.sec, 0, (size_t)((S.cb + n * B::CB_BYTES) - reinterpret_cast<uint8_t *>(S.sec)), st));  // (the matrix rows are public)
    HIP_TRY(hipMemsetAsync(tail_scratch, 0, tail_units_ws * SG<MODE>::SCRATCH_BYTES + up256(4 * tail_units_ws) + tail_units_ws * SG<MODE>::SPEC_STRIDE, st));
    HIP_TRY(hipStreamSynchronize(st));  // the pinned slot and the events go back to their pools
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

template <int MODE>
int mldsa_sign_dev_impl(const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob,
                        const uint64_t *ctx_off, const uint8_t *rnd, int internal, uint8_t *sig, size_t n, void *ws, size_t ws_bytes,
                        hipStream_t st, bool shared = false) {
    using S = circl::mldsa::SG<MODE>;
    using namespace circl::mldsa;
    if (n == 0) return CIRCL_HIP_OK;
    if (ws_bytes < mldsa_sign_ws_bytes<MODE>(n) || !aligned16(ws) || !aligned16(sk) || !aligned16(rnd) || rnd == nullptr)
        return CIRCL_HIP_EWORKSPACE;
    uint8_t *dead = static_cast<uint8_t *>(ws) + mldsa_sign_ws_core<MODE>(n);
    if (n >= kSignBatchedMin) return mldsa_sign_batched<MODE>(sk, msg_blob, msg_off, ctx_blob, ctx_off, rnd, internal, sig, n, ws, st, shared, dead);
    uint8_t *mr = static_cast<uint8_t *>(ws);
    unsigned *work = reinterpret_cast<unsigned *>(mr + up256(128 * n));
    uint8_t *scratch = mr + up256(128 * n) + 256;
    HIP_TRY(hipMemsetAsync(work, 0, 256, st));
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_HASH, st);
        hipLaunchKernelGGL(mldsa_sign_prep_kernel<MODE>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sk, msg_blob,
                           msg_off, ctx_blob, ctx_off, rnd, internal, mr, n, shared ? 1 : 0, dead);
    }
    {
        auto kern = mldsa_sign_kernel<MODE>;
        unsigned resident = resident_blocks(kern, S::LDS_TOTAL);
        resident = std::min<unsigned>(resident, (unsigned)(cu_count() * kSignBlocksPerCU));
        const unsigned blocks = (unsigned)std::min<size_t>(n, resident);
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_SIGN, st);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), S::LDS_TOTAL, st, sk, (const uint8_t *)mr, sig, scratch, work,
                           (const uint32_t *)nullptr, (const uint32_t *)nullptr, n, 1u, (uint32_t *)nullptr, (uint8_t *)nullptr, shared ? 1 : 0);
        hipLaunchKernelGGL(mldsa_sign_zero_dead_kernel<MODE>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sig, (const uint8_t *)dead, n);
    }
    HIP_TRY(hipMemsetAsync(ws, 0, mldsa_sign_ws_core<MODE>(n), st));  // rho'', NTT-domain secrets (see mldsa_sign_batched)
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

int mldsa_sign_dev_any(int param, const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob,
                       const uint64_t *ctx_off, const uint8_t *rnd, int internal, uint8_t *sig, size_t n, void *ws, size_t wsb,
                       hipStream_t st, bool shared = false) {
    if (ndev() <= 0) return CIRCL_HIP_ENODEV;
#define CALL(M) mldsa_sign_dev_impl<M>(sk, msg_blob, msg_off, ctx_blob, ctx_off, rnd, internal, sig, n, ws, wsb, st, shared)
    DSA_SWITCH(param, CALL)
#undef CALL
    return CIRCL_HIP_EPARAM;
}

size_t mldsa_sign_ws_any(int param, size_t n) {
#define CALL(M) mldsa_sign_ws_bytes<M>(n)
    DSA_SWITCH(param, CALL)
#undef CALL
    return 0;
}

int mldsa_sign_host(int param, const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob,
                    const uint64_t *ctx_off, const uint8_t *rnd, int internal, uint8_t *sig, size_t n, int device, bool shared = false) {
    const size_t SK = circl_hip_mldsa_sk_size(param), SIG = circl_hip_mldsa_sig_size(param);
    if (!SK) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    if (!internal && check_contexts(param, ctx_blob, ctx_off, n) != CTX_OK) return CIRCL_HIP_EPARAM;  // sign.ErrContextTooLong / ErrContextNotSupported
    const PipeOpts opts = dsa_opts(size_t(1) << 14, true, /*depth=*/1);  // the batched signer synchronises its stream: nothing to overlap
    std::vector<uint8_t> zeros;
    if (!rnd) zeros.assign(32 * std::min(n, opts.chunk_items), 0);  // deterministic signing: 32 zero bytes per item
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        std::vector<HIn> ins;
        ins.push_back(shared ? HIn{sk, SK, true, true} : HIn{sk + lo * SK, SK, true});
        ins.push_back(rnd ? HIn{rnd + lo * 32, 32, true} : HIn{zeros.data(), zeros.size(), false, true});
        return run_pipeline(dev, cnt, ins, {{msg_blob, msg_off + lo}, {ctx_blob, ctx_blob ? ctx_off + lo : nullptr}}, {{sig + lo * SIG, SIG}},
                            [&](size_t c) { return mldsa_sign_ws_any(param, c); }, opts, [&](Chunk &c) {
                                return mldsa_sign_dev_any(param, c.in[0], c.blob[0], c.off[0], c.blob[1], c.off[1], c.in[1], internal, c.out[0], c.cnt, c.ws,
                                                          c.ws_bytes, c.st, shared);
                            });
    });
}

int mldsa_verify_host(int param, const uint8_t *pk, const uint8_t *sig, const uint8_t *msg_blob, const uint64_t *msg_off,
                      const uint8_t *ctx_blob, const uint64_t *ctx_off, int internal, uint8_t *ok, size_t n, int device, bool shared) {
    const size_t PK = circl_hip_mldsa_pk_size(param), SIG = circl_hip_mldsa_sig_size(param);
    if (!PK) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    if (!internal && check_contexts(param, ctx_blob, ctx_off, n) == CTX_UNSUPPORTED) return CIRCL_HIP_EPARAM;  // round 3: sign.ErrContextNotSupported
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        if (shared)
            return mldsa_verify_host_one<KM_SHARED>(param, dev, pk, 1, nullptr, sig + lo * SIG, msg_blob, msg_off + lo, ctx_blob,
                                                    ctx_blob ? ctx_off + lo : nullptr, internal, ok + lo, cnt);
        return mldsa_verify_host_one<KM_ITEM>(param, dev, pk + lo * PK, 0, nullptr, sig + lo * SIG, msg_blob, msg_off + lo, ctx_blob,
                                              ctx_blob ? ctx_off + lo : nullptr, internal, ok + lo, cnt);
    });
}

}  // namespace

// =============================================================================================
extern "C" {

// 44 / 65 / 87 = ML-DSA; 2 / 3 / 5 = round-3 Dilithium2/3/5 (32-byte tr and c~)
size_t circl_hip_mldsa_pk_size(int param) { return param == 44 || param == 2 ? 1312 : param == 65 || param == 3 ? 1952 : param == 87 || param == 5 ? 2592 : 0; }
size_t circl_hip_mldsa_sig_size(int param) {
    return param == 44 || param == 2 ? 2420 : param == 65 ? 3309 : param == 3 ? 3293 : param == 87 ? 4627 : param == 5 ? 4595 : 0;
}
size_t circl_hip_mldsa_sk_size(int param) {
    return param == 44 ? 2560 : param == 2 ? 2528 : param == 65 ? 4032 : param == 3 ? 4000 : param == 87 ? 4896 : param == 5 ? 4864 : 0;
}

size_t circl_hip_mldsa_workspace_size(int param, size_t n) { return mldsa_ws_any(param, n); }
size_t circl_hip_mldsa_keyed_workspace_size(int param, size_t n, size_t nkeys) {
    const size_t base = mldsa_ws_any(param, n);
    return base ? base + mldsa_table_any(param, nkeys) : 0;
}

int circl_hip_mldsa_verify_dev(int param, const uint8_t *d_pk, const uint8_t *d_sig, const uint8_t *d_msg_blob,
                               const uint64_t *d_msg_off, const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off, uint8_t *d_ok,
                               size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    return mldsa_verify_dev_any<KM_ITEM>(param, d_pk, 0, nullptr, d_sig, d_msg_blob, d_msg_off, d_ctx_blob, d_ctx_off, 0, d_ok, n, d_ws, ws_bytes,
                                         static_cast<hipStream_t>(stream));
}
int circl_hip_mldsa_verify_shared_dev(int param, const uint8_t *d_pk, const uint8_t *d_sig, const uint8_t *d_msg_blob,
                                      const uint64_t *d_msg_off, const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off, uint8_t *d_ok,
                                      size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    return mldsa_verify_dev_any<KM_SHARED>(param, d_pk, 1, nullptr, d_sig, d_msg_blob, d_msg_off, d_ctx_blob, d_ctx_off, 0, d_ok, n, d_ws, ws_bytes,
                                           static_cast<hipStream_t>(stream));
}
int circl_hip_mldsa_verify_keyed_dev(int param, const uint8_t *d_pk_table, size_t nkeys, const uint32_t *d_key_idx, const uint8_t *d_sig,
                                     const uint8_t *d_msg_blob, const uint64_t *d_msg_off, const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off,
                                     uint8_t *d_ok, size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    return mldsa_verify_dev_any<KM_KEYED>(param, d_pk_table, nkeys, d_key_idx, d_sig, d_msg_blob, d_msg_off, d_ctx_blob, d_ctx_off, 0, d_ok, n, d_ws,
                                          ws_bytes, static_cast<hipStream_t>(stream));
}

int circl_hip_mldsa_verify(int param, const uint8_t *pk, const uint8_t *sig, const uint8_t *msg_blob, const uint64_t *msg_off,
                           const uint8_t *ctx_blob, const uint64_t *ctx_off, uint8_t *ok, size_t n, int device) {
    return mldsa_verify_host(param, pk, sig, msg_blob, msg_off, ctx_blob, ctx_off, 0, ok, n, device, false);
}
int circl_hip_mldsa_verify_internal(int param, const uint8_t *pk, const uint8_t *sig, const uint8_t *msg_blob,
                                    const uint64_t *msg_off, uint8_t *ok, size_t n, int device) {
    return mldsa_verify_host(param, pk, sig, msg_blob, msg_off, nullptr, nullptr, 1, ok, n, device, false);
}
int circl_hip_mldsa_verify_shared(int param, const uint8_t *pk, const uint8_t *sig, const uint8_t *msg_blob, const uint64_t *msg_off,
                                  const uint8_t *ctx_blob, const uint64_t *ctx_off, uint8_t *ok, size_t n, int device) {
    return mldsa_verify_host(param, pk, sig, msg_blob, msg_off, ctx_blob, ctx_off, 0, ok, n, device, true);
}
int circl_hip_mldsa_verify_keyed(int param, const uint8_t *pk_table, size_t nkeys, const uint32_t *key_idx, const uint8_t *sig,
                                 const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off, uint8_t *ok,
                                 size_t n, int device) {
    const size_t PK = circl_hip_mldsa_pk_size(param), SIG = circl_hip_mldsa_sig_size(param);
    if (!PK) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    if (nkeys == 0 || nkeys > 0xffffffffull) return CIRCL_HIP_EPARAM;
    for (size_t i = 0; i < n; i++)
        if (key_idx[i] >= nkeys) return CIRCL_HIP_EPARAM;
    if (check_contexts(param, ctx_blob, ctx_off, n) == CTX_UNSUPPORTED) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return mldsa_verify_host_one<KM_KEYED>(param, dev, pk_table, nkeys, key_idx + lo, sig + lo * SIG, msg_blob, msg_off + lo, ctx_blob,
                                               ctx_blob ? ctx_off + lo : nullptr, 0, ok + lo, cnt);
    });
}

int circl_hip_mldsa_keygen_dev(int param, const uint8_t *d_seed32, uint8_t *d_pk, uint8_t *d_sk, size_t n, void *d_ws, size_t ws_bytes,
                               void *stream) {
    if (ndev() <= 0) return CIRCL_HIP_ENODEV;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define CALL(M) mldsa_keygen_dev_impl<M>(d_seed32, d_pk, d_sk, n, d_ws, ws_bytes, st)
    DSA_SWITCH(param, CALL)
#undef CALL
    return CIRCL_HIP_EPARAM;
}

int circl_hip_mldsa_keygen(int param, const uint8_t *seed32, uint8_t *pk, uint8_t *sk, size_t n, int device) {
    const size_t PK = circl_hip_mldsa_pk_size(param), SK = circl_hip_mldsa_sk_size(param);
    if (!PK) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{seed32 + lo * 32, 32, true}}, {}, {{pk + lo * PK, PK}, {sk + lo * SK, SK, true}},
                            [&](size_t c) { return mldsa_ws_any(param, c); }, dsa_opts(size_t(1) << 14, true),
                            [&](Chunk &c) { return circl_hip_mldsa_keygen_dev(param, c.in[0], c.out[0], c.out[1], c.cnt, c.ws, c.ws_bytes, c.st); });
    });
}

size_t circl_hip_mldsa_sign_workspace_size(int param, size_t n) { return mldsa_sign_ws_any(param, n); }

int circl_hip_mldsa_sign_dev(int param, const uint8_t *d_sk, const uint8_t *d_msg_blob, const uint64_t *d_msg_off,
                             const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off, const uint8_t *d_rnd, int internal, uint8_t *d_sig,
                             size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    return mldsa_sign_dev_any(param, d_sk, d_msg_blob, d_msg_off, d_ctx_blob, d_ctx_off, d_rnd, internal, d_sig, n, d_ws, ws_bytes,
                              static_cast<hipStream_t>(stream));
}
int circl_hip_mldsa_sign_shared_dev(int param, const uint8_t *d_sk, const uint8_t *d_msg_blob, const uint64_t *d_msg_off,
                                    const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off, const uint8_t *d_rnd, int internal, uint8_t *d_sig,
                                    size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    return mldsa_sign_dev_any(param, d_sk, d_msg_blob, d_msg_off, d_ctx_blob, d_ctx_off, d_rnd, internal, d_sig, n, d_ws, ws_bytes,
                              static_cast<hipStream_t>(stream), true);
}

int circl_hip_mldsa_sign(int param, const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob,
                         const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sig, size_t n, int device) {
    return mldsa_sign_host(param, sk, msg_blob, msg_off, ctx_blob, ctx_off, rnd, 0, sig, n, device);
}
int circl_hip_mldsa_sign_shared(int param, const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob,
                                const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sig, size_t n, int device) {
    return mldsa_sign_host(param, sk, msg_blob, msg_off, ctx_blob, ctx_off, rnd, 0, sig, n, device, true);
}
int circl_hip_mldsa_sign_internal(int param, const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *rnd,
                                  uint8_t *sig, size_t n, int device) {
    return mldsa_sign_host(param, sk, msg_blob, msg_off, nullptr, nullptr, rnd, 1, sig, n, device);
}

}  // extern "C"
