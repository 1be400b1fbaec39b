// api_mldsa.hip -- ML-DSA (FIPS 204) and round-3 Dilithium entry points of the C ABI (include/circl_hip.h).
//
// No CPU path: every compute entry point launches the HIP kernels of mldsa_kernels.h / mldsa_sign_batched.h or fails
// with CIRCL_HIP_ENODEV.
#include "host_common.h"
#include "keytable.h"
#include "mldsa_kernels.h"
#include "mldsa_sign_batched.h"

using namespace circl::host;
using circl::KeyIdx;

namespace {

using circl::mlkem::KM_ITEM;
using circl::mlkem::KM_KEYED;
using circl::mlkem::KM_SHARED;

bool is_r3(int param) { return param == 2 || param == 3 || param == 5; }

// ---- device-resident ML-DSA verify ------------------------------------------------------------

// ML-DSA verify / keygen workspace: per-item intermediates, the ticket counter, and one 48 KB scratch slice
// (the sampled matrix rows) per resident workgroup of the persistent kernel.
int dsa_blocks_per_cu() { return kMaxBlocksPerCU; }
template <int MODE> size_t mldsa_groups(size_t n) { return (n + circl::mldsa::DG<MODE>::IT - 1) / circl::mldsa::DG<MODE>::IT; }
// scratch slices the workspace provides: enough for any visible device
template <int MODE> size_t mldsa_scratch_blocks(size_t n) {
    return std::min<size_t>(mldsa_groups<MODE>(n), (size_t)max_cu_count() * dsa_blocks_per_cu());
}
template <int MODE> size_t mldsa_item_ws_bytes(size_t n) {
    using G = circl::mldsa::DG<MODE>;
    return up256(n * G::MUW1) + up256(n * circl::mldsa::kBallStateBytes) + up256(n);
}
// up to here a lane per item leaves the SIMDs at or below one wavefront each: hash chains go on lane pairs (keccak_f1600_split)
constexpr size_t kMidBatch = size_t(1) << 14;
constexpr size_t kLongCtlBytes = (sizeof(circl::mldsa::LongCtl) + 255) & ~size_t(255);
template <int MODE> size_t mldsa_ws_bytes(size_t n) {
    return mldsa_item_ws_bytes<MODE>(n) + 256 + mldsa_scratch_blocks<MODE>(n) * circl::mldsa::DG<MODE>::SCRATCH_BYTES + 256 + kLongCtlBytes;  // + tr of a shared key + long-message list
}
// mu of the batch's long messages ahead of the per-lane kernels (mldsa_kernels.h, kLongMsg): scan, then two messages per wavefront.
// The two halves are separate so that a small verification batch can run the second one beside its other kernels (SideStream).
inline int mldsa_long_scan(const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off, int internal, circl::mldsa::LongCtl *ctl, size_t n,
                           hipStream_t st) {
    using namespace circl::mldsa;
    HIP_TRY(hipMemsetAsync(ctl, 0, 256, st));
    hipLaunchKernelGGL(mldsa_long_scan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, msg_off, ctx_blob, ctx_off, internal, n, ctl);
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}
template <int TRW>
int mldsa_long_mu(const uint8_t *tr_base, size_t tr_stride, KeyIdx key_idx, const uint8_t *pk, size_t pk_stride, int pk_words,
                  const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off, int internal, uint8_t *mu_out,
                  size_t mu_stride, const circl::mldsa::LongCtl *ctl, size_t n, hipStream_t st) {
    using namespace circl::mldsa;
    const unsigned grid = (unsigned)std::min<size_t>((std::min<size_t>(n, kLongCap) + 1) / 2, (size_t)cu_count() * 8);
    hipLaunchKernelGGL(mldsa_mu_long_kernel<TRW>, dim3(grid), dim3(64), 0, st, tr_base, tr_stride, key_idx, pk, pk_stride, pk_words, msg_blob, msg_off,
                       ctx_blob, ctx_off, internal, mu_out, mu_stride, ctl);
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}
template <int TRW>
int mldsa_long_prepass(const uint8_t *tr_base, size_t tr_stride, KeyIdx key_idx, const uint8_t *pk, size_t pk_stride, int pk_words,
                       const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off, int internal, uint8_t *mu_out,
                       size_t mu_stride, circl::mldsa::LongCtl *ctl, size_t n, hipStream_t st) {
    if (int rc = mldsa_long_scan(msg_off, ctx_blob, ctx_off, internal, ctl, n, st)) return rc;
    return mldsa_long_mu<TRW>(tr_base, tr_stride, key_idx, pk, pk_stride, pk_words, msg_blob, msg_off, ctx_blob, ctx_off, internal, mu_out, mu_stride, ctl, n, st);
}
// One of the library's auxiliary streams borrowed for the length of a call: begin() orders it after what the caller's stream
// holds so far, join() orders the caller's stream after it; the destructor joins on every path that did not (error returns),
// falling back to a host-side wait if the event calls themselves fail.
struct SideStream {
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_done = nullptr;
    bool open = false;
    bool begin(hipStream_t st) {
        hipStream_t aux[2];
        if (aux_streams(current_device(), aux) != CIRCL_HIP_OK) return false;
        if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_done, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev_fork, st) != hipSuccess ||
            hipStreamWaitEvent(aux[0], ev_fork, 0) != hipSuccess) {
            (void)hipGetLastError();
            return false;  // (nothing was enqueued on the side stream: the caller stays on its own stream)
        }
        main = st;
        side = aux[0];
        open = true;
        return true;
    }
    void join() {
        if (!open) return;
        open = false;
        if (hipEventRecord(ev_done, side) != hipSuccess || hipStreamWaitEvent(main, ev_done, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(side);
        }
    }
    ~SideStream() {
        join();
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_done) (void)hipEventDestroy(ev_done);
    }
};
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
// Resident public keys: batches up to 2^CIRCL_HIP_DSA_CHAIN items (0 = never) are verified in one launch, a workgroup of K + 1
// wavefronts per item (mldsa_verify_chain_kernel)
// ... and, with keys that are NOT parsed beforehand (tr and the matrix expansion inside the workgroup), up to 2^CIRCL_HIP_DSA_CHAIN_ITEM
// (measured, tools/dsa_latency.py at 2^9: ML-DSA-44 142 against 224 us, ML-DSA-65 170 / 205, ML-DSA-87 296 / 218 -- so 2^9 up to K = 6, 2^8 beyond)
inline size_t dsa_chain_batch(bool resident = true, int k = 6) {
    static const int lg_r = env_int("CIRCL_HIP_DSA_CHAIN", 11, 0, 16), lg_i = env_int("CIRCL_HIP_DSA_CHAIN_ITEM", -1, -1, 16);
    const int lg = resident ? lg_r : (lg_i >= 0 ? lg_i : (k <= 6 ? 9 : 8));
    return lg <= 0 ? size_t(0) : size_t(1) << lg;
}
template <int MODE, int KM>
int mldsa_verify_dev_impl(const uint8_t *pk, size_t nkeys, const uint32_t *key_idx, const uint8_t *sig, const uint8_t *msg_blob,
                          const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off, int internal, uint8_t *ok, size_t n,
                          void *ws, size_t ws_bytes, hipStream_t st, const circl_hip_keytable *cached = nullptr) {
    // cached (KM_KEYED only): a key table that lives across calls (keytable.h) -- pk, the expanded rows and tr come from it, the
    // workspace needs no table tail, and key_idx == nullptr means entry 0 for every item
    using G = circl::mldsa::DG<MODE>;
    using namespace circl::mldsa;
    if (n == 0) return CIRCL_HIP_OK;
    if (KM == KM_KEYED && nkeys == 0) return CIRCL_HIP_EPARAM;
    if (cached) pk = cached->d_keys;
    const size_t need = mldsa_ws_bytes<MODE>(n) + (KM == KM_KEYED && !cached ? mldsa_table_bytes<MODE>(nkeys) : 0);
    if (ws_bytes < need || !aligned16(ws) || !aligned16(pk) || (reinterpret_cast<uintptr_t>(key_idx) & 3)) return CIRCL_HIP_EWORKSPACE;
    const KeyIdx kx{KM == KM_KEYED ? key_idx : nullptr, nkeys ? (uint32_t)(nkeys - 1) : 0u};  // a device index vector is bounded to the table on every read
    uint8_t *muw1 = static_cast<uint8_t *>(ws);
    uint8_t *ball = muw1 + up256(n * G::MUW1);
    uint8_t *fail = ball + up256(n * kBallStateBytes);
    unsigned *work = reinterpret_cast<unsigned *>(muw1 + mldsa_item_ws_bytes<MODE>(n));
    uint8_t *scratch = reinterpret_cast<uint8_t *>(work) + 256;
    uint8_t *tr = scratch + mldsa_scratch_blocks<MODE>(n) * G::SCRATCH_BYTES;  // shared key: 64 bytes behind the scratch slices
    LongCtl *lctl = reinterpret_cast<LongCtl *>(tr + 256);
    uint32_t *key_rows = nullptr;
    const uint8_t *tr_arg = nullptr;
    if (KM == KM_KEYED && cached && n <= dsa_chain_batch()) {
        // a resident key table, a small batch: the whole verification of an item in ONE launch (mldsa_verify_chain_kernel)
        const size_t padded = (nkeys + G::IT - 1) / G::IT * G::IT;
        const uint32_t *rows = reinterpret_cast<const uint32_t *>(cached->d_table);
        const uint8_t *key_tr = cached->d_table + up256(padded * G::STREAMS * kPackedRowDwords * 4);
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_VERIFY, st);
        circl::TailFlag tail{nullptr, nullptr, 0};  // (the whole call is this ONE launch and verification holds no secret: a coalesced batch's flag may ride on it)
        take_tail_flag(&tail.flag, &tail.count, &tail.value);
        hipLaunchKernelGGL(mldsa_verify_chain_kernel<MODE>, dim3((unsigned)n), dim3((DP<MODE>::K + 1) * 64), 0, st, pk, kx, rows, key_tr, sig, msg_blob,
                           msg_off, ctx_blob, ctx_off, internal, ok, n, (uint8_t *)nullptr, (size_t)G::PK, tail);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
    // (the route writes item t's rows into scratch slice t / IT: only while the workspace holds a slice per IT items -- a
    // CIRCL_HIP_DSA_CHAIN_ITEM beyond the slices the workspace guarantees falls through to the scratch routes)
    if ((KM == KM_ITEM || KM == KM_SHARED) && n <= dsa_chain_batch(false, DP<MODE>::K) && mldsa_groups<MODE>(n) <= mldsa_scratch_blocks<MODE>(n)) {
        // every item under its own, unparsed key (or all under ONE unparsed key: stride 0), a small batch: the same kernel with tr and
        // the matrix expansion inside the workgroup (the rows of item t in its part of scratch slice t / IT: the workspace holds a
        // slice per IT items at these sizes)
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_VERIFY, st);
        hipLaunchKernelGGL((mldsa_verify_chain_kernel<MODE, false>), dim3((unsigned)n), dim3((DP<MODE>::K + 2) * 64), 0, st, pk, KeyIdx{},
                           (const uint32_t *)nullptr, (const uint8_t *)nullptr, sig, msg_blob, msg_off, ctx_blob, ctx_off, internal, ok, n, scratch,
                           KM == KM_SHARED ? size_t(0) : (size_t)G::PK);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
    SideStream side;  // (declared ahead of every early return below: its destructor joins)
    HIP_TRY(hipMemsetAsync(work, 0, 256, st));
    const unsigned hb = (unsigned)((n + 255) / 256);
    if (KM == KM_KEYED) {
        const size_t padded = (nkeys + G::IT - 1) / G::IT * G::IT;
        key_rows = reinterpret_cast<uint32_t *>(cached ? cached->d_table : muw1 + mldsa_ws_bytes<MODE>(n));
        uint8_t *key_tr = reinterpret_cast<uint8_t *>(key_rows) + up256(padded * G::STREAMS * kPackedRowDwords * 4);
        tr_arg = key_tr;
        if (!cached) {
            ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_KEYTABLE, st);
            hipLaunchKernelGGL(mldsa_tr_table_kernel<MODE>, dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, st, pk, key_tr, nkeys);
            hipLaunchKernelGGL(mldsa_expand_keys_kernel<MODE>, dim3((unsigned)(padded / G::IT)), dim3(64), G::LDS_FIFO, st, pk, key_rows, nkeys);
        }
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_HASH, st);
        const int internal_eff = DP<MODE>::NIST ? internal : 1;  // round 3: mu = CRH(tr || msg)
        if (int rc = mldsa_long_scan(msg_off, ctx_blob, ctx_off, internal_eff, lctl, n, st)) return rc;
        // small batches: every mu comes from the cooperative pre-pass (kSmallMu), and nothing before the final hash reads it --
        // so tr and mu run on a side stream next to SampleInBall's sponge and the verify kernel (n=1: ~75 us off the chain)
        hipStream_t mu_st = st;
        if (n <= kSmallMu && side.begin(st)) mu_st = side.side;
        if (KM == KM_SHARED) {
            hipLaunchKernelGGL(mldsa_tr_kernel<MODE>, dim3(1), dim3(64), 0, mu_st, pk, tr);
            tr_arg = tr;
        }
        {
            const int rc = tr_arg ? mldsa_long_mu<DP<MODE>::TR / 8>(tr_arg, kx ? 64 : 0, kx, nullptr, 0, 0, msg_blob, msg_off, ctx_blob, ctx_off,
                                                                    internal_eff, muw1, G::MUW1, lctl, n, mu_st)
                                  : mldsa_long_mu<DP<MODE>::TR / 8>(nullptr, 0, KeyIdx{}, pk, G::PK, G::PK / 8, msg_blob, msg_off, ctx_blob, ctx_off, internal_eff,
                                                                    muw1, G::MUW1, lctl, n, mu_st);
            if (rc) return rc;
        }
        // (with the side stream open, prep never takes its own mu path: n <= kSmallMu means every mu is pre-made, or -- more than
        // kLongCap items cannot occur below kSmallMu -- none is; so it does not read tr before the side stream wrote it)
        // medium batches of distinct keys: tr on lane pairs first (into the items' ball slots, which prep reads before it writes them)
        size_t tr_stride = 0;
        if (KM == KM_ITEM && n > kSmallMu && n <= kMidBatch) {
            hipLaunchKernelGGL(mldsa_tr_split_kernel<MODE>, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, st, pk, ball, (size_t)kBallStateBytes, n);
            tr_arg = ball;
            tr_stride = kBallStateBytes;
        }
        hipLaunchKernelGGL(mldsa_prep_kernel<MODE>, dim3(hb), dim3(256), 0, st, pk, sig, msg_blob, msg_off, ctx_blob, ctx_off, internal, muw1, ball,
                           fail, n, tr_arg, kx, (const LongCtl *)lctl, tr_stride);
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_VERIFY, st);
        auto kern = mldsa_verify_kernel<MODE, 0, KM>;
        const unsigned vb = std::min<unsigned>((unsigned)mldsa_scratch_blocks<MODE>(n), dsa_resident_blocks(kern, G::LDS_V_TOTAL));
        hipLaunchKernelGGL(kern, dim3(vb), dim3(64), G::LDS_V_TOTAL, st, pk, sig, muw1, (const uint8_t *)ball, fail, scratch, work, n, kx,
                           (const uint32_t *)key_rows);
    }
    side.join();
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_HASH, st);
        if (n <= kSmallMu)
            hipLaunchKernelGGL(mldsa_final_coop_kernel<MODE>, dim3((unsigned)((n + 1) / 2)), dim3(64), 0, st, sig, (const uint8_t *)muw1, (const uint8_t *)fail, ok, n);
        else if (n <= kMidBatch)
            hipLaunchKernelGGL(mldsa_final_split_kernel<MODE>, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, st, sig, (const uint8_t *)muw1, (const uint8_t *)fail, ok, n);
        else
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
    // small batches: the whole key generation of an item in ONE launch, a workgroup of K wavefronts per key (mldsa_keygen_chain_kernel);
    // up to 2^CIRCL_HIP_DSA_KEYGEN_CHAIN keys (0: never), and only while the workspace holds a scratch slice per IT items
    static const size_t chain_keys = [] { const int lg = env_int("CIRCL_HIP_DSA_KEYGEN_CHAIN", 9, 0, 12); return lg <= 0 ? size_t(0) : size_t(1) << lg; }();
    if (n <= chain_keys && mldsa_groups<MODE>(n) <= mldsa_scratch_blocks<MODE>(n)) {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_KEYGEN, st);
        hipLaunchKernelGGL(mldsa_keygen_chain_kernel<MODE>, dim3((unsigned)n), dim3(DP<MODE>::K * 64), 0, st, seed32, es, pk, sk, scratch, n);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
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
        // tr = H(pk): up to 2^14 keys the chain of a lone lane is the launch (two keys per wavefront up to kSmallMu, lane pairs beyond)
        if (n <= kSmallMu) hipLaunchKernelGGL(mldsa_keygen_finish_small_kernel<MODE>, dim3((unsigned)((n + 1) / 2)), dim3(64), 0, st, (const uint8_t *)pk, sk, n, 1);
        else if (n <= kMidBatch) hipLaunchKernelGGL(mldsa_keygen_finish_small_kernel<MODE>, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, st, (const uint8_t *)pk, sk, n, 2);
        else hipLaunchKernelGGL(mldsa_keygen_finish_kernel<MODE>, dim3(hb), dim3(256), 0, st, (const uint8_t *)pk, sk, n);
    }
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

// PrivateKey.Public() over a batch (sign/mldsa/mldsa65/internal/dilithium.go:473-484): t1 recomputed from the packed private keys
template <int MODE>
int mldsa_public_dev_impl(const uint8_t *sk, uint8_t *pk, size_t n, void *ws, size_t ws_bytes, hipStream_t st) {
    using Kg = circl::mldsa::KG<MODE>;
    using namespace circl::mldsa;
    if (n == 0) return CIRCL_HIP_OK;
    if (ws_bytes < mldsa_ws_bytes<MODE>(n) || !aligned16(ws) || !aligned16(sk) || !aligned16(pk)) return CIRCL_HIP_EWORKSPACE;
    unsigned *work = reinterpret_cast<unsigned *>(static_cast<uint8_t *>(ws) + mldsa_item_ws_bytes<MODE>(n));
    uint8_t *scratch = reinterpret_cast<uint8_t *>(work) + 256;
    HIP_TRY(hipMemsetAsync(work, 0, 256, st));
    ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_KEYGEN, st);
    auto kern = mldsa_keygen_kernel<MODE, true>;
    const unsigned kb = std::min<unsigned>((unsigned)mldsa_scratch_blocks<MODE>(n), dsa_resident_blocks(kern, Kg::LDS_TOTAL));
    hipLaunchKernelGGL(kern, dim3(kb), dim3(64), Kg::LDS_TOTAL, st, sk, pk, (uint8_t *)nullptr, scratch, work, n);
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

template <int MODE> int mldsa_table_build(circl_hip_keytable *t, hipStream_t st) {
    using G = circl::mldsa::DG<MODE>;
    using namespace circl::mldsa;
    const size_t padded = (t->nkeys + G::IT - 1) / G::IT * G::IT;
    uint32_t *key_rows = reinterpret_cast<uint32_t *>(t->d_table);
    uint8_t *key_tr = t->d_table + up256(padded * G::STREAMS * kPackedRowDwords * 4);
    hipLaunchKernelGGL(mldsa_tr_table_kernel<MODE>, dim3((unsigned)((t->nkeys + 255) / 256)), dim3(256), 0, st, (const uint8_t *)t->d_keys, key_tr, t->nkeys);
    hipLaunchKernelGGL(mldsa_expand_keys_kernel<MODE>, dim3((unsigned)(padded / G::IT)), dim3(64), G::LDS_FIFO, st, (const uint8_t *)t->d_keys, key_rows, t->nkeys);
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

template <int KM>
int mldsa_verify_dev_any(int param, const uint8_t *pk, size_t nkeys, const uint32_t *key_idx, const uint8_t *sig, const uint8_t *msg_blob,
                         const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off, int internal, uint8_t *ok, size_t n, void *ws,
                         size_t wsb, hipStream_t st, const circl_hip_keytable *cached = nullptr) {
    if (ndev() <= 0) return CIRCL_HIP_ENODEV;
#define CALL(M) mldsa_verify_dev_impl<M, KM>(pk, nkeys, key_idx, sig, msg_blob, msg_off, ctx_blob, ctx_off, internal, ok, n, ws, wsb, st, cached)
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

// a coalesced batch through a resident public-key table: its launch is ONE circl_hip_mldsa_verify_table_dev call, whose one-launch route may raise
// the batch's completion flag itself (host_common.h TailOffer); nothing of a verification is secret
PipeOpts dsa_verify_table_opts() {
    PipeOpts o = dsa_opts(size_t(1) << 13, false);
    o.tail_flag_ok = true;
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
    const std::vector<HBlob> blobs = {{msg_blob, msg_off}, {ctx_blob, ctx_blob ? ctx_off : nullptr}};
    const std::vector<HOut> outs = {{ok, 1}};
    auto ws_fn = [&](size_t c) { return mldsa_ws_any(param, c) + (KM == KM_KEYED ? mldsa_table_any(param, nkeys) : 0); };
    auto launch = [&](Chunk &c) {
        return mldsa_verify_dev_any<KM>(param, c.in[0], nkeys, KM == KM_KEYED ? reinterpret_cast<const uint32_t *>(c.in[2]) : nullptr, c.in[1], c.blob[0], c.off[0],
                                        c.blob[1], c.off[1], internal, c.out[0], c.cnt, c.ws, c.ws_bytes, c.st);
    };
    if (KM == KM_ITEM && msg_blob && all_inputs_present(ins)) {  // circl_hip_set_coalesce: every item brings its own key, so calls of different callers mix freely
        const int slot = param == 44 ? 0 : param == 65 ? 1 : param == 87 ? 2 : -1;  // (round-3 Dilithium: no slot, never coalesced)
        if (Coalescer *co = call_coalescer(internal ? kCoDsaVerifyInternal : kCoDsaVerify, slot, dev)) {
            const int rc = coalesce_run(co, n, ins, blobs, outs, ws_fn, dsa_opts(size_t(1) << 13, false), launch);
            if (rc != kNotCoalesced) return rc;
        }
    }
    return run_pipeline(dev, n, ins, blobs, outs, ws_fn, dsa_opts(size_t(1) << 13, false), launch);
}

// ---- ML-DSA sign ------------------------------------------------------------------------------
constexpr int kSignBlocksPerCU = 8;
// streams of a round up to which its hash kernels run a stream per lane pair (CIRCL_HIP_SIGN_SPLIT_LOG2: tuning aid)
// ... and up to which they run two streams per wavefront on the cooperative permutation (CIRCL_HIP_SIGN_COOP_LOG2, 0 = never)
size_t sign_coop_streams() {
    static const int lg = env_int("CIRCL_HIP_SIGN_COOP_LOG2", 11, 0, 20);
    return lg <= 0 ? size_t(0) : size_t(1) << lg;
}
size_t sign_split_lanes() {
    static const size_t v = size_t(1) << env_int("CIRCL_HIP_SIGN_SPLIT_LOG2", 16, 0, 30);
    return v;
}

// below this many items the single persistent kernel signs the batch (CIRCL_HIP_SIGN_BATCHED_MIN; default 1 = never: measured, the
// round structure with wide speculation is faster at every size -- n = 1 674 -> 499 us, n = 8 971 -> 386 us; the route stays for A/B runs)
size_t sign_batched_min() {
    static const size_t v = (size_t)env_int("CIRCL_HIP_SIGN_BATCHED_MIN", 1, 1, 1 << 20);
    return v;
}

// CIRCL_HIP_SIGN_PAIR=1: the long rounds try two attempts per item with shared matrix reads (lazy pairs, sign_next_k in
// mldsa_sign_batched.h).  Measured on MI355X (profiles/r03_sign_sweep.txt): the w kernel's HBM traffic per attempt falls by 35 %
// and its time by 17 %, which the 10.8 % extra masks and products eat: 8.39e6 against 8.35e6 sig/s for ML-DSA-65 at 2^18, with
// twice the per-attempt workspace -- so the default stays one attempt per item.
inline bool sign_pair_mode() {
    static const bool v = [] {
        const char *e = getenv("CIRCL_HIP_SIGN_PAIR");
        return e && atoi(e) != 0;
    }();
    return v;
}

// The carving of a signing workspace (one definition for the size query and for the launches).
template <int MODE> struct SignLayout {
    using S = circl::mldsa::SG<MODE>;
    using B = circl::mldsa::SB<MODE>;
    size_t n, E, tail_units;
    size_t o_mr, o_work, o_scratch;                 // persistent kernel (also the tail of the batched path)
    size_t o_A, o_sec, o_y, o_w0, o_muw1, o_cb, o_attempts, o_best, o_list0, o_list1, o_ctl, o_secret_end;  // batched path
    size_t o_dead, o_long, total;
    explicit SignLayout(size_t n_) : n(n_) {
        // entries: one per item (two with lazy pairs) in the long rounds; small batches get room for up to 64 attempts per item
        // and round, capped at kMinEntryCapacity entries
        constexpr size_t min_entries = circl::mldsa::kMinEntryCapacity;
        E = std::max((sign_pair_mode() ? 2 : 1) * n, std::min(min_entries, 64 * n));
        tail_units = (size_t)max_cu_count() * kSignBlocksPerCU;
        size_t o = 0;
        auto take = [&](size_t bytes) { const size_t at = o; o += up256(bytes); return at; };
        o_mr = take(128 * n);
        o_work = take(256);
        o_scratch = take(tail_units * S::SCRATCH_BYTES);
        if (n >= sign_batched_min()) {
            o_A = take(n * B::A_BYTES);
            o_sec = take(n * B::SEC_BYTES);
            o_y = take(E * B::Y_BYTES);
            o_w0 = take(E * B::W0_BYTES);
            o_muw1 = take(E * B::MUW1_BYTES);
            o_cb = take(E * B::CB_BYTES);
            o_secret_end = o;
            o_attempts = take(4 * n);
            o_best = take(4 * n);
            o_list0 = take(4 * E);
            o_list1 = take(4 * E);
            o_ctl = take(256);
        } else {
            o_A = o_sec = o_y = o_w0 = o_muw1 = o_cb = o_secret_end = o_attempts = o_best = o_list0 = o_list1 = o_ctl = o;
        }
        o_dead = take(n);  // one byte per item: the "context refused" flags of mldsa_sign_prep_kernel
        o_long = take(kLongCtlBytes);  // the list of long messages (mldsa_kernels.h, kLongMsg)
        total = o;
    }
};
// Batches of at least kSignSplitMin items are signed as two halves side by side on two library-owned streams (forked from
// and joined to the caller's stream by events): the late rounds of a half are latency-bound chains on a mostly idle chip
// (a lone wave's five ExpandMask blocks, eight challenge permutations) and fill the gaps of the other half's.  Measured
// (tools/sign_split.py, ML-DSA-65): +9 % at 2^18, +4.5 % at 2^16; four parts are slower than one.
constexpr size_t kSignSplitMin = size_t(1) << 16;
inline size_t sign_first_half(size_t n) { return ((n / 2) + 63) & ~size_t(63); }
template <int MODE> size_t mldsa_sign_ws_bytes(size_t n) {
    const size_t whole = SignLayout<MODE>(n).total;
    if (n < kSignSplitMin) return whole;
    const size_t h0 = sign_first_half(n);
    return std::max(whole, up256(SignLayout<MODE>(h0).total) + SignLayout<MODE>(n - h0).total);
}

// Phase-split signing: rounds over the list of unsigned items (mldsa_sign_batched.h), driven by the device: the host
// enqueues a fixed schedule of rounds plus the persistent tail and reads nothing back, so the call is asynchronous.
// A private key prepared once (circl_hip_mldsa_privkey_new): its entry point parks the table here for the length of its call, and the
// round signer takes A and the transformed secrets from it instead of expanding them (thread-local: calls are per thread)
thread_local const circl_hip_keytable *tl_sign_prepared = nullptr;
// ... and, for a table of SEVERAL prepared keys, the device array that names every item's entry (nullptr: entry 0)
thread_local const uint32_t *tl_sign_key_idx = nullptr;
// The host-buffer signing entry points have checked every context on the host (check_contexts == CTX_OK) before anything reaches the
// device: no item can be "dead" (mldsa_sign_prep_kernel), so the launch that zeroes dead items' signatures is not enqueued.  Set for the
// length of the _dev call by the host path's launch callback, on the thread that makes it.
thread_local bool tl_sign_ctx_ok = false;
struct SignCtxOk {
    bool prev;
    SignCtxOk() : prev(tl_sign_ctx_ok) { tl_sign_ctx_ok = true; }
    ~SignCtxOk() { tl_sign_ctx_ok = prev; }
};

template <int MODE>
int mldsa_sign_batched_part(const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob,
                            const uint64_t *ctx_off, const uint8_t *rnd, int internal, uint8_t *sig, size_t n, void *ws, hipStream_t st,
                            bool shared) {
    using namespace circl::mldsa;
    constexpr int K = DP<MODE>::K, L = DP<MODE>::L;
    const SignLayout<MODE> lay(n);
    if (lay.E >= (size_t(1) << kEntryShift)) return CIRCL_HIP_EPARAM;
    const int cus = cu_count();
    uint8_t *base = static_cast<uint8_t *>(ws);
    SignState S;
    S.shared = shared ? 1u : 0u;
    const circl_hip_keytable *prep = shared ? tl_sign_prepared : nullptr;
    const uint32_t *key_idx = prep ? tl_sign_key_idx : nullptr;
    const KeyIdx kx{key_idx, prep ? (uint32_t)(prep->nkeys - 1) : 0u};  // a device index vector is bounded to the table on every read
    S.key_idx = KeyIdx{};
    S.mr = base + lay.o_mr;
    S.A = reinterpret_cast<uint32_t *>(base + lay.o_A);
    S.sec = reinterpret_cast<uint32_t *>(base + lay.o_sec);
    if (prep) {
        S.A = reinterpret_cast<uint32_t *>(prep->d_table);
        S.sec = reinterpret_cast<uint32_t *>(prep->d_table + up256(prep->nkeys * SB<MODE>::A_BYTES));
        S.shared = 2u;
        S.key_idx = kx;
    }
    S.y = reinterpret_cast<uint32_t *>(base + lay.o_y);
    S.w0 = reinterpret_cast<uint32_t *>(base + lay.o_w0);
    S.muw1 = base + lay.o_muw1;
    S.cb = base + lay.o_cb;
    S.attempts = reinterpret_cast<uint32_t *>(base + lay.o_attempts);
    S.best = reinterpret_cast<uint32_t *>(base + lay.o_best);
    S.list[0] = reinterpret_cast<uint32_t *>(base + lay.o_list0);
    S.list[1] = reinterpret_cast<uint32_t *>(base + lay.o_list1);
    S.count = reinterpret_cast<uint32_t *>(base + lay.o_ctl);
    S.kk = S.count + 2;
    S.chain_done = S.count + 8;
    S.capacity = (uint32_t)lay.E;
    uint8_t *dead = base + lay.o_dead;
    unsigned *tail_work = reinterpret_cast<unsigned *>(base + lay.o_work);
    uint8_t *tail_scratch = base + lay.o_scratch;
    S.tail_work = tail_work;
    // Speculative rounds: once at most spec_target entries would result, every surviving item gets
    // k = spec_target / items (<= 64) consecutive attempts per round (a round costs its five dependent launches whatever
    // the count).
    static const uint32_t spec_per_cu = [] {  // tuning aid: CIRCL_HIP_SIGN_SPEC = list entries per CU below which rounds speculate (0 = never)
        const char *e = getenv("CIRCL_HIP_SIGN_SPEC");
        const int x = e ? atoi(e) : -1;
        return (uint32_t)(x >= 0 && x <= 4096 ? x : 128);  // plateau 96 .. 256 at 2^16 items
    }();
    // ... and small batches less widely than the chip could take: a round costs ~200 us plus ~13 us per 1 024 entries, so 8 192
    // entries (or 8 attempts per item) in each of two rounds beat 32 k entries in one (measured 2^8 .. 2^12 items, tools/sign_spec_sweep.sh)
    static const bool spec_env = getenv("CIRCL_HIP_SIGN_SPEC") != nullptr;
    S.spec_target = (uint32_t)std::min<size_t>((size_t)cus * spec_per_cu, lay.E);
    if (!spec_env) S.spec_target = (uint32_t)std::min<size_t>(S.spec_target, std::max<size_t>(8192, 8 * n));
    S.pair = sign_pair_mode() ? 1u : 0u;
    const unsigned k0 = sign_next_k(n, S.spec_target, S.pair);
    constexpr int kMaxRounds = 400;
    size_t entries_upper[kMaxRounds];
    bool lazy[kMaxRounds];
    static const double eps = __builtin_ldexp(1.0, -env_int("CIRCL_HIP_SIGN_EPS_LOG2", 16, 1, 60));  // expected unsigned items behind the schedule
    const int rounds = sign_round_schedule<MODE>(n, k0, S.spec_target, S.pair, entries_upper, lazy, kMaxRounds, eps);
    const unsigned nb256 = (unsigned)((n + 255) / 256);
    // small batches: mu, rho'', the dead flags (and a prepared key's list set-up) in ONE launch, two items per wavefront on the
    // cooperative permutation (mldsa_sign_front_kernel) instead of fill + scan + mu + prep; CIRCL_HIP_SIGN_FRONT=0: the separate kernels
    static const bool one_front = env_int("CIRCL_HIP_SIGN_FRONT", 1, 0, 1) != 0;
    if (one_front && n <= kSmallMu) {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_HASH, st);
        hipLaunchKernelGGL(mldsa_sign_front_kernel<MODE>, dim3((unsigned)((n + 1) / 2)), dim3(64), 0, st, sk, (shared && !key_idx) ? size_t(0) : (size_t)KG<MODE>::SK,
                           kx, msg_blob, msg_off, ctx_blob, ctx_off, rnd, internal, S.mr, n, dead, prep ? S.attempts : nullptr, prep ? S.best : nullptr,
                           prep ? S.list[0] : nullptr, prep ? S.count : nullptr, k0);
    } else {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_HASH, st);
        LongCtl *lctl = reinterpret_cast<LongCtl *>(base + lay.o_long);
        if (int rc = mldsa_long_prepass<DP<MODE>::TR / 8>(sk + 64, (shared && !key_idx) ? 0 : KG<MODE>::SK, kx, nullptr, 0, 0, msg_blob, msg_off, ctx_blob,
                                                          ctx_off, DP<MODE>::NIST ? internal : 1, S.mr, 128, lctl, n, st))
            return rc;
        // (a prepared key: the prep kernel also sets up the round signer's lists -- sign_secrets_kernel has nothing else to do then)
        if (prep)
            hipLaunchKernelGGL(mldsa_sign_prep_kernel<MODE>, dim3(nb256), dim3(256), 0, st, sk, msg_blob, msg_off, ctx_blob, ctx_off, rnd, internal,
                               S.mr, n, shared ? 1 : 0, dead, (const LongCtl *)lctl, kx, S.attempts, S.best, S.list[0], S.count, k0);
        else
            hipLaunchKernelGGL(mldsa_sign_prep_kernel<MODE>, dim3(nb256), dim3(256), 0, st, sk, msg_blob, msg_off, ctx_blob, ctx_off, rnd, internal,
                               S.mr, n, shared ? 1 : 0, dead, (const LongCtl *)lctl, kx);
    }
    if (!prep) {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_SIGN, st);
        const size_t nkeys = shared ? 1 : n;
        hipLaunchKernelGGL(sign_expand_a_kernel<MODE>, dim3((unsigned)((nkeys * K * L + 255) / 256)), dim3(256), 0, st, sk, S, nkeys);
        hipLaunchKernelGGL(sign_secrets_kernel<MODE>, dim3((unsigned)n), dim3(64), 0, st, sk, S, n, k0);
    }
    // grids: the kernels loop over the device-side count, so any grid is correct; the schedule's upper estimate of a
    // round's entries gives (nearly) one workgroup per entry while the lists are long, and small launches for the late and
    // the empty rounds
    const size_t lane_cap = (size_t)cus * 10;
    const unsigned small = (unsigned)cus * 4;  // grid of the kernels that usually have nothing to do (grid-stride loops: any grid is correct)
    for (int round = 0; round < rounds; round++) {
        const int cur = round & 1;
        const size_t upper = std::min(round == 0 ? n * k0 : entries_upper[round], lay.E);
        // a pass of a lazy round handles every other entry; the schedule's plan picks the grid, the device-side counts decide
        const size_t pass0 = lazy[round] ? (upper + 1) / 2 : upper, pass1 = lazy[round] ? (upper + 1) / 2 : 0;
        const unsigned gw = (unsigned)std::max<size_t>(1, (upper + 1) / 2);
        const unsigned gm = (unsigned)std::max<size_t>(1, std::min((upper * L + 255) / 256, lane_cap));
        const unsigned gc = (unsigned)std::max<size_t>(1, std::min((upper + 255) / 256, lane_cap));
        auto g256 = [&](size_t work) { return (unsigned)std::max<size_t>(1, std::min((work + 255) / 256, lane_cap)); };
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_SIGN, st);
        // hash chains of short rounds on lane pairs (sign_mask_kernel): while a lane per stream leaves the SIMDs at or below one wavefront each
        const bool split_mask = upper * L <= sign_split_lanes(), split_ch = upper <= sign_split_lanes();
        const bool coop_mask = upper * L <= sign_coop_streams(), coop_ch = upper <= sign_coop_streams();
        // a SHORT round in one launch: a workgroup of K wavefronts per entry runs mask -> w -> challenge -> finish (sign_round_chain_kernel);
        // up to 2^CIRCL_HIP_SIGN_CHAIN_LOG2 entries (0: never), never a lazy round
        static const size_t chain_entries = [] { const int lg = env_int("CIRCL_HIP_SIGN_CHAIN_LOG2", 8, 0, 16); return lg <= 0 ? size_t(0) : size_t(1) << lg; }();
        if (upper <= chain_entries && !lazy[round]) {
            hipLaunchKernelGGL(sign_round_chain_kernel<MODE>, dim3((unsigned)std::max<size_t>(1, upper)), dim3(K * 64), 0, st, S, cur, sig, round == rounds - 1 ? 1 : 0);
            continue;  // (its last workgroup commits and compacts)
        }
        if (coop_mask) hipLaunchKernelGGL(sign_mask_coop_kernel<MODE>, dim3((unsigned)std::max<size_t>(1, (upper * L + 1) / 2)), dim3(64), 0, st, S, cur);
        else if (split_mask) hipLaunchKernelGGL((sign_mask_kernel<MODE, true>), dim3((unsigned)std::max<size_t>(1, std::min((upper * L + 127) / 128, lane_cap))), dim3(256), 0, st, S, cur);
        else hipLaunchKernelGGL((sign_mask_kernel<MODE, false>), dim3(gm), dim3(256), 0, st, S, cur);
        // L = 7 (ML-DSA-87, Dilithium5): the long rounds (the list is longer than the speculation threshold and pairs are off: one entry
        // per item) take the form of the w kernel without the paired path: 4 wavefronts per SIMD instead of 3.  Measured on one box,
        // alternating: ML-DSA-87 6.836 / 6.867 -> 6.873 / 6.888e6 signatures/s (+0.4 %: the kernel sits on its matrix reads either way);
        // where the general form already runs 4 wavefronts nothing moves (ML-DSA-65 8.92 / 8.88 -> 8.90 / 8.94e6) or it loses (ML-DSA-44
        // 1.343 -> 1.315e7), so only L = 7 takes it (profiles/r05_sign_ab.txt; CIRCL_HIP_SIGN_W_SINGLES=0: always the general form)
        static const bool w_singles = env_int("CIRCL_HIP_SIGN_W_SINGLES", 1, 0, 1) != 0;
        if (w_singles && L > 5 && !S.pair && upper > S.spec_target) hipLaunchKernelGGL((sign_w_kernel<MODE, false>), dim3(gw), dim3(64), 0, st, S, cur);
        else hipLaunchKernelGGL((sign_w_kernel<MODE, true>), dim3(gw), dim3(64), 0, st, S, cur);
        if (coop_ch) hipLaunchKernelGGL(sign_challenge_coop_kernel<MODE>, dim3((unsigned)std::max<size_t>(1, (pass0 + 1) / 2)), dim3(64), 0, st, S, cur, 0);
        else if (split_ch) hipLaunchKernelGGL((sign_challenge_kernel<MODE, true>), dim3(g256(2 * pass0)), dim3(256), 0, st, S, cur, 0);
        else hipLaunchKernelGGL((sign_challenge_kernel<MODE, false>), dim3(g256(pass0)), dim3(256), 0, st, S, cur, 0);
        hipLaunchKernelGGL(sign_finish_kernel<MODE>, dim3((unsigned)std::max<size_t>(1, pass0)), dim3(64), 0, st, S, cur, 0, sig);
        if (S.pair) {  // second passes exist only in lazy rounds, and a round can only be lazy with pairs on (sign_next_k): otherwise two launches less
            hipLaunchKernelGGL((sign_challenge_kernel<MODE, false>), dim3(pass1 ? g256(pass1) : 1u), dim3(256), 0, st, S, cur, 1);
            hipLaunchKernelGGL(sign_finish_kernel<MODE>, dim3(pass1 ? (unsigned)pass1 : small), dim3(64), 0, st, S, cur, 1, sig);
        }
        hipLaunchKernelGGL(sign_commit_kernel<MODE>, dim3(lazy[round] ? small : std::min<unsigned>((unsigned)std::max<size_t>(1, upper), (unsigned)cus * 32)),
                           dim3(64), 0, st, S, cur, sig);
        hipLaunchKernelGGL(sign_compact_kernel, dim3(gc), dim3(256), 0, st, S, cur, round == rounds - 1 ? 1 : 0);
    }
    {
        // whatever the schedule left unsigned (probability below 2^-40 by construction): one wavefront per item runs that
        // item's remaining rejection iterations to the end
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_SIGN, st);
        if (rounds == 0) HIP_TRY(hipMemsetAsync(tail_work, 0, 256, st));  // (otherwise the last round's compaction zeroed the ticket counter)
        const int fin = rounds & 1;
        hipLaunchKernelGGL(mldsa_sign_kernel<MODE>, dim3((unsigned)std::min<size_t>(lay.tail_units, 512)), dim3(64), SG<MODE>::LDS_TOTAL, st, sk,
                           (const uint8_t *)S.mr, sig, tail_scratch, tail_work, (const uint32_t *)S.list[fin], (const uint32_t *)S.attempts, (size_t)0, 1u,
                           (uint32_t *)nullptr, (uint8_t *)nullptr, shared ? 1 : 0, (const uint32_t *)(S.count + fin), kx);
        // (an item is dead only when it brings a context the scheme refuses: impossible without contexts, for the internal form, and
        // after the host path's own check)
        if (ctx_blob && !internal && !tl_sign_ctx_ok)
            hipLaunchKernelGGL(mldsa_sign_zero_dead_kernel<MODE>, dim3(nb256), dim3(256), 0, st, sig, (const uint8_t *)dead, n);
    }
    // The workspace held rho'', the NTT-domain secrets, the accepted attempts' y next to c~ (z - y = c s1) and parked
    // signatures: nothing key-equivalent stays behind in the caller's workspace (the matrix rows are public).
    // (mu / rho'', the ticket counter and the tail's scratch slices lie back to back: one fill; the tail kernel's workgroup b works in
    // slice b and only workgroups below the number of unsigned items -- at most n -- do anything)
    const size_t tail_used = std::min<size_t>(n, std::min<size_t>(lay.tail_units, 512));
    const size_t wipe_a = (size_t)(tail_scratch - S.mr) + tail_used * SG<MODE>::SCRATCH_BYTES, wipe_b = lay.o_secret_end - lay.o_sec;
    if (wipe_a + wipe_b <= (size_t(8) << 20) && wipe_a % 16 == 0 && wipe_b % 16 == 0) {  // a small call: both ranges in one launch
        hipLaunchKernelGGL(sign_wipe2_kernel, dim3((unsigned)std::min<size_t>((wipe_a + wipe_b) / 16 / 256 + 1, (size_t)cus * 8)), dim3(256), 0, st,
                           reinterpret_cast<uint4 *>(S.mr), wipe_a / 16, reinterpret_cast<uint4 *>(base + lay.o_sec), wipe_b / 16);
    } else {
        HIP_TRY(hipMemsetAsync(S.mr, 0, wipe_a, st));
        HIP_TRY(hipMemsetAsync(base + lay.o_sec, 0, wipe_b, st));  // (the workspace's; a prepared key's table stays)
    }
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

template <int MODE>
int mldsa_sign_batched(const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob,
                       const uint64_t *ctx_off, const uint8_t *rnd, int internal, uint8_t *sig, size_t n, void *ws, hipStream_t st,
                       bool shared) {
    static const bool no_split = getenv("CIRCL_HIP_SIGN_NOSPLIT") != nullptr;  // tuning aid
    if (n < kSignSplitMin || no_split)
        return mldsa_sign_batched_part<MODE>(sk, msg_blob, msg_off, ctx_blob, ctx_off, rnd, internal, sig, n, ws, st, shared);
    constexpr size_t SK = circl::mldsa::KG<MODE>::SK, SIG = circl::mldsa::DG<MODE>::SIG;
    hipStream_t aux[2];
    if (int rc = aux_streams(current_device(), aux)) return rc;
    hipEvent_t fork = nullptr, done[2] = {nullptr, nullptr};
    struct Events {  // destroyed on every path; destroying a recorded event releases it once it has completed
        hipEvent_t &a, &b, &c;
        ~Events() {
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
            if (c) (void)hipEventDestroy(c);
        }
    } guard{fork, done[0], done[1]};
    HIP_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&done[0], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&done[1], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(fork, st));  // the halves start after whatever the caller queued before the call
    const size_t h0 = sign_first_half(n);
    const size_t lo[2] = {0, h0}, cnt[2] = {h0, n - h0};
    uint8_t *wsp[2] = {static_cast<uint8_t *>(ws), static_cast<uint8_t *>(ws) + up256(SignLayout<MODE>(h0).total)};
    int rc = CIRCL_HIP_OK;
    auto note = [&](hipError_t e, const char *what) {  // remember the first failure, keep going: both halves must be joined
        if (e == hipSuccess) return true;
        if (rc == CIRCL_HIP_OK) {
            g_err = std::string("mldsa_sign_batched: ") + what + " -> " + hipGetErrorString(e);
            rc = e == hipErrorOutOfMemory ? CIRCL_HIP_ENOMEM : CIRCL_HIP_EHIP;
        }
        (void)hipGetLastError();
        return false;
    };
    for (int p = 0; p < 2; p++) {
        if (note(hipStreamWaitEvent(aux[p], fork, 0), "fork")) {
            const uint32_t *kidx = tl_sign_key_idx;  // (a table of several prepared keys: this half's slice of the index array)
            tl_sign_key_idx = kidx ? kidx + lo[p] : nullptr;
            const int r = mldsa_sign_batched_part<MODE>(shared ? sk : sk + lo[p] * SK, msg_blob, msg_off + lo[p], ctx_blob, ctx_off ? ctx_off + lo[p] : nullptr,
                                                        rnd + lo[p] * 32, internal, sig + lo[p] * SIG, cnt[p], wsp[p], aux[p], shared);
            tl_sign_key_idx = kidx;
            if (r != CIRCL_HIP_OK && rc == CIRCL_HIP_OK) rc = r;
        }
        // join on EVERY path: whatever was enqueued on the library's streams is ordered before the caller's later work (the
        // caller may reuse sig / the workspace as soon as its own stream gets there); if the event calls themselves fail,
        // wait for the library's stream on the host instead
        if (!note(hipEventRecord(done[p], aux[p]), "join record") || !note(hipStreamWaitEvent(st, done[p], 0), "join wait"))
            (void)hipStreamSynchronize(aux[p]);
    }
    return rc;
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
    if (n >= sign_batched_min()) return mldsa_sign_batched<MODE>(sk, msg_blob, msg_off, ctx_blob, ctx_off, rnd, internal, sig, n, ws, st, shared);
    const SignLayout<MODE> lay(n);
    const uint32_t *key_idx = (shared && tl_sign_prepared) ? tl_sign_key_idx : nullptr;
    const KeyIdx kx{key_idx, key_idx ? (uint32_t)(tl_sign_prepared->nkeys - 1) : 0u};
    uint8_t *base = static_cast<uint8_t *>(ws);
    uint8_t *mr = base + lay.o_mr, *dead = base + lay.o_dead, *scratch = base + lay.o_scratch;
    unsigned *work = reinterpret_cast<unsigned *>(base + lay.o_work);
    HIP_TRY(hipMemsetAsync(work, 0, 256, st));
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_HASH, st);
        LongCtl *lctl = reinterpret_cast<LongCtl *>(base + lay.o_long);
        if (int rc = mldsa_long_prepass<DP<MODE>::TR / 8>(sk + 64, (shared && !key_idx) ? 0 : KG<MODE>::SK, kx, nullptr, 0, 0, msg_blob, msg_off, ctx_blob,
                                                          ctx_off, DP<MODE>::NIST ? internal : 1, mr, 128, lctl, n, st))
            return rc;
        hipLaunchKernelGGL(mldsa_sign_prep_kernel<MODE>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sk, msg_blob,
                           msg_off, ctx_blob, ctx_off, rnd, internal, mr, n, shared ? 1 : 0, dead, (const LongCtl *)lctl, kx);
    }
    {
        auto kern = mldsa_sign_kernel<MODE>;
        unsigned resident = resident_blocks(kern, S::LDS_TOTAL);
        resident = std::min<unsigned>(resident, (unsigned)(cu_count() * kSignBlocksPerCU));
        const unsigned blocks = (unsigned)std::min<size_t>(n, resident);
        ProfScope ps(CIRCL_HIP_KERNEL_MLDSA_SIGN, st);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), S::LDS_TOTAL, st, sk, (const uint8_t *)mr, sig, scratch, work,
                           (const uint32_t *)nullptr, (const uint32_t *)nullptr, n, 1u, (uint32_t *)nullptr, (uint8_t *)nullptr, shared ? 1 : 0,
                           (const uint32_t *)nullptr, kx);
        if (ctx_blob && !internal && !tl_sign_ctx_ok)
            hipLaunchKernelGGL(mldsa_sign_zero_dead_kernel<MODE>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sig, (const uint8_t *)dead, n);
    }
    HIP_TRY(hipMemsetAsync(mr, 0, up256(128 * n), st));  // rho'' and the NTT-domain secrets do not stay behind
    HIP_TRY(hipMemsetAsync(scratch, 0, std::min<size_t>(n, lay.tail_units) * S::SCRATCH_BYTES, st));
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

// the table of nkeys prepared private keys: [A: nkeys x K L packed rows][s1-hat, s2-hat, t0-hat: nkeys x (L + 2 K) packed rows][set-up
// scratch], made by the round signer's own set-up kernels (every key is an "item" of an unshared batch of nkeys)
template <int MODE> int mldsa_privkey_build(circl_hip_keytable *t, const uint8_t *sk, hipStream_t st) {
    using namespace circl::mldsa;
    using B = SB<MODE>;
    constexpr int K = DP<MODE>::K, L = DP<MODE>::L;
    const size_t nk = t->nkeys;
    const size_t o_sec = up256(nk * B::A_BYTES), o_scr = o_sec + up256(nk * B::SEC_BYTES);
    t->table_bytes = o_scr + up256(12 * nk) + 1024;
    if (hipMalloc(reinterpret_cast<void **>(&t->d_keys), t->keys_bytes) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&t->d_table), t->table_bytes) != hipSuccess) {
        (void)hipGetLastError();
        return CIRCL_HIP_ENOMEM;
    }
    if (int rc = upload_secret(t->d_keys, sk, KG<MODE>::SK * nk, st)) return rc;  // (through wiped page-locked staging)
    uint32_t *scratch = reinterpret_cast<uint32_t *>(t->d_table + o_scr);
    SignState S{};
    S.shared = 0u;
    S.A = reinterpret_cast<uint32_t *>(t->d_table);
    S.sec = reinterpret_cast<uint32_t *>(t->d_table + o_sec);
    S.count = scratch; S.kk = scratch + 2; S.chain_done = scratch + 8; S.attempts = scratch + 64; S.best = scratch + 64 + nk; S.list[0] = scratch + 64 + 2 * nk; S.list[1] = S.list[0];
    hipLaunchKernelGGL(sign_expand_a_kernel<MODE>, dim3((unsigned)((nk * K * L + 255) / 256)), dim3(256), 0, st, (const uint8_t *)t->d_keys, S, nk);
    hipLaunchKernelGGL(sign_secrets_kernel<MODE>, dim3((unsigned)nk), dim3(64), 0, st, (const uint8_t *)t->d_keys, S, nk, 1u);
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

int mldsa_sign_host(int param, const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob,
                    const uint64_t *ctx_off, const uint8_t *rnd, int internal, uint8_t *sig, size_t n, int device, bool shared = false) {
    const size_t SK = circl_hip_mldsa_sk_size(param), SIG = circl_hip_mldsa_sig_size(param);
    if (!SK) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    if (!internal && check_contexts(param, ctx_blob, ctx_off, n) != CTX_OK) return CIRCL_HIP_EPARAM;  // sign.ErrContextTooLong / ErrContextNotSupported
    const PipeOpts opts = dsa_opts(size_t(1) << 13, true, /*depth=*/3);  // (each chunk's workspace is ~60 KB per item)
    std::vector<uint8_t> zeros;
    if (!rnd) zeros.assign(32 * std::min(n, opts.chunk_items), 0);  // deterministic signing: 32 zero bytes per item
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        std::vector<HIn> ins;
        ins.push_back(shared ? HIn{sk, SK, true, true} : HIn{sk + lo * SK, SK, true});
        ins.push_back(rnd ? HIn{rnd + lo * 32, 32, true} : HIn{zeros.data(), zeros.size(), false, true});
        return run_pipeline(dev, cnt, ins, {{msg_blob, msg_off + lo}, {ctx_blob, ctx_blob ? ctx_off + lo : nullptr}}, {{sig + lo * SIG, SIG}},
                            [&](size_t c) { return mldsa_sign_ws_any(param, c); }, opts, [&](Chunk &c) {
                                SignCtxOk checked;  // (check_contexts above)
                                return mldsa_sign_dev_any(param, c.in[0], c.blob[0], c.off[0], c.blob[1], c.off[1], c.in[1], internal, c.out[0], c.cnt, c.ws,
                                                          c.ws_bytes, c.st, shared);
                            });
    }, kHeavyOneDeviceMax);
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

// ---- a public-key table that lives across calls (keytable.h): A and tr of every entry, once ---------------------------------------
static int mldsa_keytable_new_one(int param, const uint8_t *pks, size_t nkeys, int device, circl_hip_keytable **out) {
    const size_t PK = circl_hip_mldsa_pk_size(param);
    HIP_TRY(hipSetDevice(physical_device(device)));
    circl_hip_keytable *t = new (std::nothrow) circl_hip_keytable();
    if (!t) return CIRCL_HIP_ENOMEM;
    t->magic = kKeytableMagic; t->family = 2; t->param = param; t->device = device; t->private_keys = 0; t->nkeys = nkeys; t->row = PK;
    t->keys_bytes = up256(PK * nkeys + 16);
    t->table_bytes = mldsa_table_any(param, nkeys);
    hipStream_t h2d = nullptr, d2h = nullptr, st = nullptr;
    int rc = pipeline_streams(device, &h2d, &d2h, &st);
    if (rc == CIRCL_HIP_OK && (hipMalloc(reinterpret_cast<void **>(&t->d_keys), t->keys_bytes) != hipSuccess ||
                               hipMalloc(reinterpret_cast<void **>(&t->d_table), t->table_bytes) != hipSuccess)) {
        (void)hipGetLastError();
        rc = CIRCL_HIP_ENOMEM;
    }
    if (rc == CIRCL_HIP_OK && hipMemcpyAsync(t->d_keys, pks, PK * nkeys, hipMemcpyHostToDevice, st) != hipSuccess) rc = CIRCL_HIP_EHIP;
    if (rc == CIRCL_HIP_OK) {
#define CALL(M) mldsa_table_build<M>(t, st)
        rc = [&]() -> int {
            DSA_SWITCH(param, CALL)
            return CIRCL_HIP_EPARAM;
        }();
#undef CALL
    }
    if (rc == CIRCL_HIP_OK && hipStreamSynchronize(st) != hipSuccess) rc = CIRCL_HIP_EHIP;
    if (rc != CIRCL_HIP_OK) {
        (void)hipGetLastError();
        circl_hip_keytable_free(t);
        return rc;
    }
    *out = t;
    return CIRCL_HIP_OK;
}
int circl_hip_mldsa_keytable_new(int param, const uint8_t *pks, size_t nkeys, int device, circl_hip_keytable **out) {
    if (out) *out = nullptr;
    if (!circl_hip_mldsa_pk_size(param) || !pks || !out || nkeys == 0 || nkeys > 0xffffffffull) return CIRCL_HIP_EPARAM;
    return keytable_replicate(device, [&](int dev, circl_hip_keytable **one) { return mldsa_keytable_new_one(param, pks, nkeys, dev, one); }, out);
}
int circl_hip_mldsa_verify_table_dev(const circl_hip_keytable *t, const uint32_t *d_key_idx, const uint8_t *d_sig, const uint8_t *d_msg_blob,
                                     const uint64_t *d_msg_off, const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off, uint8_t *d_ok, size_t n, void *d_ws,
                                     size_t ws_bytes, void *stream) {
    t = keytable_here(t);
    if (!t || t->family != 2 || t->private_keys) return CIRCL_HIP_EPARAM;
    return mldsa_verify_dev_any<KM_KEYED>(t->param, t->d_keys, t->nkeys, d_key_idx, d_sig, d_msg_blob, d_msg_off, d_ctx_blob, d_ctx_off, 0, d_ok, n, d_ws,
                                          ws_bytes, static_cast<hipStream_t>(stream), t);
}
int circl_hip_mldsa_verify_table(const circl_hip_keytable *t, const uint32_t *key_idx, const uint8_t *sig, const uint8_t *msg_blob, const uint64_t *msg_off,
                                 const uint8_t *ctx_blob, const uint64_t *ctx_off, uint8_t *ok, size_t n) {
    if (!t || t->magic != kKeytableMagic || t->family != 2 || t->private_keys) return CIRCL_HIP_EPARAM;
    const int param = t->param;
    const size_t SIG = circl_hip_mldsa_sig_size(param);
    if (n == 0) return CIRCL_HIP_OK;
    if (key_idx)
        for (size_t i = 0; i < n; i++)
            if (key_idx[i] >= t->nkeys) return CIRCL_HIP_EPARAM;
    if (check_contexts(param, ctx_blob, ctx_off, n) == CTX_UNSUPPORTED) return CIRCL_HIP_EPARAM;
    return table_shard(t, n, [&](const circl_hip_keytable *r, size_t lo, size_t cnt) {
        const uint32_t *ki = key_idx ? key_idx + lo : nullptr;
        Coalescer *co = usable_coalescer(r);
        if (co && cnt <= coalescer_call_max(co)) {  // a small call joins the table's cross-caller batch (absent key_idx: zeros; absent contexts: empty rows)
            const int rc = coalesce_run(co, cnt, {{sig ? sig + lo * SIG : nullptr, SIG}, {reinterpret_cast<const uint8_t *>(ki), size_t(4), false, false, true}},
                                        {{msg_blob, msg_off + lo}, {ctx_blob, ctx_blob ? ctx_off + lo : nullptr}}, {{ok + lo, 1}},
                                        [&](size_t c) { return mldsa_ws_any(param, c); }, dsa_verify_table_opts(), [&](Chunk &c) {
                                            return circl_hip_mldsa_verify_table_dev(r, reinterpret_cast<const uint32_t *>(c.in[1]), c.in[0], c.blob[0], c.off[0], c.blob[1],
                                                                                    c.off[1], c.out[0], c.cnt, c.ws, c.ws_bytes, c.st);
                                        });
            if (rc != kNotCoalesced) return rc;
        }
        return run_pipeline(r->device, cnt, {{sig ? sig + lo * SIG : nullptr, SIG}, {reinterpret_cast<const uint8_t *>(ki), ki ? size_t(4) : size_t(0)}},
                            {{msg_blob, msg_off + lo}, {ctx_blob, ctx_blob ? ctx_off + lo : nullptr}}, {{ok + lo, 1}}, [&](size_t c) { return mldsa_ws_any(param, c); },
                            dsa_opts(size_t(1) << 13, false), [&](Chunk &c) {
                                return circl_hip_mldsa_verify_table_dev(r, ki ? reinterpret_cast<const uint32_t *>(c.in[1]) : nullptr, c.in[0], c.blob[0], c.off[0],
                                                                        c.blob[1], c.off[1], c.out[0], c.cnt, c.ws, c.ws_bytes, c.st);
                            });
    });
}

// the asynchronous form of circl_hip_mldsa_verify_table (include/circl_hip.h: circl_hip_keytable_async_start)
int circl_hip_mldsa_verify_table_submit(const circl_hip_keytable *t, const uint32_t *key_idx, const uint8_t *sig, const uint8_t *msg_blob, const uint64_t *msg_off,
                                        const uint8_t *ctx_blob, const uint64_t *ctx_off, uint8_t *ok, size_t n, uint64_t *ticket) {
    if (ticket) *ticket = 0;
    if (!t || t->magic != kKeytableMagic || t->family != 2 || t->private_keys || !ticket) return CIRCL_HIP_EPARAM;
    if (n && (!sig || !ok || !msg_off || (ctx_blob && !ctx_off))) return CIRCL_HIP_EPARAM;
    const int param = t->param;
    const size_t SIG = circl_hip_mldsa_sig_size(param);
    if (n == 0) return CIRCL_HIP_OK;
    if (key_idx)
        for (size_t i = 0; i < n; i++)
            if (key_idx[i] >= t->nkeys) return CIRCL_HIP_EPARAM;
    if (check_contexts(param, ctx_blob, ctx_off, n) == CTX_UNSUPPORTED) return CIRCL_HIP_EPARAM;
    return table_submit(t, ticket, [&](const circl_hip_keytable *, Coalescer *co, uint64_t *seq) {
        return coalesce_submit(co, n, {{sig, SIG}, {reinterpret_cast<const uint8_t *>(key_idx), size_t(4), false, false, true}},
                               {{msg_blob, msg_off}, {ctx_blob, ctx_blob ? ctx_off : nullptr}}, {{ok, 1}}, seq, false);
    });
}
}  // extern "C"
namespace circl {
namespace host {
int dsa_table_async_start(const circl_hip_keytable *r, Coalescer *co, bool want_eventfd) {
    const int param = r->param;
    const size_t SIG = circl_hip_mldsa_sig_size(param);
    return coalescer_async_start(co, {{nullptr, SIG}, {nullptr, size_t(4), false, false, true}}, {{nullptr, nullptr}, {nullptr, nullptr}}, {{nullptr, 1}},
                                 [param](size_t c) { return mldsa_ws_any(param, c); }, dsa_verify_table_opts(), [r](Chunk &c) {
                                     return circl_hip_mldsa_verify_table_dev(r, reinterpret_cast<const uint32_t *>(c.in[1]), c.in[0], c.blob[0], c.off[0], c.blob[1], c.off[1],
                                                                             c.out[0], c.cnt, c.ws, c.ws_bytes, c.st);
                                 }, want_eventfd);
}
}  // namespace host
}  // namespace circl
extern "C" {

// ---- private keys prepared once: A and the NTT-domain secrets of the reference's parsed PrivateKey (internal/dilithium.go:149-179) ----
static int mldsa_privkeys_new_one(int param, const uint8_t *sks, size_t nkeys, int device, circl_hip_keytable **out) {
    const size_t SK = circl_hip_mldsa_sk_size(param);
    HIP_TRY(hipSetDevice(physical_device(device)));
    circl_hip_keytable *t = new (std::nothrow) circl_hip_keytable();
    if (!t) return CIRCL_HIP_ENOMEM;
    t->magic = kKeytableMagic; t->family = 2; t->param = param; t->device = device; t->private_keys = 1; t->nkeys = nkeys; t->row = SK;
    t->keys_bytes = up256(SK * nkeys + 16);
    hipStream_t h2d = nullptr, d2h = nullptr, st = nullptr;
    int rc = pipeline_streams(device, &h2d, &d2h, &st);
#define CALL(M) mldsa_privkey_build<M>(t, sks, st)
    if (rc == CIRCL_HIP_OK)
        rc = [&]() -> int {
            DSA_SWITCH(param, CALL)
            return CIRCL_HIP_EPARAM;
        }();
#undef CALL
    if (rc == CIRCL_HIP_OK && hipStreamSynchronize(st) != hipSuccess) rc = CIRCL_HIP_EHIP;
    if (rc != CIRCL_HIP_OK) {
        (void)hipGetLastError();
        circl_hip_keytable_free(t);
        return rc;
    }
    *out = t;
    return CIRCL_HIP_OK;
}
int circl_hip_mldsa_privkeys_new(int param, const uint8_t *sks, size_t nkeys, int device, circl_hip_keytable **out) {
    if (out) *out = nullptr;
    if (!circl_hip_mldsa_sk_size(param) || !sks || !out || nkeys == 0 || nkeys >= (size_t(1) << 26)) return CIRCL_HIP_EPARAM;
    return keytable_replicate(device, [&](int dev, circl_hip_keytable **one) { return mldsa_privkeys_new_one(param, sks, nkeys, dev, one); }, out);
}
int circl_hip_mldsa_privkey_new(int param, const uint8_t *sk, int device, circl_hip_keytable **out) {
    return circl_hip_mldsa_privkeys_new(param, sk, 1, device, out);
}
int circl_hip_mldsa_sign_table_keyed_dev(const circl_hip_keytable *t, const uint32_t *d_key_idx, const uint8_t *d_msg_blob, const uint64_t *d_msg_off,
                                         const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off, const uint8_t *d_rnd, int internal, uint8_t *d_sig, size_t n,
                                         void *d_ws, size_t ws_bytes, void *stream) {
    t = keytable_here(t);
    if (!t || t->family != 2 || !t->private_keys || (reinterpret_cast<uintptr_t>(d_key_idx) & 3)) return CIRCL_HIP_EPARAM;
    struct Park {  // (restored on every path)
        Park(const circl_hip_keytable *p, const uint32_t *k) { tl_sign_prepared = p; tl_sign_key_idx = k; }
        ~Park() { tl_sign_prepared = nullptr; tl_sign_key_idx = nullptr; }
    } park(t, d_key_idx);
    return mldsa_sign_dev_any(t->param, t->d_keys, d_msg_blob, d_msg_off, d_ctx_blob, d_ctx_off, d_rnd, internal, d_sig, n, d_ws, ws_bytes,
                              static_cast<hipStream_t>(stream), true);
}
int circl_hip_mldsa_sign_table_dev(const circl_hip_keytable *t, const uint8_t *d_msg_blob, const uint64_t *d_msg_off, const uint8_t *d_ctx_blob,
                                   const uint64_t *d_ctx_off, const uint8_t *d_rnd, int internal, uint8_t *d_sig, size_t n, void *d_ws, size_t ws_bytes,
                                   void *stream) {
    return circl_hip_mldsa_sign_table_keyed_dev(t, nullptr, d_msg_blob, d_msg_off, d_ctx_blob, d_ctx_off, d_rnd, internal, d_sig, n, d_ws, ws_bytes, stream);
}
int circl_hip_mldsa_sign_table_keyed(const circl_hip_keytable *t, const uint32_t *key_idx, const uint8_t *msg_blob, const uint64_t *msg_off,
                                     const uint8_t *ctx_blob, const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sig, size_t n) {
    if (!t || t->magic != kKeytableMagic || t->family != 2 || !t->private_keys) return CIRCL_HIP_EPARAM;
    const int param = t->param;
    const size_t SIG = circl_hip_mldsa_sig_size(param);
    if (n == 0) return CIRCL_HIP_OK;
    if (key_idx)
        for (size_t i = 0; i < n; i++)
            if (key_idx[i] >= t->nkeys) return CIRCL_HIP_EPARAM;
    if (check_contexts(param, ctx_blob, ctx_off, n) != CTX_OK) return CIRCL_HIP_EPARAM;  // sign.ErrContextTooLong / ErrContextNotSupported
    const PipeOpts opts = dsa_opts(size_t(1) << 13, true, /*depth=*/3);
    std::vector<uint8_t> zeros;
    if (!rnd) zeros.assign(32 * std::min(n, opts.chunk_items), 0);  // deterministic signing: 32 zero bytes per item
    return table_shard(t, n, [&](const circl_hip_keytable *r, size_t lo, size_t cnt) {
        const uint32_t *ki = key_idx ? key_idx + lo : nullptr;
        Coalescer *co = usable_coalescer(r);
        if (co && cnt <= coalescer_call_max(co)) {  // a small call joins the table's cross-caller batch (absent rnd / key_idx: zeros)
            const int rc = coalesce_run(co, cnt, {{rnd ? rnd + lo * 32 : nullptr, size_t(32), true, false, true}, {reinterpret_cast<const uint8_t *>(ki), size_t(4), false, false, true}},
                                        {{msg_blob, msg_off + lo}, {ctx_blob, ctx_blob ? ctx_off + lo : nullptr}}, {{sig + lo * SIG, SIG}},
                                        [&](size_t c) { return mldsa_sign_ws_any(param, c); }, opts, [&](Chunk &c) {
                                            SignCtxOk checked;  // (every caller of the batch passed check_contexts)
                                            return circl_hip_mldsa_sign_table_keyed_dev(r, reinterpret_cast<const uint32_t *>(c.in[1]), c.blob[0], c.off[0], c.blob[1],
                                                                                        c.off[1], c.in[0], 0, c.out[0], c.cnt, c.ws, c.ws_bytes, c.st);
                                        });
            if (rc != kNotCoalesced) return rc;
        }
        std::vector<HIn> ins;
        ins.push_back(rnd ? HIn{rnd + lo * 32, 32, true} : HIn{zeros.data(), zeros.size(), false, true});
        ins.push_back(HIn{reinterpret_cast<const uint8_t *>(ki), ki ? size_t(4) : size_t(0)});
        return run_pipeline(r->device, cnt, ins, {{msg_blob, msg_off + lo}, {ctx_blob, ctx_blob ? ctx_off + lo : nullptr}}, {{sig + lo * SIG, SIG}},
                            [&](size_t c) { return mldsa_sign_ws_any(param, c); }, opts, [&](Chunk &c) {
                                SignCtxOk checked;  // (check_contexts above)
                                return circl_hip_mldsa_sign_table_keyed_dev(r, ki ? reinterpret_cast<const uint32_t *>(c.in[1]) : nullptr, c.blob[0], c.off[0], c.blob[1],
                                                                            c.off[1], c.in[0], 0, c.out[0], c.cnt, c.ws, c.ws_bytes, c.st);
                            });
    }, kHeavyOneDeviceMax);
}
int circl_hip_mldsa_sign_table(const circl_hip_keytable *t, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off,
                               const uint8_t *rnd, uint8_t *sig, size_t n) {
    return circl_hip_mldsa_sign_table_keyed(t, nullptr, msg_blob, msg_off, ctx_blob, ctx_off, rnd, sig, n);
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

// PolyDeriveUniformBall (sign/mldsa/mldsa65/internal/sample.go:299-339) of n challenge seeds: a unit-level primitive for
// the parity tests, like circl_hip_kyber_ntt.  sequential != 0 takes the reference-order scan that is otherwise only the
// fallback of the block-parallel form.
int circl_hip_mldsa_sample_in_ball(int param, const uint8_t *ctilde, uint32_t *polys, size_t n, int sequential, int device) {
    using namespace circl::mldsa;
    const size_t CT = param == 44 ? 32 : param == 65 ? 48 : param == 87 ? 64 : (param == 2 || param == 3 || param == 5) ? 32 : 0;
    if (!CT) return CIRCL_HIP_EPARAM;
    uint8_t *po = reinterpret_cast<uint8_t *>(polys);
    PipeOpts o;
    o.chunk_items = host_chunk_items(size_t(1) << 14);
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{ctilde + lo * CT, CT}}, {}, {{po + lo * 1024, 1024}}, [](size_t) { return size_t(0); }, o, [&](Chunk &c) {
#define CALL(M) hipLaunchKernelGGL(mldsa_sample_in_ball_kernel<M>, dim3((unsigned)c.cnt), dim3(64), 0, c.st, (const uint8_t *)c.in[0], \
                                   reinterpret_cast<uint32_t *>(c.out[0]), sequential)
            int rc = CIRCL_HIP_OK;
            switch (param) {
            case 44: CALL(44); break;
            case 65: CALL(65); break;
            case 87: CALL(87); break;
            case 2: CALL(2); break;
            case 3: CALL(3); break;
            case 5: CALL(5); break;
            default: rc = CIRCL_HIP_EPARAM;
            }
#undef CALL
            HIP_TRY(hipGetLastError());
            return rc;
        });
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

int circl_hip_mldsa_public_from_private_dev(int param, const uint8_t *d_sk, uint8_t *d_pk, size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    if (ndev() <= 0) return CIRCL_HIP_ENODEV;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define CALL(M) mldsa_public_dev_impl<M>(d_sk, d_pk, n, d_ws, ws_bytes, st)
    DSA_SWITCH(param, CALL)
#undef CALL
    return CIRCL_HIP_EPARAM;
}
int circl_hip_mldsa_public_from_private(int param, const uint8_t *sk, uint8_t *pk, size_t n, int device) {
    const size_t PK = circl_hip_mldsa_pk_size(param), SK = circl_hip_mldsa_sk_size(param);
    if (!PK || (n && (!sk || !pk))) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{sk + lo * SK, SK, true}}, {}, {{pk + lo * PK, PK}}, [&](size_t c) { return mldsa_ws_any(param, c); },
                            dsa_opts(size_t(1) << 13, true),
                            [&](Chunk &c) { return circl_hip_mldsa_public_from_private_dev(param, c.in[0], c.out[0], c.cnt, c.ws, c.ws_bytes, c.st); });
    });
}
int circl_hip_mldsa_keygen(int param, const uint8_t *seed32, uint8_t *pk, uint8_t *sk, size_t n, int device) {
    const size_t PK = circl_hip_mldsa_pk_size(param), SK = circl_hip_mldsa_sk_size(param);
    if (!PK) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{seed32 + lo * 32, 32, true}}, {}, {{pk + lo * PK, PK}, {sk + lo * SK, SK, true}},
                            [&](size_t c) { return mldsa_ws_any(param, c); }, dsa_opts(size_t(1) << 14, true),
                            [&](Chunk &c) { return circl_hip_mldsa_keygen_dev(param, c.in[0], c.out[0], c.out[1], c.cnt, c.ws, c.ws_bytes, c.st); });
    }, kHeavyOneDeviceMax);
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
