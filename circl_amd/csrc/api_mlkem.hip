// api_mlkem.hip -- ML-KEM (FIPS 203) and round-3 Kyber entry points of the C ABI (include/circl_hip.h).
//
// There is deliberately no CPU path in this file: every compute entry point launches the HIP kernels of
// mlkem_kernels.h or fails with CIRCL_HIP_ENODEV.
#include "host_common.h"
#include "keytable.h"
#include "mlkem_kernels.h"

using namespace circl::host;
using circl::KeyIdx;

namespace {

// ML-KEM workspace: 129 B per item + one 32 KB scratch slice per resident workgroup
constexpr size_t kKemWsPerItem = 129;  // four 32-byte slots + one status byte (round-3 decapsulation has no caller-side status)
size_t kem_scratch_bytes() { return 256 + max_resident_blocks() * 64 * 512; }
// Small batches (encapsulation): hashing and matrix expansion run side by side in one launch and the encrypt kernel takes its
// rows from a cache behind the scratch slices (mlkem_small_pre_kernel).  CIRCL_HIP_KEM_SMALL = log2 of the largest such batch
// (0 = never); the cache is sized for the largest parameter set (K = 4: 8 KB per item, whole groups).
size_t kem_small_batch() {
    static const size_t v = [] {
        const char *e = getenv("CIRCL_HIP_KEM_SMALL");
        const int lg = e ? atoi(e) : 15;
        return lg <= 0 ? size_t(0) : size_t(1) << std::min(lg, 20);
    }();
    return v;
}
size_t kem_coop_batch() {
    static const size_t v = [] {
        const char *e = getenv("CIRCL_HIP_KEM_COOP");  // log2 of the largest batch hashed two items per wavefront (0 = never)
        const int lg = e ? atoi(e) : 11;
        return lg <= 0 ? size_t(0) : size_t(1) << std::min(lg, 20);
    }();
    return v;
}
// How the hashing wavefronts of a small batch hold their sponges: 1 = two per wavefront on the cooperative permutation (shortest
// chain, LDS-bound beyond a few thousand), 2 = one per lane pair (keccak_f1600_split), 0 = one per lane (CIRCL_HIP_KEM_SPLIT=0).
int kem_hash_form(size_t n, size_t coop_max) {
    static const bool split = env_int("CIRCL_HIP_KEM_SPLIT", 1, 0, 1) != 0;  // tuning aid
    return n <= coop_max ? 1 : split ? 2 : 0;
}
unsigned kem_hash_blocks(size_t n, int form) { return (unsigned)(form == 1 ? (n + 1) / 2 : form == 2 ? (n + 31) / 32 : (n + 63) / 64); }
// One-key batches keep ONE entry in that cache (G copies of it), so their small-batch routes reach further (measured: the
// encapsulation wins up to 2^17 items, the decapsulation -- a workgroup per item for K-PKE.Decrypt -- up to 2^15):
// CIRCL_HIP_KEM_SMALL_SHARED / CIRCL_HIP_KEM_SMALL_SHARED_DECAPS = log2 of the largest such batch.
size_t kem_small_shared_batch(bool decaps) {
    static const size_t enc = size_t(1) << env_int("CIRCL_HIP_KEM_SMALL_SHARED", 17, 0, 24);
    static const size_t dec = size_t(1) << env_int("CIRCL_HIP_KEM_SMALL_SHARED_DECAPS", 15, 0, 24);
    return decaps ? dec : enc;  // (log2 0 = batches of one only)
}
// The row cache behind the scratch slices holds min(n, kem_small_batch()) entries (at least one: the one-key routes), so the
// workspace size is MONOTONE in n: a workspace sized once for the largest batch serves every smaller one.
size_t kem_cache_bytes(size_t entries) { return up256((std::max<size_t>(entries, 1) + 15) / 16 * 16 * size_t(16 * 512)); }
// Resident private keys: batches up to 2^CIRCL_HIP_KEM_CHAIN items (0 = never) decapsulate in one launch (mlkem_decaps_chain_kernel)
// ... and encapsulate in one launch up to 2^CIRCL_HIP_KEM_CHAIN_ENCAPS items (mlkem_encaps_chain_kernel)
size_t kem_chain_batch(bool decaps = true) {
    static const int lg_d = env_int("CIRCL_HIP_KEM_CHAIN", 11, 0, 20), lg_e = env_int("CIRCL_HIP_KEM_CHAIN_ENCAPS", 10, 0, 20);
    const int lg = decaps ? lg_d : lg_e;
    return lg <= 0 ? size_t(0) : size_t(1) << lg;
}
// Keys that are NOT resident: batches up to 2^CIRCL_HIP_KEM_CHAIN_ITEM items run the same one-launch form with the key work inside
// (mlkem_*_chain_kernel<K, false>: two / four wavefronts per item)
size_t kem_chain_item_batch() {
    static const int lg = env_int("CIRCL_HIP_KEM_CHAIN_ITEM", 9, 0, 20);
    return lg <= 0 ? size_t(0) : size_t(1) << lg;
}
size_t kem_small_table_bytes(size_t n) { return kem_cache_bytes(std::min(n, kem_small_batch())); }
size_t kem_ws_base(size_t n) { return up256(kKemWsPerItem * n) + kem_scratch_bytes(); }
size_t kem_small_table_ofs(size_t n) { return kem_ws_base(n); }
size_t kem_ws_bytes(size_t n) { return kem_ws_base(n) + kem_small_table_bytes(n); }
// What a call on n items cannot do without (the big-batch routes and the one-entry cache); with at least kem_ws_base(n) +
// kem_cache_bytes(n) bytes a batch of n <= kem_small_batch() takes the small-batch routes, with less the scratch routes.
size_t kem_ws_min(size_t n) { return kem_ws_base(n) + kem_cache_bytes(1); }
bool kem_small_route(size_t n, size_t ws_bytes) { return n <= kem_small_batch() && ws_bytes >= kem_ws_base(n) + kem_cache_bytes(n); }

// Items per ring-phase workgroup of a small batch: about two groups per SIMD -- one item per workgroup up to 2 x 4 x CUs items,
// then as few per group as that allows.  Measured 2^11 .. 2^15 (groups of up to four items run their PRF streams on lane pairs,
// prf_streams_split, so small groups are cheap): 8 groups per CU up to 2^13 items (the re-encryption: 2^12), 16 beyond.
size_t kem_small_group(size_t n, bool reencrypt = false) {
    static const size_t per_cu_env = (size_t)env_int("CIRCL_HIP_KEM_SMALL_WGS", 0, 1, 32);  // tuning aid
    const size_t per_cu = per_cu_env ? per_cu_env : (n <= (size_t(1) << (reencrypt ? 12 : 13)) ? 8 : 16);
    return std::max<size_t>(1, (n + per_cu * (size_t)cu_count() - 1) / (per_cu * (size_t)cu_count()));
}

int kem_k(int param) { return param == 512 ? 2 : param == 768 ? 3 : param == 1024 ? 4 : 0; }

// key-table cache behind the per-item workspace: A^T rows (whole groups of G entries), H(ek) and a status byte per entry
template <int K> size_t kem_table_bytes(size_t nkeys) {
    using Gm = circl::mlkem::Geom<K>;
    const size_t padded = (nkeys + Gm::G - 1) / Gm::G * Gm::G;
    return up256(padded * K * K * 512) + up256(nkeys * 32) + up256(nkeys);
}
size_t kem_table_bytes_any(int param, size_t nkeys) {
    switch (kem_k(param)) {
    case 2: return kem_table_bytes<2>(nkeys);
    case 3: return kem_table_bytes<3>(nkeys);
    case 4: return kem_table_bytes<4>(nkeys);
    }
    return 0;
}

struct KemWs {  // the carving of a ML-KEM workspace every launch sequence below uses
    uint8_t *slot0, *slot1, *slot2, *slot3, *status_slot;
    unsigned *work;
    uint8_t *scratch;
    KemWs(void *ws, size_t n) {
        slot0 = static_cast<uint8_t *>(ws);
        slot1 = slot0 + 32 * n;
        slot2 = slot0 + 64 * n;
        slot3 = slot0 + 96 * n;
        status_slot = slot0 + 128 * n;
        work = reinterpret_cast<unsigned *>(slot0 + up256(kKemWsPerItem * n));
        scratch = slot0 + up256(kKemWsPerItem * n) + 256;
    }
};

// ---- device-resident ML-KEM ---------------------------------------------------------------

// R3 = round-3 Kyber (kem/kyber/kyber768/kyber.go:105-154): m = H(seed), lenient key decoding, K = KDF(K' || H(ct)).
template <int K, bool R3 = false>
int encaps_dev_impl(const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n,
                    void *ws, size_t ws_bytes, hipStream_t st) {
    using Gm = circl::mlkem::Geom<K>;
    using namespace circl::mlkem;
    if (n == 0) return CIRCL_HIP_OK;
    if (ws_bytes < kem_ws_min(n) || !aligned16(ws) || !aligned16(ek) || !aligned16(m) || !aligned16(ct) || !aligned16(ss))
        return CIRCL_HIP_EWORKSPACE;
    KemWs w(ws, n);
    uint8_t *r_ws = w.slot0, *m_ws = w.slot1;
    const unsigned hb = (unsigned)((n + 255) / 256);
    if (!R3 && n <= kem_chain_item_batch()) {  // one launch: H(ek) -> G -> PRF beside A^T, then K-PKE.Encrypt, two wavefronts per item
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        hipLaunchKernelGGL((mlkem_encaps_chain_kernel<K, false>), dim3((unsigned)n), dim3(128), 0, st, ek, (size_t)Gm::EK, KeyIdx{},
                           (const int16_t *)nullptr, (const uint8_t *)nullptr, m, ct, ss, status, n);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
    if (!R3 && kem_small_route(n, ws_bytes)) {
        // small batch: [H(ek), G] and [A^T] side by side in one launch, then PRF + ring phase with the rows from the cache
        int16_t *key_rows = reinterpret_cast<int16_t *>(static_cast<uint8_t *>(ws) + kem_small_table_ofs(n));
        // up to kem_coop_batch() items the hashes run two items per wavefront (25 lanes per state): shorter chains while the chip
        // has SIMDs to spare (n / 2 hashing wavefronts); beyond it, one item per lane
        // (beyond that an item per lane pair, keccak_f1600_split: 2/3 of the chain at 4/3 of the issue slots)
        const int coop = kem_hash_form(n, kem_coop_batch());
        const unsigned nb_expand = (unsigned)((n + Gm::G - 1) / Gm::G), nb_hash = kem_hash_blocks(n, coop);
        static_assert(Gm::LDS_FIFO >= 108 * 8, "the cooperative hash's exchange area fits the FIFO area");
        const size_t want = kem_small_group(n);
        HIP_TRY(hipMemsetAsync(w.work, 0, 256, st));
        {
            ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
            hipLaunchKernelGGL(mlkem_small_pre_kernel<K>, dim3(nb_hash + nb_expand), dim3(64), Gm::LDS_FIFO, st, ek, m, ss, r_ws, key_rows, n, nb_hash, coop);
        }
        auto kern = mlkem_encrypt_kernel<K, ENCAPS, 0, true, KM_KEYED>;
        const unsigned eb = std::min<unsigned>((unsigned)((n + want - 1) / want), resident_blocks(kern, Gm::LDS_SHARED_TOTAL));
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        hipLaunchKernelGGL(kern, dim3(eb), dim3(64), Gm::LDS_SHARED_TOTAL, st, ek, (size_t)Gm::EK, m, (const uint8_t *)r_ws, ct, ss, status,
                           (const uint8_t *)nullptr, (const uint8_t *)nullptr, w.scratch, w.work, n, KeyIdx{}, (const int16_t *)key_rows);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
    HIP_TRY(hipMemsetAsync(w.work, 0, 256, st));
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
        if (R3) hipLaunchKernelGGL(kyber_r3_hash_kernel<K>, dim3(hb), dim3(256), 0, st, ek, m, ss, r_ws, m_ws, n);
        else hipLaunchKernelGGL(mlkem_hash_kernel<K>, dim3(hb), dim3(256), 0, st, ek, m, ss, r_ws, n);
    }
    {
        auto kern = mlkem_encrypt_kernel<K, R3 ? ENCAPS_LENIENT : ENCAPS, 0, true>;
        const unsigned eb = std::min<unsigned>((unsigned)((n + Gm::G - 1) / Gm::G), resident_blocks(kern, Gm::LDS_SCRATCH_TOTAL));
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        hipLaunchKernelGGL(kern, dim3(eb), dim3(64), Gm::LDS_SCRATCH_TOTAL, st, ek, (size_t)Gm::EK, R3 ? (const uint8_t *)m_ws : m, (const uint8_t *)r_ws,
                           ct, ss, status, (const uint8_t *)nullptr, (const uint8_t *)nullptr, w.scratch, w.work, n, KeyIdx{},
                           (const int16_t *)nullptr);
    }
    if (R3) {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
        hipLaunchKernelGGL(kyber_r3_finish_kernel<K>, dim3(hb), dim3(256), 0, st, (const uint8_t *)ct, ss, n);
    }
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

// Shared-key encapsulation: one ek for all n items (the reference's BenchmarkEncapsulate shape; SURVEY 8d "secondary
// input").  H(ek) and A^T are computed once (per launch / per resident workgroup), leaving 8 permutations per item.
template <int K>
int encaps_shared_dev_impl(const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, void *ws,
                           size_t ws_bytes, hipStream_t st) {
    using Gm = circl::mlkem::Geom<K>;
    using namespace circl::mlkem;
    if (n == 0) return CIRCL_HIP_OK;
    if (ws_bytes < kem_ws_min(n) || !aligned16(ws) || !aligned16(ek) || !aligned16(m) || !aligned16(ct) || !aligned16(ss))
        return CIRCL_HIP_EWORKSPACE;
    KemWs w(ws, n);
    uint8_t *r_ws = w.slot0, *h_ws = w.slot1;
    if (n <= kem_chain_item_batch()) {  // one launch, the key work inside every item's workgroup (key stride 0: the one key)
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        hipLaunchKernelGGL((mlkem_encaps_chain_kernel<K, false>), dim3((unsigned)n), dim3(128), 0, st, ek, (size_t)0, KeyIdx{},
                           (const int16_t *)nullptr, (const uint8_t *)nullptr, m, ct, ss, status, n);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
    HIP_TRY(hipMemsetAsync(w.work, 0, 256, st));
    if (n <= kem_small_shared_batch(false)) {
        // small batch: [A^T of the key] and [H(ek), G] side by side in one launch, then PRF + ring phase with the rows from the cache
        int16_t *key_rows = reinterpret_cast<int16_t *>(static_cast<uint8_t *>(ws) + kem_small_table_ofs(n));
        static_assert(Gm::LDS_FIFO >= 108 * 8, "the cooperative hash's exchange area fits the FIFO area");
        {
            ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
            hipLaunchKernelGGL(mlkem_small_shared_pre_kernel<K>, dim3(1 + (unsigned)((n + 31) / 32)), dim3(64), Gm::LDS_FIFO, st, ek, m, ss, r_ws, key_rows, n);
        }
        auto kern = mlkem_encrypt_kernel<K, ENCAPS, 0, true, KM_KEYED>;
        const size_t want = kem_small_group(n);
        const unsigned eb = std::min<unsigned>((unsigned)((n + want - 1) / want), resident_blocks(kern, Gm::LDS_SHARED_TOTAL));
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        hipLaunchKernelGGL(kern, dim3(eb), dim3(64), Gm::LDS_SHARED_TOTAL, st, ek, (size_t)0, m, (const uint8_t *)r_ws, ct, ss, status,
                           (const uint8_t *)nullptr, (const uint8_t *)nullptr, w.scratch, w.work, n, KeyIdx{}, (const int16_t *)key_rows);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
        hipLaunchKernelGGL(mlkem_hek_kernel<K>, dim3(1), dim3(64), 0, st, ek, h_ws);
        hipLaunchKernelGGL(mlkem_g_shared_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint8_t *)h_ws, m, ss,
                           r_ws, n, KeyIdx{});
    }
    {
        auto kern = mlkem_encrypt_kernel<K, ENCAPS, 0, true, KM_SHARED>;
        const unsigned eb = std::min<unsigned>((unsigned)((n + Gm::GS - 1) / Gm::GS), resident_blocks(kern, Gm::LDS_SHARED_TOTAL));
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        hipLaunchKernelGGL(kern, dim3(eb), dim3(64), Gm::LDS_SHARED_TOTAL, st, ek, (size_t)0, m, (const uint8_t *)r_ws, ct, ss, status,
                           (const uint8_t *)nullptr, (const uint8_t *)nullptr, w.scratch, w.work, n, KeyIdx{},
                           (const int16_t *)nullptr);
    }
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

// Key-table encapsulation: item i encapsulates to entry key_idx[i] of ek_table (nkeys rows).  Per TABLE ENTRY: H(ek) and
// A^T (the reference's parsed-key cache, kyber.go:39-43 / cpapke.go:19-25); per ITEM: G, the 2K+1 PRF streams and the ring
// phase (8 permutations for ML-KEM-768).  The workspace is kem_ws_bytes(n) followed by kem_table_bytes<K>(nkeys).
template <int K>
int encaps_keyed_dev_impl(const uint8_t *ek_table, size_t nkeys, const uint32_t *key_idx, const uint8_t *m, uint8_t *ct, uint8_t *ss,
                          uint8_t *status, size_t n, void *ws, size_t ws_bytes, hipStream_t st) {
    using Gm = circl::mlkem::Geom<K>;
    using namespace circl::mlkem;
    if (n == 0) return CIRCL_HIP_OK;
    if (nkeys == 0) return CIRCL_HIP_EPARAM;
    if (ws_bytes < kem_ws_bytes(n) + kem_table_bytes<K>(nkeys) || !aligned16(ws) || !aligned16(ek_table) || !aligned16(m) || !aligned16(ct) ||
        !aligned16(ss) || (reinterpret_cast<uintptr_t>(key_idx) & 3))
        return CIRCL_HIP_EWORKSPACE;
    KemWs w(ws, n);
    const KeyIdx kx{key_idx, (uint32_t)(nkeys - 1)};  // a device index vector is bounded to the table on every read
    uint8_t *r_ws = w.slot0;
    const size_t padded = (nkeys + Gm::G - 1) / Gm::G * Gm::G;
    int16_t *key_rows = reinterpret_cast<int16_t *>(static_cast<uint8_t *>(ws) + kem_ws_bytes(n));
    uint8_t *key_h = reinterpret_cast<uint8_t *>(key_rows) + up256(padded * K * K * 512);
    HIP_TRY(hipMemsetAsync(w.work, 0, 256, st));
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_KEYTABLE, st);
        hipLaunchKernelGGL(mlkem_hek_table_kernel<K>, dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, st, ek_table, (size_t)Gm::EK, (size_t)0,
                           key_h, (uint8_t *)nullptr, 0, nkeys);
        hipLaunchKernelGGL(mlkem_expand_keys_kernel<K>, dim3((unsigned)(padded / Gm::G)), dim3(64), Gm::LDS_FIFO, st, ek_table, (size_t)Gm::EK,
                           (size_t)(384 * K), key_rows, nkeys);
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
        hipLaunchKernelGGL(mlkem_g_shared_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint8_t *)key_h, m, ss, r_ws, n, kx);
    }
    {
        auto kern = mlkem_encrypt_kernel<K, ENCAPS, 0, true, KM_KEYED>;
        const unsigned eb = std::min<unsigned>((unsigned)((n + Gm::GS - 1) / Gm::GS), resident_blocks(kern, Gm::LDS_SHARED_TOTAL));
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        hipLaunchKernelGGL(kern, dim3(eb), dim3(64), Gm::LDS_SHARED_TOTAL, st, ek_table, (size_t)Gm::EK, m, (const uint8_t *)r_ws, ct, ss, status,
                           (const uint8_t *)nullptr, (const uint8_t *)nullptr, w.scratch, w.work, n, kx, (const int16_t *)key_rows);
    }
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

// Shared-key decapsulation: one dk for all n ciphertexts (the reference's parsed PrivateKey).  The key's hash check and
// A^T happen once; per item: decrypt, G, J(z || ct) and the re-encryption's 2K+1 PRF streams (17 permutations).
template <int K>
int decaps_shared_dev_impl(const uint8_t *dk, const uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, void *ws, size_t ws_bytes,
                           hipStream_t st) {
    using Gm = circl::mlkem::Geom<K>;
    using namespace circl::mlkem;
    if (n == 0) return CIRCL_HIP_OK;
    if (ws_bytes < kem_ws_min(n) || !aligned16(ws) || !aligned16(dk) || !aligned16(ct) || !aligned16(ss)) return CIRCL_HIP_EWORKSPACE;
    KemWs w(ws, n);
    uint8_t *mprime = w.slot0, *r_ws = w.slot1, *kbar = w.slot2, *ssrej = w.slot3;
    uint8_t *key_status = reinterpret_cast<uint8_t *>(w.work) + 128;  // second half of the ticket-counter slot
    if (n <= kem_chain_item_batch()) {  // one launch, four wavefronts per item, key stride 0: every workgroup checks and expands the one key
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_DECRYPT, st);
        hipLaunchKernelGGL((mlkem_decaps_chain_kernel<K, false>), dim3((unsigned)n), dim3(256), 0, st, dk, (size_t)0, KeyIdx{},
                           (const int16_t *)nullptr, (const uint8_t *)nullptr, ct, ss, status, n);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
    HIP_TRY(hipMemsetAsync(w.work, 0, 256, st));
    const unsigned hb = (unsigned)((n + 255) / 256);
    if (n <= kem_small_shared_batch(true)) {
        // small batch under one key: J(z || ct) per item, the key's hash check (one workgroup) and Decrypt + G side by side, then the
        // shared-key re-encryption with as few items per workgroup as the idle SIMDs allow
        const int coop = kem_hash_form(n, kem_coop_batch());
        const unsigned nb_j = kem_hash_blocks(n, coop);
        {
            ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
            // (+ ONE expansion workgroup behind the n decrypting ones: the key's A^T into the row cache, G copies -- stride 0)
            int16_t *key_rows = reinterpret_cast<int16_t *>(static_cast<uint8_t *>(ws) + kem_small_table_ofs(n));
            hipLaunchKernelGGL(mlkem_small_decaps_pre_kernel<K>, dim3(nb_j + 1 + (unsigned)n + 1), dim3(64), Gm::LDS_FIFO, st, dk, (size_t)0, ct, mprime, kbar, r_ws,
                               ssrej, status, key_status, key_rows, n, nb_j, 1u, coop);
            hipLaunchKernelGGL(mlkem_fill_status_kernel, dim3(hb), dim3(256), 0, st, status, (const uint8_t *)key_status, n);
        }
        const int16_t *key_rows = reinterpret_cast<const int16_t *>(static_cast<uint8_t *>(ws) + kem_small_table_ofs(n));
        auto kern = mlkem_encrypt_kernel<K, REENCRYPT, 0, true, KM_KEYED>;
        const size_t want = kem_small_group(n, true);
        const unsigned eb = std::min<unsigned>((unsigned)((n + want - 1) / want), resident_blocks(kern, Gm::LDS_SHARED_TOTAL));
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        hipLaunchKernelGGL(kern, dim3(eb), dim3(64), Gm::LDS_SHARED_TOTAL, st, dk + 384 * K, (size_t)0, (const uint8_t *)mprime, (const uint8_t *)r_ws,
                           const_cast<uint8_t *>(ct), ss, status, (const uint8_t *)kbar, (const uint8_t *)ssrej, w.scratch, w.work, n,
                           KeyIdx{}, key_rows);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_DECRYPT, st);
        hipLaunchKernelGGL(mlkem_decrypt_kernel<K>, dim3((unsigned)n), dim3(64), 0, st, dk, (size_t)0, ct, mprime, n, KeyIdx{});
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
        hipLaunchKernelGGL(mlkem_dk_check_kernel<K>, dim3(1), dim3(64), 0, st, dk, key_status);
        hipLaunchKernelGGL(mlkem_decaps_hash_kernel<K>, dim3(hb), dim3(256), 0, st, dk, (size_t)0, ct, (const uint8_t *)mprime, kbar,
                           r_ws, ssrej, status, n, (const uint8_t *)key_status, KeyIdx{});
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        auto kern = mlkem_encrypt_kernel<K, REENCRYPT, 0, true, KM_SHARED>;
        const unsigned eb = std::min<unsigned>((unsigned)((n + Gm::GS - 1) / Gm::GS), resident_blocks(kern, Gm::LDS_SHARED_TOTAL));
        hipLaunchKernelGGL(kern, dim3(eb), dim3(64), Gm::LDS_SHARED_TOTAL, st, dk + 384 * K, (size_t)0, (const uint8_t *)mprime, (const uint8_t *)r_ws,
                           const_cast<uint8_t *>(ct), ss, status, (const uint8_t *)kbar, (const uint8_t *)ssrej, w.scratch, w.work, n,
                           KeyIdx{}, (const int16_t *)nullptr);
    }
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

// Key-table decapsulation: item i is decapsulated with entry key_idx[i] of dk_table.  Per table entry: the private key's
// hash check (kyber.go:219-228) and A^T; per item the 17 permutations of the shared-key path.
template <int K>
int decaps_keyed_dev_impl(const uint8_t *dk_table, size_t nkeys, const uint32_t *key_idx, const uint8_t *ct, uint8_t *ss, uint8_t *status,
                          size_t n, void *ws, size_t ws_bytes, hipStream_t st) {
    using Gm = circl::mlkem::Geom<K>;
    using namespace circl::mlkem;
    if (n == 0) return CIRCL_HIP_OK;
    if (nkeys == 0) return CIRCL_HIP_EPARAM;
    if (ws_bytes < kem_ws_bytes(n) + kem_table_bytes<K>(nkeys) || !aligned16(ws) || !aligned16(dk_table) || !aligned16(ct) || !aligned16(ss) ||
        (reinterpret_cast<uintptr_t>(key_idx) & 3))
        return CIRCL_HIP_EWORKSPACE;
    KemWs w(ws, n);
    const KeyIdx kx{key_idx, (uint32_t)(nkeys - 1)};
    uint8_t *mprime = w.slot0, *r_ws = w.slot1, *kbar = w.slot2, *ssrej = w.slot3;
    const size_t padded = (nkeys + Gm::G - 1) / Gm::G * Gm::G;
    int16_t *key_rows = reinterpret_cast<int16_t *>(static_cast<uint8_t *>(ws) + kem_ws_bytes(n));
    uint8_t *key_h = reinterpret_cast<uint8_t *>(key_rows) + up256(padded * K * K * 512);
    uint8_t *key_status = key_h + up256(nkeys * 32);
    HIP_TRY(hipMemsetAsync(w.work, 0, 256, st));
    const unsigned hb = (unsigned)((n + 255) / 256);
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_KEYTABLE, st);
        hipLaunchKernelGGL(mlkem_hek_table_kernel<K>, dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, st, dk_table, (size_t)Gm::DK,
                           (size_t)(384 * K), key_h, key_status, 1, nkeys);
        hipLaunchKernelGGL(mlkem_expand_keys_kernel<K>, dim3((unsigned)(padded / Gm::G)), dim3(64), Gm::LDS_FIFO, st, dk_table, (size_t)Gm::DK,
                           (size_t)(768 * K), key_rows, nkeys);
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_DECRYPT, st);
        hipLaunchKernelGGL(mlkem_decrypt_kernel<K>, dim3((unsigned)n), dim3(64), 0, st, dk_table, (size_t)Gm::DK, ct, mprime, n, kx);
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
        hipLaunchKernelGGL(mlkem_decaps_hash_kernel<K>, dim3(hb), dim3(256), 0, st, dk_table, (size_t)Gm::DK, ct, (const uint8_t *)mprime, kbar,
                           r_ws, ssrej, status, n, (const uint8_t *)key_status, kx);
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        auto kern = mlkem_encrypt_kernel<K, REENCRYPT, 0, true, KM_KEYED>;
        const unsigned eb = std::min<unsigned>((unsigned)((n + Gm::GS - 1) / Gm::GS), resident_blocks(kern, Gm::LDS_SHARED_TOTAL));
        hipLaunchKernelGGL(kern, dim3(eb), dim3(64), Gm::LDS_SHARED_TOTAL, st, dk_table + 384 * K, (size_t)Gm::DK, (const uint8_t *)mprime,
                           (const uint8_t *)r_ws, const_cast<uint8_t *>(ct), ss, status, (const uint8_t *)kbar, (const uint8_t *)ssrej, w.scratch,
                           w.work, n, kx, (const int16_t *)key_rows);
    }
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

// R3 = round-3 Kyber (kem/kyber/kyber768/kyber.go:156-197): no private-key check, K = KDF((ct' == ct ? K'' : z) || H(ct));
// `status` may then be null (an n-byte slot of the workspace is used).
template <int K, bool R3 = false>
int decaps_dev_impl(const uint8_t *dk, const uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, void *ws, size_t ws_bytes,
                    hipStream_t st) {
    using Gm = circl::mlkem::Geom<K>;
    using namespace circl::mlkem;
    if (n == 0) return CIRCL_HIP_OK;
    if (ws_bytes < kem_ws_min(n) || !aligned16(ws) || !aligned16(dk) || !aligned16(ct) || !aligned16(ss)) return CIRCL_HIP_EWORKSPACE;
    KemWs w(ws, n);
    uint8_t *mprime = w.slot0, *r_ws = w.slot1, *kbar = w.slot2, *ssrej = w.slot3;
    if (R3) status = w.status_slot;
    if (!R3 && n <= kem_chain_item_batch()) {  // one launch, four wavefronts per item: [Decrypt -> G -> PRF] J, H(ek) check, A^T, then the re-encryption
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_DECRYPT, st);
        hipLaunchKernelGGL((mlkem_decaps_chain_kernel<K, false>), dim3((unsigned)n), dim3(256), 0, st, dk, (size_t)Gm::DK, KeyIdx{},
                           (const int16_t *)nullptr, (const uint8_t *)nullptr, ct, ss, status, n);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
    HIP_TRY(hipMemsetAsync(w.work, 0, 256, st));
    const unsigned hb = (unsigned)((n + 255) / 256);
    if (!R3 && kem_small_route(n, ws_bytes)) {
        // small batch: J(z || ct), the key's hash check, Decrypt + G and A^T side by side in one launch, then the key-table form
        // of the re-encryption (mlkem_small_decaps_pre_kernel)
        int16_t *key_rows = reinterpret_cast<int16_t *>(static_cast<uint8_t *>(ws) + kem_small_table_ofs(n));
        const int coop = kem_hash_form(n, kem_coop_batch() / 2);  // two sponges per item here: half the encapsulation's threshold
        const unsigned nb_hash = kem_hash_blocks(n, coop), nb_expand = (unsigned)((n + Gm::G - 1) / Gm::G);
        {
            ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
            hipLaunchKernelGGL(mlkem_small_decaps_pre_kernel<K>, dim3(2 * nb_hash + (unsigned)n + nb_expand), dim3(64), Gm::LDS_FIFO, st, dk, (size_t)Gm::DK, ct,
                               mprime, kbar, r_ws, ssrej, status, (uint8_t *)nullptr, key_rows, n, nb_hash, nb_hash, coop);
        }
        auto kern = mlkem_encrypt_kernel<K, REENCRYPT, 0, true, KM_KEYED>;
        const size_t want = kem_small_group(n, true);
        const unsigned eb = std::min<unsigned>((unsigned)((n + want - 1) / want), resident_blocks(kern, Gm::LDS_SHARED_TOTAL));
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        hipLaunchKernelGGL(kern, dim3(eb), dim3(64), Gm::LDS_SHARED_TOTAL, st, dk + 384 * K, (size_t)Gm::DK, (const uint8_t *)mprime, (const uint8_t *)r_ws,
                           const_cast<uint8_t *>(ct), ss, status, (const uint8_t *)kbar, (const uint8_t *)ssrej, w.scratch, w.work, n,
                           KeyIdx{}, (const int16_t *)key_rows);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_DECRYPT, st);
        hipLaunchKernelGGL(mlkem_decrypt_kernel<K>, dim3((unsigned)n), dim3(64), 0, st, dk, (size_t)Gm::DK, ct, mprime, n, KeyIdx{});
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
        if (R3)
            hipLaunchKernelGGL(kyber_r3_decaps_hash_kernel<K>, dim3(hb), dim3(256), 0, st, dk, (const uint8_t *)mprime, kbar, r_ws,
                               ssrej, status, n);
        else
            hipLaunchKernelGGL(mlkem_decaps_hash_kernel<K>, dim3(hb), dim3(256), 0, st, dk, (size_t)Gm::DK, ct, (const uint8_t *)mprime,
                               kbar, r_ws, ssrej, status, n, (const uint8_t *)nullptr, KeyIdx{});
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        auto kern = mlkem_encrypt_kernel<K, REENCRYPT, 0, true>;
        const unsigned eb = std::min<unsigned>((unsigned)((n + Gm::G - 1) / Gm::G), resident_blocks(kern, Gm::LDS_SCRATCH_TOTAL));
        hipLaunchKernelGGL(kern, dim3(eb), dim3(64), Gm::LDS_SCRATCH_TOTAL, st, dk + 384 * K, (size_t)Gm::DK, (const uint8_t *)mprime,
                           (const uint8_t *)r_ws, const_cast<uint8_t *>(ct), ss, status, (const uint8_t *)kbar, (const uint8_t *)ssrej,
                           w.scratch, w.work, n, KeyIdx{}, (const int16_t *)nullptr);
    }
    if (R3) {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
        hipLaunchKernelGGL(kyber_r3_finish_kernel<K>, dim3(hb), dim3(256), 0, st, ct, ss, n);
    }
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

template <int K, bool R3 = false>
int keygen_dev_impl(const uint8_t *seed64, uint8_t *ek, uint8_t *dk, size_t n, void *ws, size_t ws_bytes, hipStream_t st) {
    using Gm = circl::mlkem::Geom<K>;
    using namespace circl::mlkem;
    if (n == 0) return CIRCL_HIP_OK;
    if (ws_bytes < kem_ws_min(n) || !aligned16(ws) || !aligned16(seed64) || !aligned16(ek) || !aligned16(dk)) return CIRCL_HIP_EWORKSPACE;
    KemWs w(ws, n);
    uint8_t *rs = w.slot0;
    if (n <= kem_chain_item_batch()) {  // one launch, two wavefronts per key: G -> [PRF, NTT(s)] beside A, then t-hat, packing, H(ek) || z
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_KEYGEN, st);
        hipLaunchKernelGGL((mlkem_keygen_chain_kernel<K, R3>), dim3((unsigned)n), dim3(128), 0, st, seed64, ek, dk, n);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
    HIP_TRY(hipMemsetAsync(w.work, 0, 256, st));
    const unsigned hb = (unsigned)((n + 255) / 256);
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
        hipLaunchKernelGGL((mlkem_keygen_seed_kernel<K, R3>), dim3(hb), dim3(256), 0, st, seed64, rs, n);
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_KEYGEN, st);
        auto kern = mlkem_keygen_kernel<K, true>;
        const unsigned kb = std::min<unsigned>((unsigned)((n + Gm::G - 1) / Gm::G), resident_blocks(kern, Gm::LDS_SCRATCH_TOTAL));
        hipLaunchKernelGGL(kern, dim3(kb), dim3(64), Gm::LDS_SCRATCH_TOTAL, st, (const uint8_t *)rs, ek, dk, w.scratch, w.work, n);
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_FINISH, st);
        // H(ek): nine permutations per key.  Up to 2^15 keys a lane per key leaves the SIMDs at or below one wavefront each, and
        // the chain is the launch: two keys per wavefront / a key per lane pair there (kem_hash_form, as for the encapsulation)
        const int form = n <= (size_t(1) << 15) ? kem_hash_form(n, kem_coop_batch()) : 0;
        if (form) hipLaunchKernelGGL(mlkem_keygen_finish_small_kernel<K>, dim3(kem_hash_blocks(n, form)), dim3(64), 0, st, seed64, (const uint8_t *)ek, dk, n, form);
        else hipLaunchKernelGGL(mlkem_keygen_finish_kernel<K>, dim3(hb), dim3(256), 0, st, seed64, (const uint8_t *)ek, dk, n);
    }
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

// Every ML-KEM host-buffer call wipes what is secret in a chunk's device staging once its results are out: the secret inputs and
// outputs (seeds, m, private keys, shared secrets: flagged per array) and the per-item workspace slots (the coins r, G's output, m',
// sigma).  The public keys, the ciphertexts, the matrix scratch and the row cache behind the slots are public and are NOT zeroed
// (a whole-slot memset was ~390 MB per 2^15-item chunk).
PipeOpts kem_opts() {
    PipeOpts o;
    o.chunk_items = host_chunk_items(size_t(1) << 15);
    o.wipe_device = true;
    o.ws_secret_bytes = [](size_t cnt) { return up256(kKemWsPerItem * cnt); };
    o.tail_flag_ok = true;  // (every coalesced ML-KEM launch is exactly one *_dev call; the resident-key one-launch routes take the offer)
    return o;
}
// (a host-buffer chunk beyond 2^13 items is PCIe-bound whichever route it takes: it gets the workspace of the scratch routes, so
// a staging slot carries at most 64 MB of row cache instead of 256 MB)
std::function<size_t(size_t)> kem_ws_fn() {
    return [](size_t cnt) { return cnt <= (size_t(1) << 13) ? kem_ws_bytes(cnt) : kem_ws_min(cnt); };
}

// host-side check of a key-index vector (the device path trusts its caller: an out-of-range index would read past the table)
int check_key_idx(const uint32_t *key_idx, size_t n, size_t nkeys) {
    if (nkeys == 0 || nkeys > 0xffffffffull) return CIRCL_HIP_EPARAM;
    for (size_t i = 0; i < n; i++)
        if (key_idx[i] >= nkeys) return CIRCL_HIP_EPARAM;
    return CIRCL_HIP_OK;
}

// ---- key tables that live across calls (keytable.h) ------------------------------------------------------------------------------
// The table memory has the layout of the per-call key tables' workspace tail (kem_table_bytes): A^T rows of whole groups, H(ek)
// per entry, a status byte per entry.  key_idx == nullptr: every item uses entry 0 (the kernels see key stride 0).
template <int K> int kem_table_build(circl_hip_keytable *t, hipStream_t st) {
    using Gm = circl::mlkem::Geom<K>;
    using namespace circl::mlkem;
    const size_t padded = (t->nkeys + Gm::G - 1) / Gm::G * Gm::G;
    int16_t *key_rows = reinterpret_cast<int16_t *>(t->d_table);
    uint8_t *key_h = t->d_table + up256(padded * K * K * 512);
    uint8_t *key_status = key_h + up256(t->nkeys * 32);
    const size_t row = t->private_keys ? Gm::DK : Gm::EK;
    hipLaunchKernelGGL(mlkem_hek_table_kernel<K>, dim3((unsigned)((t->nkeys + 255) / 256)), dim3(256), 0, st, (const uint8_t *)t->d_keys, row,
                       (size_t)(t->private_keys ? 384 * K : 0), key_h, key_status, t->private_keys ? 1 : 0, t->nkeys);
    hipLaunchKernelGGL(mlkem_expand_keys_kernel<K>, dim3((unsigned)(padded / Gm::G)), dim3(64), Gm::LDS_FIFO, st, (const uint8_t *)t->d_keys, row,
                       (size_t)(t->private_keys ? 768 * K : 384 * K), key_rows, t->nkeys);
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}
template <int K>
int encaps_table_dev_impl(const circl_hip_keytable *t, const uint32_t *key_idx, const uint8_t *m, uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n,
                          void *ws, size_t ws_bytes, hipStream_t st) {
    using Gm = circl::mlkem::Geom<K>;
    using namespace circl::mlkem;
    if (n == 0) return CIRCL_HIP_OK;
    if (ws_bytes < kem_ws_min(n) || !aligned16(ws) || !aligned16(m) || !aligned16(ct) || !aligned16(ss) || (reinterpret_cast<uintptr_t>(key_idx) & 3))
        return CIRCL_HIP_EWORKSPACE;
    KemWs w(ws, n);
    const KeyIdx kx{key_idx, (uint32_t)(t->nkeys - 1)};
    uint8_t *r_ws = w.slot0;
    const size_t padded = (t->nkeys + Gm::G - 1) / Gm::G * Gm::G;
    const int16_t *key_rows = reinterpret_cast<const int16_t *>(t->d_table);
    const uint8_t *key_h = t->d_table + up256(padded * K * K * 512);
    const size_t stride = key_idx ? (size_t)Gm::EK : 0;
    if (n <= kem_chain_batch(false)) {  // one launch, a wavefront per item: G -> PRF -> K-PKE.Encrypt (mlkem_encaps_chain_kernel)
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        circl::TailFlag tail{nullptr, nullptr, 0};  // a coalesced batch's completion flag, raised by this launch's last workgroup (the kernel
        take_tail_flag(&tail.flag, &tail.count, &tail.value);  // keeps everything in LDS: no workspace to wipe behind it)
        hipLaunchKernelGGL(mlkem_encaps_chain_kernel<K>, dim3((unsigned)n), dim3(64), 0, st, (const uint8_t *)t->d_keys, (size_t)Gm::EK, kx, key_rows, key_h, m, ct,
                           ss, status, n, tail);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
    HIP_TRY(hipMemsetAsync(w.work, 0, 256, st));
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
        hipLaunchKernelGGL(mlkem_g_shared_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, key_h, m, ss, r_ws, n, kx);
    }
    {
        auto kern = mlkem_encrypt_kernel<K, ENCAPS, 0, true, KM_KEYED>;
        // small batches: as few items per workgroup as the idle SIMDs allow (kem_small_group), like the other latency-oriented routes
        const size_t want = n <= kem_small_shared_batch(false) ? kem_small_group(n) : (size_t)Gm::GS;
        const unsigned eb = std::min<unsigned>((unsigned)((n + want - 1) / want), resident_blocks(kern, Gm::LDS_SHARED_TOTAL));
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        hipLaunchKernelGGL(kern, dim3(eb), dim3(64), Gm::LDS_SHARED_TOTAL, st, (const uint8_t *)t->d_keys, stride, m, (const uint8_t *)r_ws, ct, ss, status,
                           (const uint8_t *)nullptr, (const uint8_t *)nullptr, w.scratch, w.work, n, kx, key_rows);
    }
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}
template <int K>
int decaps_table_dev_impl(const circl_hip_keytable *t, const uint32_t *key_idx, const uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, void *ws,
                          size_t ws_bytes, hipStream_t st) {
    using Gm = circl::mlkem::Geom<K>;
    using namespace circl::mlkem;
    if (n == 0) return CIRCL_HIP_OK;
    if (ws_bytes < kem_ws_min(n) || !aligned16(ws) || !aligned16(ct) || !aligned16(ss) || (reinterpret_cast<uintptr_t>(key_idx) & 3)) return CIRCL_HIP_EWORKSPACE;
    KemWs w(ws, n);
    const KeyIdx kx{key_idx, (uint32_t)(t->nkeys - 1)};
    uint8_t *mprime = w.slot0, *r_ws = w.slot1, *kbar = w.slot2, *ssrej = w.slot3;
    const size_t padded = (t->nkeys + Gm::G - 1) / Gm::G * Gm::G;
    const int16_t *key_rows = reinterpret_cast<const int16_t *>(t->d_table);
    const uint8_t *key_status = t->d_table + up256(padded * K * K * 512) + up256(t->nkeys * 32);
    const uint8_t *dk = t->d_keys;
    const size_t stride = key_idx ? (size_t)Gm::DK : 0;
    if (status == nullptr) status = w.status_slot;  // (the kernels want one; the caller may not)
    const unsigned hb = (unsigned)((n + 255) / 256);
    // up to kem_chain_batch() items: the whole decapsulation of an item in ONE launch, a two-wavefront workgroup per item
    // (mlkem_decaps_chain_kernel: J beside Decrypt -> G -> PRF -> re-encryption, one barrier, then the select)
    if (n <= kem_chain_batch()) {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_DECRYPT, st);
        circl::TailFlag tail{nullptr, nullptr, 0};
        take_tail_flag(&tail.flag, &tail.count, &tail.value);
        hipLaunchKernelGGL(mlkem_decaps_chain_kernel<K>, dim3((unsigned)n), dim3(128), 0, st, dk, (size_t)Gm::DK, kx, key_rows, key_status, ct, ss, status, n, tail);
        HIP_TRY(hipGetLastError());
        return CIRCL_HIP_OK;
    }
    HIP_TRY(hipMemsetAsync(w.work, 0, 256, st));
    const bool small = key_idx == nullptr && n <= kem_small_shared_batch(true);
    if (small) {
        // ONE key, small batch: J(z || ct) on the cooperative permutation / lane pairs beside Decrypt + G (mlkem_small_decaps_pre_kernel
        // without its hash-check and expansion workgroups: both are in the table)
        const int coop = kem_hash_form(n, kem_coop_batch());
        const unsigned nb_j = kem_hash_blocks(n, coop);
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
        hipLaunchKernelGGL(mlkem_small_decaps_pre_kernel<K>, dim3(nb_j + (unsigned)n), dim3(64), Gm::LDS_FIFO, st, dk, (size_t)0, ct, mprime, kbar, r_ws, ssrej,
                           status, (uint8_t *)nullptr, (int16_t *)nullptr, n, nb_j, 0u, coop);
        hipLaunchKernelGGL(mlkem_fill_status_kernel, dim3(hb), dim3(256), 0, st, status, key_status, n);
    } else {
        {
            ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_DECRYPT, st);
            hipLaunchKernelGGL(mlkem_decrypt_kernel<K>, dim3((unsigned)n), dim3(64), 0, st, dk, stride, ct, mprime, n, kx);
        }
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_HASH, st);
        hipLaunchKernelGGL(mlkem_decaps_hash_kernel<K>, dim3(hb), dim3(256), 0, st, dk, stride, ct, (const uint8_t *)mprime, kbar, r_ws, ssrej, status, n,
                           key_status, kx);
    }
    {
        ProfScope ps(CIRCL_HIP_KERNEL_MLKEM_ENCRYPT, st);
        auto kern = mlkem_encrypt_kernel<K, REENCRYPT, 0, true, KM_KEYED>;
        const size_t want = n <= kem_small_shared_batch(true) ? kem_small_group(n, true) : (size_t)Gm::GS;
        const unsigned eb = std::min<unsigned>((unsigned)((n + want - 1) / want), resident_blocks(kern, Gm::LDS_SHARED_TOTAL));
        hipLaunchKernelGGL(kern, dim3(eb), dim3(64), Gm::LDS_SHARED_TOTAL, st, dk + 384 * K, stride, (const uint8_t *)mprime, (const uint8_t *)r_ws,
                           const_cast<uint8_t *>(ct), ss, status, (const uint8_t *)kbar, (const uint8_t *)ssrej, w.scratch, w.work, n, kx, key_rows);
    }
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}
bool kem_table_ok(const circl_hip_keytable *t, int want_private) {
    return t && t->magic == kKeytableMagic && t->family == 1 && t->private_keys == want_private && kem_k(t->param) != 0;
}

}  // namespace

// for the hybrid KEMs' host pipeline (api_hybrid.hip), which carries an ML-KEM workspace behind its own rows: the smallest workspace a
// chunk of n items can run in (the scratch routes: no 8 KB-per-item row cache) and how much of a workspace's head is secret
namespace circl {
namespace host {
size_t mlkem_ws_min_bytes(size_t n) { return kem_ws_min(n); }
size_t mlkem_ws_secret_bytes(size_t n) { return up256(kKemWsPerItem * n); }
}  // namespace host
}  // namespace circl

// =============================================================================================
extern "C" {

size_t circl_hip_mlkem_ek_size(int param) { const int k = kem_k(param); return k ? 384 * k + 32 : 0; }
size_t circl_hip_mlkem_dk_size(int param) { const int k = kem_k(param); return k ? 768 * k + 96 : 0; }
size_t circl_hip_mlkem_ct_size(int param) {
    switch (param) {
    case 512: return 768;
    case 768: return 1088;
    case 1024: return 1568;
    }
    return 0;
}

size_t circl_hip_mlkem_workspace_size(int param, size_t n) { return kem_k(param) ? kem_ws_bytes(n) : 0; }
size_t circl_hip_mlkem_keyed_workspace_size(int param, size_t n, size_t nkeys) {
    return kem_k(param) ? kem_ws_bytes(n) + kem_table_bytes_any(param, nkeys) : 0;
}

#define KEM_DISPATCH(call2, call3, call4)                  \
    if (ndev() <= 0) return CIRCL_HIP_ENODEV;              \
    hipStream_t st = static_cast<hipStream_t>(stream);     \
    switch (kem_k(param)) {                                \
    case 2: return call2;                                  \
    case 3: return call3;                                  \
    case 4: return call4;                                  \
    }                                                      \
    return CIRCL_HIP_EPARAM

int circl_hip_mlkem_encaps_dev(int param, const uint8_t *d_ek, const uint8_t *d_m, uint8_t *d_ct, uint8_t *d_ss,
                               uint8_t *d_status, size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    KEM_DISPATCH(encaps_dev_impl<2>(d_ek, d_m, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 encaps_dev_impl<3>(d_ek, d_m, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 encaps_dev_impl<4>(d_ek, d_m, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st));
}
int circl_hip_mlkem_decaps_dev(int param, const uint8_t *d_dk, const uint8_t *d_ct, uint8_t *d_ss, uint8_t *d_status, size_t n,
                               void *d_ws, size_t ws_bytes, void *stream) {
    KEM_DISPATCH(decaps_dev_impl<2>(d_dk, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 decaps_dev_impl<3>(d_dk, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 decaps_dev_impl<4>(d_dk, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st));
}
int circl_hip_mlkem_keygen_dev(int param, const uint8_t *d_seed64, uint8_t *d_ek, uint8_t *d_dk, size_t n, void *d_ws,
                               size_t ws_bytes, void *stream) {
    KEM_DISPATCH(keygen_dev_impl<2>(d_seed64, d_ek, d_dk, n, d_ws, ws_bytes, st),
                 keygen_dev_impl<3>(d_seed64, d_ek, d_dk, n, d_ws, ws_bytes, st),
                 keygen_dev_impl<4>(d_seed64, d_ek, d_dk, n, d_ws, ws_bytes, st));
}
int circl_hip_mlkem_encaps_shared_dev(int param, const uint8_t *d_ek, const uint8_t *d_m, uint8_t *d_ct, uint8_t *d_ss, uint8_t *d_status,
                                      size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    KEM_DISPATCH(encaps_shared_dev_impl<2>(d_ek, d_m, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 encaps_shared_dev_impl<3>(d_ek, d_m, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 encaps_shared_dev_impl<4>(d_ek, d_m, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st));
}
int circl_hip_mlkem_decaps_shared_dev(int param, const uint8_t *d_dk, const uint8_t *d_ct, uint8_t *d_ss, uint8_t *d_status, size_t n,
                                      void *d_ws, size_t ws_bytes, void *stream) {
    KEM_DISPATCH(decaps_shared_dev_impl<2>(d_dk, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 decaps_shared_dev_impl<3>(d_dk, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 decaps_shared_dev_impl<4>(d_dk, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st));
}
int circl_hip_mlkem_encaps_keyed_dev(int param, const uint8_t *d_ek_table, size_t nkeys, const uint32_t *d_key_idx, const uint8_t *d_m,
                                     uint8_t *d_ct, uint8_t *d_ss, uint8_t *d_status, size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    KEM_DISPATCH(encaps_keyed_dev_impl<2>(d_ek_table, nkeys, d_key_idx, d_m, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 encaps_keyed_dev_impl<3>(d_ek_table, nkeys, d_key_idx, d_m, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 encaps_keyed_dev_impl<4>(d_ek_table, nkeys, d_key_idx, d_m, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st));
}
int circl_hip_mlkem_decaps_keyed_dev(int param, const uint8_t *d_dk_table, size_t nkeys, const uint32_t *d_key_idx, const uint8_t *d_ct,
                                     uint8_t *d_ss, uint8_t *d_status, size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    KEM_DISPATCH(decaps_keyed_dev_impl<2>(d_dk_table, nkeys, d_key_idx, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 decaps_keyed_dev_impl<3>(d_dk_table, nkeys, d_key_idx, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 decaps_keyed_dev_impl<4>(d_dk_table, nkeys, d_key_idx, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st));
}

// ---- host buffers ---------------------------------------------------------------------------------

int circl_hip_mlkem_encaps(int param, const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss, uint8_t *status,
                           size_t n, int device) {
    const size_t EK = circl_hip_mlkem_ek_size(param), CT = circl_hip_mlkem_ct_size(param);
    if (!EK) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        const std::vector<HIn> ins = {{ek + lo * EK, EK}, {m + lo * 32, 32, true}};
        const std::vector<HOut> outs = {{ct + lo * CT, CT}, {ss + lo * 32, 32, true}, {status ? status + lo : nullptr, 1}};
        auto launch = [&](Chunk &c) { return circl_hip_mlkem_encaps_dev(param, c.in[0], c.in[1], c.out[0], c.out[1], c.out[2], c.cnt, c.ws, c.ws_bytes, c.st); };
        // circl_hip_set_coalesce: the small calls of concurrent callers share launches -- a TLS server's shape: every handshake
        // encapsulates once, to a key of its own (kem/hybrid/hybrid.go:95-99 -> kem/mlkem/mlkem768/kyber.go:359-370)
        Coalescer *co = all_inputs_present(ins) ? call_coalescer(kCoKemEncaps, kem_k(param) - 2, dev) : nullptr;
        if (co) {
            const int rc = coalesce_run(co, cnt, ins, {}, outs, kem_ws_fn(), kem_opts(), launch);
            if (rc != kNotCoalesced) return rc;
        }
        return run_pipeline(dev, cnt, ins, {}, outs, kem_ws_fn(), kem_opts(), launch);
    });
}

int circl_hip_mlkem_decaps(int param, const uint8_t *dk, const uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, int device) {
    const size_t DK = circl_hip_mlkem_dk_size(param), CT = circl_hip_mlkem_ct_size(param);
    if (!DK) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        const std::vector<HIn> ins = {{dk + lo * DK, DK, true}, {ct + lo * CT, CT}};
        const std::vector<HOut> outs = {{ss + lo * 32, 32, true}, {status ? status + lo : nullptr, 1}};
        auto launch = [&](Chunk &c) { return circl_hip_mlkem_decaps_dev(param, c.in[0], c.in[1], c.out[0], c.out[1], c.cnt, c.ws, c.ws_bytes, c.st); };
        Coalescer *co = all_inputs_present(ins) ? call_coalescer(kCoKemDecaps, kem_k(param) - 2, dev) : nullptr;
        if (co) {  // circl_hip_set_coalesce: the small calls of concurrent callers share launches
            const int rc = coalesce_run(co, cnt, ins, {}, outs, kem_ws_fn(), kem_opts(), launch);
            if (rc != kNotCoalesced) return rc;
        }
        return run_pipeline(dev, cnt, ins, {}, outs, kem_ws_fn(), kem_opts(), launch);
    });
}

int circl_hip_mlkem_keygen(int param, const uint8_t *seed64, uint8_t *ek, uint8_t *dk, size_t n, int device) {
    const size_t EK = circl_hip_mlkem_ek_size(param), DK = circl_hip_mlkem_dk_size(param);
    if (!EK) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{seed64 + lo * 64, 64, true}}, {}, {{ek + lo * EK, EK}, {dk + lo * DK, DK, true}}, kem_ws_fn(), kem_opts(),
                            [&](Chunk &c) { return circl_hip_mlkem_keygen_dev(param, c.in[0], c.out[0], c.out[1], c.cnt, c.ws, c.ws_bytes, c.st); });
    });
}

int circl_hip_mlkem_decaps_shared(int param, const uint8_t *dk, const uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, int device) {
    const size_t DK = circl_hip_mlkem_dk_size(param), CT = circl_hip_mlkem_ct_size(param);
    if (!DK) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{dk, DK, true, true}, {ct + lo * CT, CT}}, {}, {{ss + lo * 32, 32, true}, {status ? status + lo : nullptr, 1}},
                            kem_ws_fn(), kem_opts(), [&](Chunk &c) {
                                return circl_hip_mlkem_decaps_shared_dev(param, c.in[0], c.in[1], c.out[0], c.out[1], c.cnt, c.ws, c.ws_bytes, c.st);
                            });
    });
}
int circl_hip_mlkem_encaps_shared(int param, const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n,
                                  int device) {
    const size_t EK = circl_hip_mlkem_ek_size(param), CT = circl_hip_mlkem_ct_size(param);
    if (!EK) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{ek, EK, false, true}, {m + lo * 32, 32, true}}, {},
                            {{ct + lo * CT, CT}, {ss + lo * 32, 32, true}, {status ? status + lo : nullptr, 1}}, kem_ws_fn(), kem_opts(), [&](Chunk &c) {
                                return circl_hip_mlkem_encaps_shared_dev(param, c.in[0], c.in[1], c.out[0], c.out[1], c.out[2], c.cnt, c.ws, c.ws_bytes, c.st);
                            });
    });
}

// Key tables through host buffers: the table travels with every chunk (it is small next to a chunk of items: nkeys rows
// against 2^15 items), so each chunk is self-contained and chunks overlap like those of the other entry points.
int circl_hip_mlkem_encaps_keyed(int param, const uint8_t *ek_table, size_t nkeys, const uint32_t *key_idx, const uint8_t *m, uint8_t *ct,
                                 uint8_t *ss, uint8_t *status, size_t n, int device) {
    const size_t EK = circl_hip_mlkem_ek_size(param), CT = circl_hip_mlkem_ct_size(param);
    if (!EK) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    if (int rc = check_key_idx(key_idx, n, nkeys)) return rc;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{ek_table, EK * nkeys, false, true}, {reinterpret_cast<const uint8_t *>(key_idx + lo), 4}, {m + lo * 32, 32, true}}, {},
                            {{ct + lo * CT, CT}, {ss + lo * 32, 32, true}, {status ? status + lo : nullptr, 1}},
                            [&](size_t c) { return circl_hip_mlkem_keyed_workspace_size(param, c, nkeys); }, kem_opts(), [&](Chunk &c) {
                                return circl_hip_mlkem_encaps_keyed_dev(param, c.in[0], nkeys, reinterpret_cast<const uint32_t *>(c.in[1]), c.in[2], c.out[0],
                                                                        c.out[1], c.out[2], c.cnt, c.ws, c.ws_bytes, c.st);
                            });
    });
}
int circl_hip_mlkem_decaps_keyed(int param, const uint8_t *dk_table, size_t nkeys, const uint32_t *key_idx, const uint8_t *ct, uint8_t *ss,
                                 uint8_t *status, size_t n, int device) {
    const size_t DK = circl_hip_mlkem_dk_size(param), CT = circl_hip_mlkem_ct_size(param);
    if (!DK) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    if (int rc = check_key_idx(key_idx, n, nkeys)) return rc;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{dk_table, DK * nkeys, true, true}, {reinterpret_cast<const uint8_t *>(key_idx + lo), 4}, {ct + lo * CT, CT}}, {},
                            {{ss + lo * 32, 32, true}, {status ? status + lo : nullptr, 1}},
                            [&](size_t c) { return circl_hip_mlkem_keyed_workspace_size(param, c, nkeys); }, kem_opts(), [&](Chunk &c) {
                                return circl_hip_mlkem_decaps_keyed_dev(param, c.in[0], nkeys, reinterpret_cast<const uint32_t *>(c.in[1]), c.in[2], c.out[0],
                                                                        c.out[1], c.cnt, c.ws, c.ws_bytes, c.st);
                            });
    });
}

// ---- key tables that live across calls ---------------------------------------------------------------------------------------
static int kem_keytable_new_one(int param, int private_keys, const uint8_t *keys, size_t nkeys, int device, uint8_t *key_status,
                                circl_hip_keytable **out) {
    const int K = kem_k(param);
    HIP_TRY(hipSetDevice(physical_device(device)));
    circl_hip_keytable *t = new (std::nothrow) circl_hip_keytable();
    if (!t) return CIRCL_HIP_ENOMEM;
    t->magic = kKeytableMagic; t->family = 1; t->param = param; t->device = device; t->private_keys = private_keys ? 1 : 0; t->nkeys = nkeys;
    t->row = private_keys ? circl_hip_mlkem_dk_size(param) : circl_hip_mlkem_ek_size(param);
    t->keys_bytes = up256(t->row * nkeys + 16);
    t->table_bytes = kem_table_bytes_any(param, nkeys);
    hipStream_t h2d = nullptr, d2h = nullptr, st = nullptr;
    int rc = pipeline_streams(device, &h2d, &d2h, &st);
    if (rc == CIRCL_HIP_OK && (hipMalloc(reinterpret_cast<void **>(&t->d_keys), t->keys_bytes) != hipSuccess ||
                               hipMalloc(reinterpret_cast<void **>(&t->d_table), t->table_bytes) != hipSuccess)) {
        (void)hipGetLastError();
        rc = CIRCL_HIP_ENOMEM;
    }
    if (rc == CIRCL_HIP_OK) {  // (private rows: through wiped page-locked staging)
        if (private_keys) rc = upload_secret(t->d_keys, keys, t->row * nkeys, st);
        else if (hipMemcpyAsync(t->d_keys, keys, t->row * nkeys, hipMemcpyHostToDevice, st) != hipSuccess) rc = CIRCL_HIP_EHIP;
    }
    if (rc == CIRCL_HIP_OK) rc = K == 2 ? kem_table_build<2>(t, st) : K == 3 ? kem_table_build<3>(t, st) : kem_table_build<4>(t, st);
    if (rc == CIRCL_HIP_OK && key_status) {
        const size_t G = K == 2 ? 16 : K == 3 ? 7 : 4, padded = (nkeys + G - 1) / G * G;  // (Geom<K>::G)
        const uint8_t *ks = t->d_table + up256(padded * K * K * 512) + up256(nkeys * 32);
        if (private_keys) {
            if (hipMemcpyAsync(key_status, ks, nkeys, hipMemcpyDeviceToHost, st) != hipSuccess) rc = CIRCL_HIP_EHIP;
        } else {
            memset(key_status, 0, nkeys);  // a public key's canonicity is reported per item by the encapsulation (status 1)
        }
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
int circl_hip_mlkem_keytable_new(int param, int private_keys, const uint8_t *keys, size_t nkeys, int device, uint8_t *key_status,
                                 circl_hip_keytable **out) {
    if (out) *out = nullptr;
    if (!kem_k(param) || !keys || !out || nkeys == 0 || nkeys > 0xffffffffull) return CIRCL_HIP_EPARAM;
    // device = CIRCL_HIP_ALL_DEVICES: the same table on every device (the verdicts are the same everywhere: replica 0 reports them)
    return keytable_replicate(device, [&](int dev, circl_hip_keytable **one) {
        return kem_keytable_new_one(param, private_keys, keys, nkeys, dev, dev == 0 || device >= 0 ? key_status : nullptr, one);
    }, out);
}
int circl_hip_mlkem_encaps_table_dev(const circl_hip_keytable *t, const uint32_t *d_key_idx, const uint8_t *d_m, uint8_t *d_ct, uint8_t *d_ss,
                                     uint8_t *d_status, size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    t = keytable_here(t);
    if (!kem_table_ok(t, 0)) return CIRCL_HIP_EPARAM;
    const int param = t->param;
    KEM_DISPATCH(encaps_table_dev_impl<2>(t, d_key_idx, d_m, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 encaps_table_dev_impl<3>(t, d_key_idx, d_m, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 encaps_table_dev_impl<4>(t, d_key_idx, d_m, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st));
}
int circl_hip_mlkem_decaps_table_dev(const circl_hip_keytable *t, const uint32_t *d_key_idx, const uint8_t *d_ct, uint8_t *d_ss, uint8_t *d_status,
                                     size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    t = keytable_here(t);
    if (!kem_table_ok(t, 1)) return CIRCL_HIP_EPARAM;
    const int param = t->param;
    KEM_DISPATCH(decaps_table_dev_impl<2>(t, d_key_idx, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 decaps_table_dev_impl<3>(t, d_key_idx, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st),
                 decaps_table_dev_impl<4>(t, d_key_idx, d_ct, d_ss, d_status, n, d_ws, ws_bytes, st));
}
// host buffers: only the per-item data travel; the keys and what was parsed out of them are already on the table's device
// (key_idx is the one OPTIONAL array: absent = entry 0 for every item; a NULL m / ct is CIRCL_HIP_EPARAM on every path)
static std::vector<HIn> kem_enc_ins(const uint32_t *ki, const uint8_t *m) { return {{reinterpret_cast<const uint8_t *>(ki), size_t(4), false, false, true}, {m, 32, true}}; }
static std::vector<HIn> kem_dec_ins(const uint32_t *ki, const uint8_t *ct, size_t CT) { return {{reinterpret_cast<const uint8_t *>(ki), size_t(4), false, false, true}, {ct, CT}}; }
int circl_hip_mlkem_encaps_table(const circl_hip_keytable *t, const uint32_t *key_idx, const uint8_t *m, uint8_t *ct, uint8_t *ss, uint8_t *status,
                                 size_t n) {
    if (!kem_table_ok(t, 0)) return CIRCL_HIP_EPARAM;
    const size_t CT = circl_hip_mlkem_ct_size(t->param);
    if (n == 0) return CIRCL_HIP_OK;
    if (key_idx)
        if (int rc = check_key_idx(key_idx, n, t->nkeys)) return rc;
    return table_shard(t, n, [&](const circl_hip_keytable *r, size_t lo, size_t cnt) {
        const uint32_t *ki = key_idx ? key_idx + lo : nullptr;
        Coalescer *co = usable_coalescer(r);
        if (co && cnt <= coalescer_call_max(co)) {  // a small call joins the table's cross-caller batch (an absent key_idx: zeros)
            const int rc = coalesce_run(co, cnt, kem_enc_ins(ki, m ? m + lo * 32 : nullptr), {},
                                        {{ct + lo * CT, CT}, {ss + lo * 32, 32, true}, {status ? status + lo : nullptr, 1}}, kem_ws_fn(), kem_opts(), [&](Chunk &c) {
                                            return circl_hip_mlkem_encaps_table_dev(r, reinterpret_cast<const uint32_t *>(c.in[0]), c.in[1], c.out[0], c.out[1], c.out[2],
                                                                                    c.cnt, c.ws, c.ws_bytes, c.st);
                                        });
            if (rc != kNotCoalesced) return rc;
        }
        return run_pipeline(r->device, cnt, {{reinterpret_cast<const uint8_t *>(ki), ki ? size_t(4) : size_t(0)}, {m ? m + lo * 32 : nullptr, 32, true}}, {},
                            {{ct + lo * CT, CT}, {ss + lo * 32, 32, true}, {status ? status + lo : nullptr, 1}}, kem_ws_fn(), kem_opts(), [&](Chunk &c) {
                                return circl_hip_mlkem_encaps_table_dev(r, ki ? reinterpret_cast<const uint32_t *>(c.in[0]) : nullptr, c.in[1], c.out[0], c.out[1],
                                                                        c.out[2], c.cnt, c.ws, c.ws_bytes, c.st);
                            });
    });
}
int circl_hip_mlkem_decaps_table(const circl_hip_keytable *t, const uint32_t *key_idx, const uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n) {
    if (!kem_table_ok(t, 1)) return CIRCL_HIP_EPARAM;
    const size_t CT = circl_hip_mlkem_ct_size(t->param);
    if (n == 0) return CIRCL_HIP_OK;
    if (key_idx)
        if (int rc = check_key_idx(key_idx, n, t->nkeys)) return rc;
    return table_shard(t, n, [&](const circl_hip_keytable *r, size_t lo, size_t cnt) {
        const uint32_t *ki = key_idx ? key_idx + lo : nullptr;
        Coalescer *co = usable_coalescer(r);
        if (co && cnt <= coalescer_call_max(co)) {
            const int rc = coalesce_run(co, cnt, kem_dec_ins(ki, ct ? ct + lo * CT : nullptr, CT), {},
                                        {{ss + lo * 32, 32, true}, {status ? status + lo : nullptr, 1}}, kem_ws_fn(), kem_opts(), [&](Chunk &c) {
                                            return circl_hip_mlkem_decaps_table_dev(r, reinterpret_cast<const uint32_t *>(c.in[0]), c.in[1], c.out[0], c.out[1], c.cnt,
                                                                                    c.ws, c.ws_bytes, c.st);
                                        });
            if (rc != kNotCoalesced) return rc;
        }
        return run_pipeline(r->device, cnt, {{reinterpret_cast<const uint8_t *>(ki), ki ? size_t(4) : size_t(0)}, {ct ? ct + lo * CT : nullptr, CT}}, {},
                            {{ss + lo * 32, 32, true}, {status ? status + lo : nullptr, 1}}, kem_ws_fn(), kem_opts(), [&](Chunk &c) {
                                return circl_hip_mlkem_decaps_table_dev(r, ki ? reinterpret_cast<const uint32_t *>(c.in[0]) : nullptr, c.in[1], c.out[0], c.out[1],
                                                                        c.cnt, c.ws, c.ws_bytes, c.st);
                            });
    });
}
// ---- the asynchronous form (include/circl_hip.h: circl_hip_keytable_async_start) ----
int circl_hip_mlkem_encaps_table_submit(const circl_hip_keytable *t, const uint32_t *key_idx, const uint8_t *m, uint8_t *ct, uint8_t *ss, uint8_t *status,
                                        size_t n, uint64_t *ticket) {
    if (ticket) *ticket = 0;
    if (!kem_table_ok(t, 0) || !ticket || (n && (!m || !ct || !ss))) return CIRCL_HIP_EPARAM;
    const size_t CT = circl_hip_mlkem_ct_size(t->param);
    if (n == 0) return CIRCL_HIP_OK;
    if (key_idx)
        if (int rc = check_key_idx(key_idx, n, t->nkeys)) return rc;
    return table_submit(t, ticket, [&](const circl_hip_keytable *, Coalescer *co, uint64_t *seq) {
        return coalesce_submit(co, n, kem_enc_ins(key_idx, m), {}, {{ct, CT}, {ss, 32, true}, {status, 1}}, seq, false);
    });
}
int circl_hip_mlkem_decaps_table_submit(const circl_hip_keytable *t, const uint32_t *key_idx, const uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n,
                                        uint64_t *ticket) {
    if (ticket) *ticket = 0;
    if (!kem_table_ok(t, 1) || !ticket || (n && (!ct || !ss))) return CIRCL_HIP_EPARAM;
    const size_t CT = circl_hip_mlkem_ct_size(t->param);
    if (n == 0) return CIRCL_HIP_OK;
    if (key_idx)
        if (int rc = check_key_idx(key_idx, n, t->nkeys)) return rc;
    return table_submit(t, ticket, [&](const circl_hip_keytable *, Coalescer *co, uint64_t *seq) {
        return coalesce_submit(co, n, kem_dec_ins(key_idx, ct, CT), {}, {{ss, 32, true}, {status, 1}}, seq, false);
    });
}
}  // extern "C"
namespace circl {
namespace host {
// circl_hip_queue for ML-KEM with the key in the call (a TLS 1.3 server encapsulates once per handshake, to the client's ephemeral key:
// kem/mlkem/mlkem768/kyber.go:359-370 behind kem/hybrid/hybrid.go:95-99): the arrays and the launch of circl_hip_mlkem_encaps / _decaps
int kem_call_queue_start(bool decaps, int param, Coalescer *co, bool want_eventfd, QueueShape *sh) {
    const size_t EK = circl_hip_mlkem_ek_size(param), DK = circl_hip_mlkem_dk_size(param), CT = circl_hip_mlkem_ct_size(param);
    if (!EK) return CIRCL_HIP_EPARAM;
    if (!decaps) {
        *sh = QueueShape{EK, 32, CT, 32, false, true};
        return coalescer_async_start(co, {{nullptr, EK}, {nullptr, 32, true}}, {}, {{nullptr, CT}, {nullptr, 32, true}, {nullptr, 1}}, kem_ws_fn(), kem_opts(), [param](Chunk &c) {
            return circl_hip_mlkem_encaps_dev(param, c.in[0], c.in[1], c.out[0], c.out[1], c.out[2], c.cnt, c.ws, c.ws_bytes, c.st);
        }, want_eventfd);
    }
    *sh = QueueShape{DK, CT, 0, 32, true, false};
    return coalescer_async_start(co, {{nullptr, DK, true}, {nullptr, CT}}, {}, {{nullptr, 32, true}, {nullptr, 1}}, kem_ws_fn(), kem_opts(), [param](Chunk &c) {
        return circl_hip_mlkem_decaps_dev(param, c.in[0], c.in[1], c.out[0], c.out[1], c.cnt, c.ws, c.ws_bytes, c.st);
    }, want_eventfd);
}
// the queue of one ML-KEM table (part): its arrays and its launch, fixed for the queue's life (`r` outlives the queue: the table owns it)
int kem_table_async_start(const circl_hip_keytable *r, Coalescer *co, bool want_eventfd) {
    const size_t CT = circl_hip_mlkem_ct_size(r->param);
    if (!r->private_keys)
        return coalescer_async_start(co, kem_enc_ins(nullptr, nullptr), {}, {{nullptr, CT}, {nullptr, 32, true}, {nullptr, 1}}, kem_ws_fn(), kem_opts(), [r](Chunk &c) {
            return circl_hip_mlkem_encaps_table_dev(r, reinterpret_cast<const uint32_t *>(c.in[0]), c.in[1], c.out[0], c.out[1], c.out[2], c.cnt, c.ws, c.ws_bytes, c.st);
        }, want_eventfd);
    return coalescer_async_start(co, kem_dec_ins(nullptr, nullptr, CT), {}, {{nullptr, 32, true}, {nullptr, 1}}, kem_ws_fn(), kem_opts(), [r](Chunk &c) {
        return circl_hip_mlkem_decaps_table_dev(r, reinterpret_cast<const uint32_t *>(c.in[0]), c.in[1], c.out[0], c.out[1], c.cnt, c.ws, c.ws_bytes, c.st);
    }, want_eventfd);
}
}  // namespace host
}  // namespace circl
extern "C" {
/* PrivateKey.Public() over a batch (kem/mlkem/mlkem768/kyber.go:323-328): the encapsulation key stored inside each decapsulation
 * key (dk = s || ek || H(ek) || z, :189-201).  No device work: a strided copy. */
int circl_hip_mlkem_public_from_private(int param, const uint8_t *dk, uint8_t *ek, size_t n) {
    const int K = kem_k(param);
    if (!K || (n && (!dk || !ek))) return CIRCL_HIP_EPARAM;
    const size_t DK = circl_hip_mlkem_dk_size(param), EK = circl_hip_mlkem_ek_size(param);
    for (size_t i = 0; i < n; i++) memcpy(ek + i * EK, dk + i * DK + 384 * (size_t)K, EK);
    return CIRCL_HIP_OK;
}

// ---- round-3 Kyber (kem/kyber/kyber{512,768,1024}), SURVEY 8f row f3 ------------------------------------

int circl_hip_kyber_keygen_dev(int param, const uint8_t *d_seed64, uint8_t *d_ek, uint8_t *d_dk, size_t n, void *d_ws, size_t ws_bytes,
                               void *stream) {
    KEM_DISPATCH((keygen_dev_impl<2, true>(d_seed64, d_ek, d_dk, n, d_ws, ws_bytes, st)),
                 (keygen_dev_impl<3, true>(d_seed64, d_ek, d_dk, n, d_ws, ws_bytes, st)),
                 (keygen_dev_impl<4, true>(d_seed64, d_ek, d_dk, n, d_ws, ws_bytes, st)));
}
int circl_hip_kyber_encaps_dev(int param, const uint8_t *d_ek, const uint8_t *d_seed32, uint8_t *d_ct, uint8_t *d_ss, size_t n, void *d_ws,
                               size_t ws_bytes, void *stream) {
    KEM_DISPATCH((encaps_dev_impl<2, true>(d_ek, d_seed32, d_ct, d_ss, nullptr, n, d_ws, ws_bytes, st)),
                 (encaps_dev_impl<3, true>(d_ek, d_seed32, d_ct, d_ss, nullptr, n, d_ws, ws_bytes, st)),
                 (encaps_dev_impl<4, true>(d_ek, d_seed32, d_ct, d_ss, nullptr, n, d_ws, ws_bytes, st)));
}
int circl_hip_kyber_decaps_dev(int param, const uint8_t *d_dk, const uint8_t *d_ct, uint8_t *d_ss, size_t n, void *d_ws, size_t ws_bytes,
                               void *stream) {
    KEM_DISPATCH((decaps_dev_impl<2, true>(d_dk, d_ct, d_ss, nullptr, n, d_ws, ws_bytes, st)),
                 (decaps_dev_impl<3, true>(d_dk, d_ct, d_ss, nullptr, n, d_ws, ws_bytes, st)),
                 (decaps_dev_impl<4, true>(d_dk, d_ct, d_ss, nullptr, n, d_ws, ws_bytes, st)));
}
int circl_hip_kyber_keygen(int param, const uint8_t *seed64, uint8_t *ek, uint8_t *dk, size_t n, int device) {
    const size_t EK = circl_hip_mlkem_ek_size(param), DK = circl_hip_mlkem_dk_size(param);
    if (!EK) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{seed64 + lo * 64, 64, true}}, {}, {{ek + lo * EK, EK}, {dk + lo * DK, DK, true}}, kem_ws_fn(), kem_opts(),
                            [&](Chunk &c) { return circl_hip_kyber_keygen_dev(param, c.in[0], c.out[0], c.out[1], c.cnt, c.ws, c.ws_bytes, c.st); });
    });
}
int circl_hip_kyber_encaps(int param, const uint8_t *ek, const uint8_t *seed32, uint8_t *ct, uint8_t *ss, size_t n, int device) {
    const size_t EK = circl_hip_mlkem_ek_size(param), CT = circl_hip_mlkem_ct_size(param);
    if (!EK) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{ek + lo * EK, EK}, {seed32 + lo * 32, 32, true}}, {}, {{ct + lo * CT, CT}, {ss + lo * 32, 32, true}}, kem_ws_fn(),
                            kem_opts(),
                            [&](Chunk &c) { return circl_hip_kyber_encaps_dev(param, c.in[0], c.in[1], c.out[0], c.out[1], c.cnt, c.ws, c.ws_bytes, c.st); });
    });
}
int circl_hip_kyber_decaps(int param, const uint8_t *dk, const uint8_t *ct, uint8_t *ss, size_t n, int device) {
    const size_t DK = circl_hip_mlkem_dk_size(param), CT = circl_hip_mlkem_ct_size(param);
    if (!DK) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{dk + lo * DK, DK, true}, {ct + lo * CT, CT}}, {}, {{ss + lo * 32, 32, true}}, kem_ws_fn(), kem_opts(),
                            [&](Chunk &c) { return circl_hip_kyber_decaps_dev(param, c.in[0], c.in[1], c.out[0], c.cnt, c.ws, c.ws_bytes, c.st); });
    });
}

}  // extern "C"
