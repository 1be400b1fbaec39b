// mlkem_kernels.h -- batch ML-KEM kernels for gfx950 (included by circl_hip.hip).
//
// Encapsulation = two launches over the same HBM-resident arrays (hash, encrypt); decapsulation =
// decrypt, decaps-hash, encrypt<REENCRYPT>; key generation = seed-hash, keygen, keygen-finish.
//
//  mlkem_hash_kernel<K>      lane = item.  H(ek) = SHA3-256 over the packed key (9 absorb
//                            blocks for ML-KEM-768), then (K,r) = SHA3-512(m || H(ek))
//                            (kem/mlkem/mlkem768/kyber.go:126-131, :258-260).  Writes the
//                            shared secret K and parks r (32 B / item) in the workspace.
//                            64 independent sponges per wavefront, states in registers.
//
//  mlkem_encrypt_kernel<K>   persistent single-wavefront workgroups pull groups of G = 64 / K^2 items from a
//                            ticket counter:
//    phase A  lane = (item, i, j): SHAKE128(rho || i || j) rejection sampling of the K^2
//             matrix polynomials of G items at once (sample.go:192-236, mat.go:13-74 with
//             transpose=true); accepted coefficients go through a per-lane LDS FIFO into the
//             stream's 512-byte row of the workgroup's global scratch slice (L2 / Infinity Cache
//             resident, reused by every group).  An LDS-resident variant (template flag) is kept
//             for A/B measurements.
//    phase B  lane = (item, nonce): SHAKE256(r || nonce) PRF blocks for the 2K+1 noise
//             polynomials (sample.go:31-95), raw bytes into LDS; its idle lanes adopt the few
//             matrix streams that need a fourth block (sample_matrix_and_prf).
//    phase C  the wave walks its G items; per item it is K-PKE.Encrypt (cpapke.go:137-181)
//             with one polynomial per wavefront: CBD, 3 forward NTTs, K(K+1) lazy MulHat
//             accumulations, K+1 inverse NTTs, compress and bit-pack straight to HBM.
//             It also decodes t-hat from ek and applies UnpackMLKEM's canonical check
//             (cpapke.go:45-55): an item with a coefficient >= q gets status 1 and zeroed
//             outputs.
#pragma once
#include "kyber_dev.h"

// minimum waves per SIMD the scratch-variant kernels are register-allocated for (<= 512 / VGPRs)
// Issue priority of a wavefront of mlkem_encrypt_kernel while it is in its ring phase (0 = the same as the sampling phases'): the ring phase
// is chains of LDS exchanges with a few instructions in between, the sampling phases are Keccak rounds back to back -- see
// CIRCL_DSA_VERIFY_PRIO (mldsa_kernels.h) for the same idea.  Measured (profiles/r06_kem_prio_ab.txt, two libraries alternating on one box,
// three rounds): mlkem_encrypt_kernel<3> 6.31 -> 6.14 ms per 2^20, the headline 1.411 -> 1.438e8 encapsulations/s (+1.9 %).
#ifndef CIRCL_KEM_RING_PRIO
#define CIRCL_KEM_RING_PRIO 1
#endif
#ifndef CIRCL_KEM_WAVES_PER_EU
#define CIRCL_KEM_WAVES_PER_EU 4
#endif

namespace circl {
namespace mlkem {

using kyber::Q;

template <int K> struct Params;
template <> struct Params<2> { static constexpr int ETA1 = 3, DU = 10, DV = 4; };
template <> struct Params<3> { static constexpr int ETA1 = 2, DU = 10, DV = 4; };
template <> struct Params<4> { static constexpr int ETA1 = 2, DU = 11, DV = 5; };

template <int K> struct Geom {
    using P = Params<K>;
    static constexpr int EK = 384 * K + 32;
    static constexpr int DK = 768 * K + 96;
    static constexpr int CT = 32 * (P::DU * K + P::DV);
    static constexpr int PAIRS = K * K;
    static constexpr int G = 64 / PAIRS;                 // items per workgroup
    static constexpr int A_STREAMS = G * PAIRS;          // <= 64
    static constexpr int NOISE = 2 * K + 1;              // PRF streams per item (encrypt); keygen uses 2K
    static constexpr int A_STRIDE = 520;                 // bytes per sampled polynomial (+1 spill slot, 8-B aligned)
    static constexpr int NOISE_BYTES = 64 * P::ETA1;     // eta1 stream length (eta2 streams use 128)
    static constexpr int NOISE_STRIDE = NOISE_BYTES + 8; // breaks the power-of-two bank stride
    static constexpr int LDS_A = A_STREAMS * A_STRIDE;
    // ML-KEM-512 packs 16 items into a group (64 / K^2) and its eta1 = 3 streams are 192 bytes long: the noise of a whole
    // group would be 16 KB of LDS (2.25 waves per SIMD).  Its PRF therefore runs in two halves of GH = 8 items, each
    // consumed by phase C before the next is produced; the other parameter sets do the whole group at once.
    static constexpr int HALVES = K == 2 ? 2 : 1;
    static constexpr int GH = G / HALVES;
    static constexpr int LDS_NOISE = GH * NOISE * NOISE_STRIDE;
    static constexpr int LDS_XCH = 1024;                 // 256 x 32-bit: the widest exchange of the inverse transform
    static constexpr int LDS_TOTAL = LDS_A + LDS_NOISE + LDS_XCH;
    // "scratch" variant: the sampled matrix goes through a per-workgroup global scratch (L2 / Infinity
    // Cache resident) instead of LDS, which cuts LDS per wave from ~40 KB to ~7 KB (4 waves per SIMD).
    static constexpr int FIFO_STRIDE = 80;               // 32 int16 slots + 16 B pad per lane
    static constexpr int LDS_FIFO = 64 * FIFO_STRIDE;
    static constexpr int LDS_SCRATCH_TOTAL = (LDS_FIFO > LDS_NOISE + LDS_XCH) ? LDS_FIFO : LDS_NOISE + LDS_XCH;
    static constexpr int SCRATCH_BYTES = 64 * 512;       // per resident workgroup
    // shared-key encapsulation (one ek for the whole batch: the shape of the reference's BenchmarkEncapsulate,
    // kem/schemes/schemes_test.go:28-38): the matrix is sampled once per workgroup, a group is GS items whose
    // 2K+1 PRF streams fill one 64-lane pass
    static constexpr int GS = K == 2 ? 8 : 64 / NOISE;
    static constexpr int LDS_NOISE_SHARED = GS * NOISE * NOISE_STRIDE;
    static constexpr int LDS_SHARED_TOTAL = (LDS_FIFO > LDS_NOISE_SHARED + LDS_XCH) ? LDS_FIFO : LDS_NOISE_SHARED + LDS_XCH;
};

// ---- little helpers -------------------------------------------------------------------------

template <int FIRST, int COUNT> __device__ __forceinline__ void xor_words(KeccakState &s, const uint64_t *p) {
    detail::static_for<0, COUNT>([&](auto ic) {
        constexpr int i = decltype(ic)::v;
        const uint64_t w = p[i];
        s.lo[FIRST + i] ^= (uint32_t)w;
        s.hi[FIRST + i] ^= (uint32_t)(w >> 32);
    });
}

// SHA3-256 of `NWORDS` 64-bit words at p (any NWORDS): rate 17 words, ds 0x06.
template <int NWORDS> __device__ __forceinline__ void sha3_256_words(KeccakState &s, const uint64_t *p) {
    constexpr int FULL = NWORDS / 17, REM = NWORDS % 17;
    keccak_zero(s);
#pragma unroll 1
    for (int b = 0; b < FULL; b++) {
        xor_words<0, 17>(s, p + 17 * b);
        keccak_f1600(s);
    }
    xor_words<0, REM>(s, p + 17 * FULL);
    s.lo[REM] ^= kDsSha3;          // REM < 17 for all three key sizes
    s.hi[16] ^= 0x80000000u;
    keccak_f1600(s);
}

// (K, r) = G(m || h) = SHA3-512 over 64 bytes: one block of rate 9 words.
__device__ __forceinline__ void sha3_512_m_h(KeccakState &g, const uint64_t *m, const KeccakState &h) {
    keccak_zero(g);
    xor_words<0, 4>(g, m);
#pragma unroll
    for (int i = 0; i < 4; i++) { g.lo[4 + i] = h.lo[i]; g.hi[4 + i] = h.hi[i]; }
    g.lo[8] = kDsSha3;
    g.hi[8] = 0x80000000u;
    keccak_f1600(g);
}

// ---- the same sponges with a state on two adjacent lanes (keccak_f1600_split): the even lane of a pair holds and loads the
// low dword of every 64-bit word, the odd lane the high one -- `p32` below is the item's data as dwords, already offset by the
// lane's parity.  The next block's words are requested before the permutation of the current one.
template <int NWORDS, class Src> __device__ __forceinline__ void split_sponge17_src(SplitState &s, Src &&src, uint32_t ds, bool hi_lane) {
    constexpr int FULL = NWORDS / 17, REM = NWORDS % 17;  // src(k) = this lane's dword of 64-bit word k
    uint32_t nx[17];
#pragma unroll
    for (int j = 0; j < 25; j++) s.w[j] = 0;
#pragma unroll
    for (int j = 0; j < 17; j++) nx[j] = (FULL > 0 || j < REM) ? src(j) : 0;
#pragma unroll 1
    for (int b = 0; b < FULL; b++) {
#pragma unroll
        for (int j = 0; j < 17; j++) s.w[j] ^= nx[j];
        const bool last = b + 1 == FULL;  // the block behind the last full one has REM words
#pragma unroll
        for (int j = 0; j < 17; j++) nx[j] = (!last || j < REM) ? src(17 * (b + 1) + j) : 0;
        keccak_f1600_split(s, hi_lane);
    }
#pragma unroll
    for (int j = 0; j < 17; j++) s.w[j] ^= nx[j];
    s.w[REM] ^= hi_lane ? 0u : ds;  // REM < 17 for every input size used
    s.w[16] ^= hi_lane ? 0x80000000u : 0u;
    keccak_f1600_split(s, hi_lane);
}
template <int NWORDS> __device__ __forceinline__ void split_sponge17(SplitState &s, const uint32_t *p32, uint32_t ds, bool hi_lane) {
    split_sponge17_src<NWORDS>(s, [&](int k) { return p32[2 * k]; }, ds, hi_lane);
}
// (K, r) = G(m || h) on a lane pair: h = this lane's dwords of the four hash words; K -> ss, r -> r_ws (kyber.go:163-181)
__device__ __forceinline__ void mlkem_g_split(const uint32_t (&h)[4], const uint8_t *__restrict__ m, uint8_t *__restrict__ ss, uint8_t *__restrict__ r_ws,
                                              size_t item, bool live, int parity) {
    const bool hi_lane = parity != 0;
    const uint32_t *mw = reinterpret_cast<const uint32_t *>(m + item * 32) + parity;
    SplitState s;
#pragma unroll
    for (int j = 0; j < 4; j++) { s.w[j] = mw[2 * j]; s.w[4 + j] = h[j]; }
    s.w[8] = hi_lane ? 0x80000000u : kDsSha3;  // one block of rate 9 words: suffix and final bit both in word 8
#pragma unroll
    for (int j = 9; j < 25; j++) s.w[j] = 0;
    keccak_f1600_split(s, hi_lane);
    if (live) {
        uint32_t *kd = reinterpret_cast<uint32_t *>(ss + item * 32) + parity, *rd = reinterpret_cast<uint32_t *>(r_ws + item * 32) + parity;
#pragma unroll
        for (int j = 0; j < 4; j++) { kd[2 * j] = s.w[j]; rd[2 * j] = s.w[4 + j]; }
    }
}
// H(ek), then G, item `item` on this lane pair
template <int K>
__device__ __forceinline__ void mlkem_hash_split(const uint8_t *__restrict__ ek, const uint8_t *__restrict__ m, uint8_t *__restrict__ ss,
                                                 uint8_t *__restrict__ r_ws, size_t item, bool live, int parity) {
    using Gm = Geom<K>;
    SplitState s;
    split_sponge17<Gm::EK / 8>(s, reinterpret_cast<const uint32_t *>(ek + item * Gm::EK) + parity, kDsSha3, parity != 0);
    const uint32_t h[4] = {s.w[0], s.w[1], s.w[2], s.w[3]};
    mlkem_g_split(h, m, ss, r_ws, item, live, parity);
}

template <int FIRST, int COUNT> __device__ __forceinline__ void store_words(uint64_t *p, const KeccakState &s) {
    detail::static_for<0, COUNT>([&](auto ic) {
        constexpr int i = decltype(ic)::v;
        p[i] = ((uint64_t)s.hi[FIRST + i] << 32) | s.lo[FIRST + i];
    });
}

// A cooperative SHA3-256 / SHAKE256 sponge (rate 17 words) over `nwords` 64-bit words delivered by `src(k)`; returns with the
// squeezed state in the lanes (word j in lane j of the half).
template <bool NW = false, class Src>
__device__ __forceinline__ void coop_sponge17(uint32_t &vlo, uint32_t &vhi, Src &&src, int nwords, uint32_t ds, const CoopLane &c, int j) {
    const int full = nwords / 17, rem = nwords % 17;
    vlo = vhi = 0;
    uint64_t next = j < 17 ? src(j) : 0;  // the block after the current one is requested before the permutation
#pragma unroll 1
    for (int b = 0; b < full; b++) {
        vlo ^= (uint32_t)next;
        vhi ^= (uint32_t)(next >> 32);
        next = (j < 17 && 17 * (b + 1) + j < nwords) ? src(17 * (b + 1) + j) : 0;
        keccak_f1600_coop2<NW>(vlo, vhi, c);
    }
    vlo ^= (uint32_t)next;
    vhi ^= (uint32_t)(next >> 32);
    if (j == rem) vlo ^= ds;
    if (j == 16) vhi ^= 0x80000000u;
    keccak_f1600_coop2<NW>(vlo, vhi, c);
}

// ---- kernel 1: H(ek), G(m || H(ek)) -----------------------------------------------------------

template <int K>
__global__ void __launch_bounds__(256) mlkem_hash_kernel(const uint8_t *__restrict__ ek, const uint8_t *__restrict__ m,
                                                         uint8_t *__restrict__ ss, uint8_t *__restrict__ r_ws, size_t n) {
    using Gm = Geom<K>;
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < n;
    if (!live) idx = n - 1;  // keep the wave converged; the duplicate result is not stored
    KeccakState h, g;
    sha3_256_words<Gm::EK / 8>(h, reinterpret_cast<const uint64_t *>(ek + idx * Gm::EK));
    sha3_512_m_h(g, reinterpret_cast<const uint64_t *>(m + idx * 32), h);
    if (live) {
        store_words<0, 4>(reinterpret_cast<uint64_t *>(ss + idx * 32), g);
        store_words<4, 4>(reinterpret_cast<uint64_t *>(r_ws + idx * 32), g);
    }
}

// ---- shared-key decapsulation: the private key's hash check (kyber.go:219-228), once -------------------
template <int K>
__global__ void __launch_bounds__(64) mlkem_dk_check_kernel(const uint8_t *__restrict__ dk, uint8_t *__restrict__ key_status) {
    __shared__ uint64_t ws[100];
    const int lane = threadIdx.x, j = lane & 31;
    const CoopLane c = coop_lane(ws, lane);
    const uint64_t *ekw = reinterpret_cast<const uint64_t *>(dk + 384 * K);
    const uint64_t *stored = reinterpret_cast<const uint64_t *>(dk + 768 * K + 32);
    uint32_t vlo, vhi;
    coop_sponge17(vlo, vhi, [&](int k) { return ekw[k]; }, Geom<K>::EK / 8, kDsSha3, c, j);
    const bool mine = j >= 4 || ((((uint64_t)vhi << 32) | vlo) == stored[j & 3]);
    const unsigned long long bad = __ballot(!mine);  // (both halves carry the same state)
    if (lane == 0) *key_status = (bad & 0xfull) == 0 ? 0 : 2;
}

// ---- shared-key encapsulation: H(ek) once, then (K, r) = G(m || H(ek)) per item ------------------

template <int K>
__global__ void __launch_bounds__(64) mlkem_hek_kernel(const uint8_t *__restrict__ ek, uint8_t *__restrict__ h_ws) {
    // one hash per call, in front of everything else: on the cooperative permutation (~36 us instead of ~97 for a lone lane)
    __shared__ uint64_t ws[100];
    const int lane = threadIdx.x, j = lane & 31;
    const CoopLane c = coop_lane(ws, lane);
    const uint64_t *ekw = reinterpret_cast<const uint64_t *>(ek);
    uint32_t vlo, vhi;
    coop_sponge17(vlo, vhi, [&](int k) { return ekw[k]; }, Geom<K>::EK / 8, kDsSha3, c, j);
    if (lane < 4) reinterpret_cast<uint64_t *>(h_ws)[lane] = ((uint64_t)vhi << 32) | vlo;
}

// ---- key tables (grouped keys): what the reference caches in a parsed key object, once per table entry ----------
// The reference keeps A^T and H(ek) in the parsed PublicKey / PrivateKey (kem/mlkem/mlkem768/kyber.go:39-43, :247-263;
// pke/kyber/kyber768/internal/cpapke.go:19-25), so a batch over a handful of distinct keys pays matrix expansion once
// per key.  lane = table entry: H(ek) of the key at keys + j * stride + ek_off -> h_out[j]; for private keys
// (stored != 0) the hash is also compared with the one stored behind the embedded ek -> key_status[j] = 0 | 2.
template <int K>
__global__ void __launch_bounds__(256) mlkem_hek_table_kernel(const uint8_t *__restrict__ keys, size_t stride, size_t ek_off,
                                                              uint8_t *__restrict__ h_out, uint8_t *__restrict__ key_status, int stored,
                                                              size_t nkeys) {
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = j < nkeys;
    if (!live) j = nkeys - 1;
    const uint8_t *ekp = keys + j * stride + ek_off;
    KeccakState h;
    sha3_256_words<Geom<K>::EK / 8>(h, reinterpret_cast<const uint64_t *>(ekp));
    if (!live) return;
    store_words<0, 4>(reinterpret_cast<uint64_t *>(h_out + j * 32), h);
    if (stored) {
        const uint64_t *st = reinterpret_cast<const uint64_t *>(ekp + Geom<K>::EK);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 4; i++) ok &= (((uint64_t)h.hi[i] << 32) | h.lo[i]) == st[i];
        key_status[j] = ok ? 0 : 2;
    }
}
// key_idx != nullptr (key-table batches): item idx uses the hash of table entry key_idx[idx].
static __global__ void __launch_bounds__(256) mlkem_g_shared_kernel(const uint8_t *__restrict__ h_ws, const uint8_t *__restrict__ m,
                                                             uint8_t *__restrict__ ss, uint8_t *__restrict__ r_ws, size_t n,
                                                             const KeyIdx key_idx) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    KeccakState g;
    keccak_zero(g);
    xor_words<0, 4>(g, reinterpret_cast<const uint64_t *>(m + idx * 32));
    xor_words<4, 4>(g, reinterpret_cast<const uint64_t *>(h_ws + (key_idx ? (size_t)key_idx[idx] * 32 : 0)));
    g.lo[8] = kDsSha3;
    g.hi[8] = 0x80000000u;
    keccak_f1600(g);
    store_words<0, 4>(reinterpret_cast<uint64_t *>(ss + idx * 32), g);
    store_words<4, 4>(reinterpret_cast<uint64_t *>(r_ws + idx * 32), g);
}

// ---- round-3 Kyber hashing (kem/kyber/kyber768/kyber.go), SURVEY 8f row f3 ---------------------

// lane = item: m = H(seed) ("the hash of shame", kyber.go:131-135), (K', r) = G(m || H(pk)) (:137-142).
// K' -> ss (replaced by kyber_r3_finish_kernel once the ciphertext exists), r and m -> workspace.
template <int K>
__global__ void __launch_bounds__(256) kyber_r3_hash_kernel(const uint8_t *__restrict__ ek, const uint8_t *__restrict__ seed,
                                                            uint8_t *__restrict__ ss, uint8_t *__restrict__ r_ws,
                                                            uint8_t *__restrict__ m_ws, size_t n) {
    using Gm = Geom<K>;
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < n;
    if (!live) idx = n - 1;
    KeccakState h, g;
    sha3_256_words<4>(g, reinterpret_cast<const uint64_t *>(seed + idx * 32));  // m in words 0..3 of g
    sha3_256_words<Gm::EK / 8>(h, reinterpret_cast<const uint64_t *>(ek + idx * Gm::EK));
    if (live) store_words<0, 4>(reinterpret_cast<uint64_t *>(m_ws + idx * 32), g);
#pragma unroll
    for (int i = 0; i < 4; i++) { g.lo[4 + i] = h.lo[i]; g.hi[4 + i] = h.hi[i]; }
#pragma unroll
    for (int i = 8; i < 25; i++) { g.lo[i] = 0; g.hi[i] = 0; }
    g.lo[8] = kDsSha3;
    g.hi[8] = 0x80000000u;
    keccak_f1600(g);
    if (live) {
        store_words<0, 4>(reinterpret_cast<uint64_t *>(ss + idx * 32), g);
        store_words<4, 4>(reinterpret_cast<uint64_t *>(r_ws + idx * 32), g);
    }
}

// lane = item: ss = KDF(ss || H(ct)) = SHAKE256(ss || SHA3-256(ct))[:32]  (kyber.go:147-154, :179-196)
template <int K>
__global__ void __launch_bounds__(256) kyber_r3_finish_kernel(const uint8_t *__restrict__ ct, uint8_t *__restrict__ ss, size_t n) {
    using Gm = Geom<K>;
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < n;
    if (!live) idx = n - 1;
    KeccakState h, g;
    sha3_256_words<Gm::CT / 8>(h, reinterpret_cast<const uint64_t *>(ct + idx * Gm::CT));
    keccak_zero(g);
    xor_words<0, 4>(g, reinterpret_cast<const uint64_t *>(ss + idx * 32));
#pragma unroll
    for (int i = 0; i < 4; i++) { g.lo[4 + i] = h.lo[i]; g.hi[4 + i] = h.hi[i]; }
    g.lo[8] = kDsShake;
    g.hi[16] = 0x80000000u;
    keccak_f1600(g);
    if (live) store_words<0, 4>(reinterpret_cast<uint64_t *>(ss + idx * 32), g);
}

// lane = item: (K'', r') = G(m' || H(pk) as stored in the private key) (kyber.go:168-173); the re-encryption
// kernel then selects K'' (ct' == ct) or z (:184-189) and kyber_r3_finish_kernel applies the KDF.
template <int K>
__global__ void __launch_bounds__(256) kyber_r3_decaps_hash_kernel(const uint8_t *__restrict__ dk, const uint8_t *__restrict__ mprime_ws,
                                                                   uint8_t *__restrict__ kbar_ws, uint8_t *__restrict__ r_ws,
                                                                   uint8_t *__restrict__ z_ws, uint8_t *__restrict__ status, size_t n) {
    using Gm = Geom<K>;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const uint8_t *dkp = dk + idx * Gm::DK;
    KeccakState g;
    keccak_zero(g);
    xor_words<0, 4>(g, reinterpret_cast<const uint64_t *>(mprime_ws + idx * 32));
    xor_words<4, 4>(g, reinterpret_cast<const uint64_t *>(dkp + 768 * K + 32));
    g.lo[8] = kDsSha3;
    g.hi[8] = 0x80000000u;
    keccak_f1600(g);
    store_words<0, 4>(reinterpret_cast<uint64_t *>(kbar_ws + idx * 32), g);
    store_words<4, 4>(reinterpret_cast<uint64_t *>(r_ws + idx * 32), g);
    const uint64_t *zw = reinterpret_cast<const uint64_t *>(dkp + 768 * K + 64);
#pragma unroll
    for (int i = 0; i < 4; i++) reinterpret_cast<uint64_t *>(z_ws + idx * 32)[i] = zw[i];
    status[idx] = 0;  // round-3 private keys are not checked (kyber.go:215-232)
}

// ---- phase A: matrix expansion --------------------------------------------------------------

// Parse one squeezed SHAKE128 block (21 words = 56 three-byte groups = 112 candidates, t1 then
// t2 of each group: sample.go:207-228) and append the accepted ones to this lane's polynomial.
// Branch-free: every candidate is stored at slot `cnt`, and cnt only advances on acceptance, so
// a rejected value is overwritten by the next accepted one.  cnt saturates at 256, where slot
// 256 is a spill slot inside the 520-byte stride.
__device__ __forceinline__ void parse_shake128_block(const KeccakState &s, int16_t *poly, int &cnt) {
    detail::static_for<0, 112>([&](auto ic) {
        constexpr int c = decltype(ic)::v;
        constexpr int bit = 12 * c, w = bit / 32, sh = bit % 32;
        // 32-bit word w of the block: word (w/2), half (w%2)
        auto word = [&](int i) -> uint32_t { return (i & 1) ? s.hi[i >> 1] : s.lo[i >> 1]; };
        uint32_t v;
        if constexpr (sh <= 20) v = (word(w) >> sh) & 0xfffu;
        else v = alignbit(word(w + 1), word(w), sh) & 0xfffu;
        poly[cnt] = (int16_t)v;
        cnt = min(cnt + (v < (uint32_t)Q ? 1 : 0), 256);
    });
}

// rho of item t is at rho + t * rho_stride (8-byte aligned).  TRANSPOSED samples stream (i,j) from
// (x=i, y=j) -- the matrix A^T used by Encrypt -- otherwise from (x=j, y=i) (mat.go:13-74).
template <int K, bool TRANSPOSED>
__device__ __forceinline__ void sample_matrix(uint8_t *lds_a, const uint8_t *__restrict__ rho, size_t rho_stride, size_t item0,
                                              size_t n, int lane) {
    using Gm = Geom<K>;
    const bool on = lane < Gm::A_STREAMS;
    const int g = on ? lane / Gm::PAIRS : 0, p = on ? lane % Gm::PAIRS : 0;
    const int i = p / K, j = p % K;
    size_t item = item0 + g;
    if (item >= n) item = n - 1;
    KeccakState s;
    keccak_zero(s);
    // SHAKE128(rho || x=i || y=j): 34 bytes -> words 0..3 = rho, word 4 = i | j<<8 | 0x1f<<16,
    // 0x80 into byte 167 (sample.go:105-119 builds the same first block).
    xor_words<0, 4>(s, reinterpret_cast<const uint64_t *>(rho + item * rho_stride));
    s.lo[4] = (TRANSPOSED ? (uint32_t)i | ((uint32_t)j << 8) : (uint32_t)j | ((uint32_t)i << 8)) | (kDsShake << 16);
    s.hi[20] = 0x80000000u;
    int16_t *poly = reinterpret_cast<int16_t *>(lds_a + (on ? lane : 0) * Gm::A_STRIDE);
    int cnt = on ? 0 : 256;
    // three blocks are needed by every stream; a fourth by 0.83 % of them, more essentially never
#pragma unroll 1
    for (int blk = 0; blk < 3 || __any(cnt < 256); blk++) {
        keccak_f1600(s);
        if (on) parse_shake128_block(s, poly, cnt);
    }
}

// Scratch variant of phase A.  Each lane appends accepted coefficients to a 32-slot LDS FIFO and
// flushes 8 of them (16 bytes) at a time to its 512-byte row of the workgroup's global scratch, so
// every global store is a full 16-byte segment.  Same branch-free acceptance as above.
// TAIL = true is used from the 4th block on, where only the few unfinished streams still matter: the
// wave leaves the block as soon as every stream is complete (checked at the 8-candidate flush points).
template <bool TAIL, int SLOTS = 32>
__device__ __forceinline__ void parse_shake128_block_fifo(const KeccakState &s, int16_t *fifo, int16_t *row, int &cnt, int &flushed) {
    bool live = true;
    detail::static_for<0, 112>([&](auto ic) {
        constexpr int c = decltype(ic)::v;
        constexpr int bit = 12 * c, w = bit / 32, sh = bit % 32;
        auto word = [&](int i) -> uint32_t { return (i & 1) ? s.hi[i >> 1] : s.lo[i >> 1]; };
        if (!TAIL || live) {
            uint32_t v;
            if constexpr (sh <= 20) v = (word(w) >> sh) & 0xfffu;
            else v = alignbit(word(w + 1), word(w), sh) & 0xfffu;
            fifo[cnt & (SLOTS - 1)] = (int16_t)v;
            cnt += v < (uint32_t)Q ? 1 : 0;
            if constexpr (c % 8 == 7) {
                // The count is capped at the row length here, once per 8 candidates, not per candidate: between two checks a
                // finished stream runs at most 8 entries past 256, into FIFO slots that hold nothing pending (fewer than 8
                // pending entries + 8 new ones <= 16 <= SLOTS), and nothing beyond entry 255 is ever flushed.
                cnt = min(cnt, 256);
                if (cnt - flushed >= 8) {  // at most 15 pending here, so one flush per check suffices
                    const uint4 d = *reinterpret_cast<const uint4 *>(fifo + (flushed & (SLOTS - 1)));
                    *reinterpret_cast<uint4 *>(row + flushed) = d;
                    flushed += 8;
                }
                if constexpr (TAIL) live = __any(flushed < 256);  // wave-uniform
            }
        }
    });
}

// XY_ARG (the unit-level primitive circl_hip_kyber_sample_uniform only): lane = item, the stream's (x, y) bytes come from
// xy[2 item], xy[2 item + 1] -- Poly.DeriveUniform(seed, x, y) for arbitrary coordinates, sample.go:192-236.
template <int K, bool TRANSPOSED, int GA = Geom<K>::G, bool XY_ARG = false>
__device__ __forceinline__ void sample_matrix_scratch(uint8_t *lds_fifo, int16_t *rows, const uint8_t *__restrict__ rho,
                                                      size_t rho_stride, size_t item0, size_t n, int lane, const uint8_t *__restrict__ xy = nullptr) {
    using Gm = Geom<K>;
    const bool on = XY_ARG ? item0 + lane < n : lane < GA * Gm::PAIRS;
    const int g = XY_ARG ? lane : on ? lane / Gm::PAIRS : 0, p = on ? lane % Gm::PAIRS : 0;
    const int i = p / K, j = p % K;
    size_t item = item0 + g;
    if (item >= n) item = n - 1;
    KeccakState s;
    keccak_zero(s);
    xor_words<0, 4>(s, reinterpret_cast<const uint64_t *>(rho + item * rho_stride));
    if constexpr (XY_ARG) s.lo[4] = (uint32_t)xy[2 * item] | ((uint32_t)xy[2 * item + 1] << 8) | (kDsShake << 16);
    else s.lo[4] = (TRANSPOSED ? (uint32_t)i | ((uint32_t)j << 8) : (uint32_t)j | ((uint32_t)i << 8)) | (kDsShake << 16);
    s.hi[20] = 0x80000000u;
    int16_t *fifo = reinterpret_cast<int16_t *>(lds_fifo + lane * Gm::FIFO_STRIDE);
    int16_t *row = rows + lane * 256;
    int cnt = on ? 0 : 256, flushed = cnt;
#pragma unroll 1
    for (int blk = 0; blk < 3; blk++) {
        keccak_f1600(s);
        if (on) parse_shake128_block_fifo<false>(s, fifo, row, cnt, flushed);
    }
    // a fourth block is needed by 0.83 % of the streams, more essentially never
#pragma unroll 1
    while (__any(flushed < 256)) {
        keccak_f1600(s);
        parse_shake128_block_fifo<true>(s, fifo, row, cnt, flushed);
    }
}

// First 128 bytes of a PRF block to the stream's LDS slot: as biased CBD nibbles for an eta = 2 stream, raw for eta = 3.
template <bool ETA2> __device__ __forceinline__ void store_prf_words(uint32_t *out, const KeccakState &s) {
    detail::static_for<0, 16>([&](auto ic) {
        constexpr int w = decltype(ic)::v;
        out[2 * w] = ETA2 ? kyber::cbd2_bias8_word(s.lo[w]) : s.lo[w];
        out[2 * w + 1] = ETA2 ? kyber::cbd2_bias8_word(s.hi[w]) : s.hi[w];
    });
}

// Phases A and B of the scratch variant in one routine, so that the stragglers of phase A can ride along
// with phase B.  After three SHAKE128 blocks 0.83 % of the matrix streams are still a few coefficients
// short; running a fourth 64-lane permutation for one or two of them costs 12 % of phase A.  The PRF pass
// of phase B, however, leaves 64 - G*NOISE lanes idle (15 for ML-KEM-768) and runs the very same
// permutation: the unfinished sponge states (plus their few pending FIFO entries) are parked in LDS,
// adopted by those idle lanes, permuted for free together with the PRF states, and then parsed from a
// 16-slot mini FIFO.  With more stragglers than free lanes, or eta1 = 3 (two PRF permutations per pass),
// the plain fourth-block loop runs instead.  A fifth block (p ~ 1e-32 per stream) falls out of the final loop.
template <int K, bool TRANSPOSED, int NOISE, int ETA1_COUNT>
__device__ __forceinline__ void sample_matrix_and_prf(uint8_t *lds_fifo, uint8_t *lds_noise, uint8_t *lds_mini, int16_t *rows,
                                                      const uint8_t *__restrict__ rho, size_t rho_stride,
                                                      const uint8_t *__restrict__ seed, size_t seed_stride, size_t item0, size_t n,
                                                      int lane) {
    using Gm = Geom<K>;
    constexpr int STREAMS = Gm::G * NOISE;
    constexpr int LAST_BASE = ((STREAMS - 1) / 64) * 64, USED_LAST = STREAMS - LAST_BASE, FREE = 64 - USED_LAST;
    constexpr int MAX_HITCH = FREE < 15 ? FREE : 15;  // mini FIFO #15 is the dummy of the other lanes
    constexpr bool CAN_HITCH = Params<K>::ETA1 == 2 && MAX_HITCH > 0;
    constexpr int STASH = 240;                          // bytes per parked stream
    static_assert(15 * STASH <= Gm::LDS_FIFO, "stash fits in the FIFO area");

    // ---- phase A: three blocks for every stream ----
    KeccakState s;
    int cnt, flushed;
    int16_t *fifo = reinterpret_cast<int16_t *>(lds_fifo + lane * Gm::FIFO_STRIDE);
    {
        const bool on = lane < Gm::A_STREAMS;
        const int g = on ? lane / Gm::PAIRS : 0, p = on ? lane % Gm::PAIRS : 0;
        const int i = p / K, j = p % K;
        size_t item = item0 + g;
        if (item >= n) item = n - 1;
        keccak_zero(s);
        xor_words<0, 4>(s, reinterpret_cast<const uint64_t *>(rho + item * rho_stride));
        s.lo[4] = (TRANSPOSED ? (uint32_t)i | ((uint32_t)j << 8) : (uint32_t)j | ((uint32_t)i << 8)) | (kDsShake << 16);
        s.hi[20] = 0x80000000u;
        int16_t *row = rows + lane * 256;
        cnt = on ? 0 : 256;
        flushed = cnt;
#pragma unroll 1
        for (int blk = 0; blk < 3; blk++) {
            keccak_f1600(s);
            if (on) parse_shake128_block_fifo<false>(s, fifo, row, cnt, flushed);
        }
        const unsigned long long smask0 = __ballot(flushed < 256);
        if (!CAN_HITCH || __popcll(smask0) > MAX_HITCH) {
#pragma unroll 1
            while (__any(flushed < 256)) {
                keccak_f1600(s);
                parse_shake128_block_fifo<true>(s, fifo, row, cnt, flushed);
            }
        }
    }
    const unsigned long long smask = __ballot(flushed < 256);  // wave-uniform; empty unless stragglers hitch-hike
    const int nstr = __popcll(smask);
    if (nstr > 0) {
        const bool mine = (smask >> lane) & 1;
        uint4 pend = make_uint4(0, 0, 0, 0);
        if (mine) pend = *reinterpret_cast<const uint4 *>(fifo + (flushed & 31));  // < 8 pending entries, one aligned chunk
        __syncthreads();  // every pending chunk is in registers before the stash overwrites FIFO rows
        if (mine) {
            uint32_t *st = reinterpret_cast<uint32_t *>(lds_fifo + __popcll(smask & ((1ull << lane) - 1)) * STASH);
#pragma unroll
            for (int w = 0; w < 25; w++) { st[2 * w] = s.lo[w]; st[2 * w + 1] = s.hi[w]; }
            st[50] = (uint32_t)cnt; st[51] = (uint32_t)flushed; st[52] = (uint32_t)lane;
            st[53] = pend.x; st[54] = pend.y; st[55] = pend.z; st[56] = pend.w;
        }
    }
    __syncthreads();

    // ---- phase B: PRF streams; the last pass adopts the parked streams on its idle lanes ----
#pragma unroll 1
    for (int base = 0; base < STREAMS; base += 64) {
        const int sidx = base + lane;
        const bool on = sidx < STREAMS;
        const int g = on ? sidx / NOISE : 0, nonce = on ? sidx % NOISE : 0;
        size_t item = item0 + g;
        if (item >= n) item = n - 1;
        keccak_zero(s);
        xor_words<0, 4>(s, reinterpret_cast<const uint64_t *>(seed + item * seed_stride));
        s.lo[4] = (uint32_t)nonce | (kDsShake << 8);
        s.hi[16] = 0x80000000u;
        const bool adopt_pass = CAN_HITCH && nstr > 0 && base == LAST_BASE;
        const int tk = lane - USED_LAST;
        const bool target = adopt_pass && tk >= 0 && tk < nstr;
        int tcnt = 256, tflushed = 256, tsrc = 0;
        uint4 pend = make_uint4(0, 0, 0, 0);
        if (target) {
            const uint32_t *st = reinterpret_cast<const uint32_t *>(lds_fifo + tk * STASH);
#pragma unroll
            for (int w = 0; w < 25; w++) { s.lo[w] = st[2 * w]; s.hi[w] = st[2 * w + 1]; }
            tcnt = (int)st[50]; tflushed = (int)st[51]; tsrc = (int)st[52];
            pend = make_uint4(st[53], st[54], st[55], st[56]);
        }
        keccak_f1600(s);
        uint32_t *out = reinterpret_cast<uint32_t *>(lds_noise + (on ? sidx : 0) * Gm::NOISE_STRIDE);
        if constexpr (Params<K>::ETA1 == 2) {
            if (on) store_prf_words<true>(out, s);
        } else {
            if (on && nonce < ETA1_COUNT) store_prf_words<false>(out, s);
            if (on && nonce >= ETA1_COUNT) store_prf_words<true>(out, s);
        }
        if constexpr (Params<K>::ETA1 == 3) {
            // 192 bytes needed for eta1 = 3 streams: word 16 of this block, then 7 more
            if (on && nonce < ETA1_COUNT) { out[32] = s.lo[16]; out[33] = s.hi[16]; }
            if (__any(on && nonce < ETA1_COUNT)) {
                keccak_f1600(s);
                if (on && nonce < ETA1_COUNT) {
                    detail::static_for<0, 7>([&](auto ic) {
                        constexpr int w = decltype(ic)::v;
                        out[34 + 2 * w] = s.lo[w];
                        out[35 + 2 * w] = s.hi[w];
                    });
                }
            }
        }
        if (adopt_pass) {  // wave-uniform
            int16_t *tf = reinterpret_cast<int16_t *>(lds_mini + (target ? tk : 15) * 32);
            int16_t *trow = rows + tsrc * 256;
            if (target) *reinterpret_cast<uint4 *>(tf + (tflushed & 15)) = pend;
            parse_shake128_block_fifo<true, 16>(s, tf, trow, tcnt, tflushed);
#pragma unroll 1
            while (__any(tflushed < 256)) {
                keccak_f1600(s);
                parse_shake128_block_fifo<true, 16>(s, tf, trow, tcnt, tflushed);
            }
        }
    }
}

// Matrix polynomial accessors for phase C (4 consecutive coefficients of stream `stream`, layout L4).
struct AFromLds {
    const uint8_t *base;
    int stride;
    __device__ __forceinline__ void load(uint32_t &a01, uint32_t &a23, int stream, int lane) const {
        const uint2 w = *reinterpret_cast<const uint2 *>(base + stream * stride + 8 * lane);
        a01 = w.x; a23 = w.y;
    }
};
struct AFromScratch {
    const int16_t *rows;
    __device__ __forceinline__ void load(uint32_t &a01, uint32_t &a23, int stream, int lane) const {
        // agent-scope relaxed load = global_load_dwordx2 sc1: served by L2, never by a stale L1 line
        // left over from the previous group that used this scratch row.  (Plain loads after an L1
        // invalidate -- rows_acquire() below, which the ML-DSA kernels use -- measured 3 % slower here:
        // the invalidate also drops the other waves' ek / noise lines.)
        const uint64_t w = __hip_atomic_load(reinterpret_cast<const uint64_t *>(rows + stream * 256 + 4 * lane), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
        a01 = (uint32_t)w; a23 = (uint32_t)(w >> 32);
    }
};
// Rows of a key table's expanded matrices: written by an earlier launch, read-only here, so ordinary cached loads.
struct AFromCache {
    const int16_t *rows;
    __device__ __forceinline__ void load(uint32_t &a01, uint32_t &a23, int stream, int lane) const {
        const uint2 w = *reinterpret_cast<const uint2 *>(rows + stream * 256 + 4 * lane);
        a01 = w.x; a23 = w.y;
    }
};
// Ends a sampling phase whose rows are read back with plain loads: the wave's row stores have left the CU (L1 is
// write-through; the release orders them), every lane has arrived, and the agent-scope acquire
// invalidates the CU's L1 (buffer_inv sc1).
__device__ __forceinline__ void rows_acquire() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// ---- phase B: PRF ---------------------------------------------------------------------------

// SHAKE256(seed || nonce) -> 128 bytes (eta=2) or 192 bytes (eta=3) written to LDS.  NOISE streams
// per item; the first ETA1_COUNT of them use eta1, the rest eta2 = 2.  The seed of item t is at
// seed + t * seed_stride.
// Items g0 .. g0 + GCOUNT - 1 of the group; stream s of that range goes to slot s of the noise area.
template <int K, int NOISE, int ETA1_COUNT, int GCOUNT = Geom<K>::G>
__device__ __forceinline__ void prf_streams(uint8_t *lds_noise, const uint8_t *__restrict__ seed, size_t seed_stride,
                                            size_t item0, size_t n, int lane, int g0 = 0) {
    using Gm = Geom<K>;
    constexpr int STREAMS = GCOUNT * NOISE;
#pragma unroll 1
    for (int base = 0; base < STREAMS; base += 64) {
        const int sidx = base + lane;
        const bool on = sidx < STREAMS;
        const int g = g0 + (on ? sidx / NOISE : 0), nonce = on ? sidx % NOISE : 0;
        size_t item = item0 + g;
        if (item >= n) item = n - 1;
        KeccakState s;
        keccak_zero(s);
        xor_words<0, 4>(s, reinterpret_cast<const uint64_t *>(seed + item * seed_stride));
        s.lo[4] = (uint32_t)nonce | (kDsShake << 8);
        s.hi[16] = 0x80000000u;
        keccak_f1600(s);
        uint32_t *out = reinterpret_cast<uint32_t *>(lds_noise + (on ? sidx : 0) * Gm::NOISE_STRIDE);
        if constexpr (Params<K>::ETA1 == 2) {
            if (on) store_prf_words<true>(out, s);
        } else {
            if (on && nonce < ETA1_COUNT) store_prf_words<false>(out, s);
            if (on && nonce >= ETA1_COUNT) store_prf_words<true>(out, s);
        }
        if constexpr (Params<K>::ETA1 == 3) {
            // 192 bytes needed for eta1 = 3 streams: word 16 of this block, then 7 more
            if (on && nonce < ETA1_COUNT) { out[32] = s.lo[16]; out[33] = s.hi[16]; }
            if (__any(on && nonce < ETA1_COUNT)) {
                keccak_f1600(s);
                if (on && nonce < ETA1_COUNT) {
                    detail::static_for<0, 7>([&](auto ic) {
                        constexpr int w = decltype(ic)::v;
                        out[34 + 2 * w] = s.lo[w];
                        out[35 + 2 * w] = s.hi[w];
                    });
                }
            }
        }
    }
}

// The same streams with a sponge per lane PAIR (keccak_f1600_split: 120 instead of 180 instructions per round) for a group of
// `count` items whose count * NOISE streams fit the 32 pairs of one pass -- the few-items-per-workgroup groups of a small batch,
// where the lane-per-stream pass would run mostly empty.  Same bytes in the same slots.
template <int K, int NOISE, int ETA1_COUNT>
__device__ __forceinline__ void prf_streams_split(uint8_t *lds_noise, const uint8_t *__restrict__ seed, size_t seed_stride, size_t item0,
                                                  size_t n, int lane, int count) {
    using Gm = Geom<K>;
    const int sidx = lane >> 1, parity = lane & 1;
    const bool on = sidx < count * NOISE;
    const int g = on ? sidx / NOISE : 0, nonce = on ? sidx % NOISE : 0;
    size_t item = item0 + g;
    if (item >= n) item = n - 1;
    const uint32_t *sw = reinterpret_cast<const uint32_t *>(seed + item * seed_stride) + parity;
    SplitState s;
#pragma unroll
    for (int w = 0; w < 25; w++) s.w[w] = w < 4 ? sw[2 * w] : 0u;
    if (parity == 0) s.w[4] = (uint32_t)nonce | (kDsShake << 8);
    else s.w[16] = 0x80000000u;
    keccak_f1600_split(s, parity != 0);
    uint32_t *out = reinterpret_cast<uint32_t *>(lds_noise + (on ? sidx : 0) * Gm::NOISE_STRIDE) + parity;
    const bool eta2 = Params<K>::ETA1 == 2 || nonce >= ETA1_COUNT;
    if (on) {
        detail::static_for<0, 16>([&](auto ic) {
            constexpr int w = decltype(ic)::v;
            out[2 * w] = eta2 ? kyber::cbd2_bias8_word(s.w[w]) : s.w[w];
        });
    }
    if constexpr (Params<K>::ETA1 == 3) {  // 192 bytes for the eta1 = 3 streams: word 16 of this block, then 7 more of the next
        if (on && !eta2) out[32] = s.w[16];
        keccak_f1600_split(s, parity != 0);
        if (on && !eta2) {
            detail::static_for<0, 7>([&](auto ic) {
                constexpr int w = decltype(ic)::v;
                out[34 + 2 * w] = s.w[w];
            });
        }
    }
}

// ---- phase C helpers -------------------------------------------------------------------------

// CBD sample of coefficient n from a PRF stream in LDS (sample.go:31-95), as the non-negative representative
// q + (-eta..eta) that the ring code of kyber_dev.h works on.  eta = 2 streams were already turned into biased nibbles
// by the PRF pass (store_prf_words / kyber::cbd2_bias8_word); eta = 3 streams are the raw bytes.
template <int ETA> __device__ __forceinline__ int cbd_coeff(const uint8_t *buf, int n) {
    if constexpr (ETA == 2) {
        return (int)((buf[n >> 1] >> (4 * (n & 1))) & 15u) + (Q - 8);
    } else {
        const int bit = 6 * n;
        const unsigned two = (unsigned)buf[bit >> 3] | ((unsigned)buf[(bit >> 3) + 1] << 8);
        return kyber::cbd3_from_6bits((two >> (bit & 7)) & 63u) + Q;
    }
}

// Bit-pack 256 D-bit values (uint16 in LDS, standard order) to global memory as 32-bit words
// (poly.go:248-332 byte formulas == little-endian bit stream).  dst is 4-byte aligned.
template <int D> __device__ __forceinline__ void pack_bits_store(uint32_t *dst, const uint16_t *vals, int lane) {
    constexpr int WORDS = 8 * D;
#pragma unroll
    for (int w0 = 0; w0 < WORDS; w0 += 64) {
        const int w = w0 + lane;
        if (w < WORDS) {
            const int lo = 32 * w;
            int c = lo / D;
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < 32 / D + 2; k++, c++) {
                const int sh = c * D - lo;
                if (c < 256 && sh < 32) {
                    const uint32_t v = vals[c];
                    acc |= sh >= 0 ? (v << sh) : (v >> (-sh));
                }
            }
            dst[w] = acc;
        }
    }
}

// Same bit stream, but compared with an existing ciphertext instead of stored (decapsulation's
// ct == ct' test, kyber.go:177-181); returns a per-lane "differs" flag.
template <int D> __device__ __forceinline__ bool pack_bits_differs(const uint32_t *ref, const uint16_t *vals, int lane) {
    constexpr int WORDS = 8 * D;
    bool diff = false;
#pragma unroll
    for (int w0 = 0; w0 < WORDS; w0 += 64) {
        const int w = w0 + lane;
        if (w < WORDS) {
            const int lo = 32 * w;
            int c = lo / D;
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < 32 / D + 2; k++, c++) {
                const int sh = c * D - lo;
                if (c < 256 && sh < 32) {
                    const uint32_t v = vals[c];
                    acc |= sh >= 0 ? (v << sh) : (v >> (-sh));
                }
            }
            diff |= ref[w] != acc;
        }
    }
    return diff;
}

// Lane-local bit packing.  `v[r]` is the D-bit value of coefficient l + 64 r (layout L1).  Field n
// starts at bit D*n = D*l + 64*D*r, so all four fields of a lane share the shift (D*l) mod 32 and
// sit 2*D dwords apart: each is OR-ed into a zeroed LDS staging area with at most two 32-bit
// atomics, then the 8*D dwords of the polynomial are streamed out (or compared) coalesced.
template <int D, bool NW = false> __device__ __forceinline__ void stage_bits_l1(uint32_t *stage, const unsigned (&v)[4], int lane) {
    constexpr int WORDS = 8 * D;
    kyber::wave_sync<NW>();  // previous users of the staging area are done
    for (int w = lane; w < WORDS; w += 64) stage[w] = 0;
    kyber::wave_sync<NW>();
    const int bit = D * lane, w0 = bit >> 5, sh = bit & 31;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        atomicOr(&stage[w0 + 2 * D * r], v[r] << sh);
        if (sh + D > 32) atomicOr(&stage[w0 + 2 * D * r + 1], v[r] >> (32 - sh));
    }
    kyber::wave_sync<NW>();
}
template <int D> __device__ __forceinline__ void store_staged(uint32_t *dst, const uint32_t *stage, int lane, bool zero) {
    constexpr int WORDS = 8 * D;
#pragma unroll
    for (int w0 = 0; w0 < WORDS; w0 += 64) {
        const int w = w0 + lane;
        if (w < WORDS) dst[w] = zero ? 0u : stage[w];
    }
}
// The same comparison with the reference words fetched ahead of time (RefWords<D>::load at the top of the
// polynomial's iteration): the global-load latency then hides behind the MulHat / InvNTT work instead of sitting
// at the end of every polynomial of a re-encryption.
template <int D> struct RefWords {
    static constexpr int WORDS = 8 * D, N = (WORDS + 63) / 64;
    uint32_t w[N];
    __device__ __forceinline__ void load(const uint32_t *ref, int lane) {
#pragma unroll
        for (int k = 0; k < N; k++) w[k] = (64 * k + lane < WORDS) ? ref[64 * k + lane] : 0u;
    }
    __device__ __forceinline__ bool differs(const uint32_t *stage, int lane) const {
        bool diff = false;
#pragma unroll
        for (int k = 0; k < N; k++)
            if (64 * k + lane < WORDS) diff |= w[k] != stage[64 * k + lane];
        return diff;
    }
};
template <int D> __device__ __forceinline__ bool staged_differs(const uint32_t *ref, const uint32_t *stage, int lane) {
    constexpr int WORDS = 8 * D;
    bool diff = false;
#pragma unroll
    for (int w0 = 0; w0 < WORDS; w0 += 64) {
        const int w = w0 + lane;
        if (w < WORDS) diff |= ref[w] != stage[w];
    }
    return diff;
}

// 12-bit decode of 4 consecutive coefficients (layout L4) from a packed polynomial (poly.go:123-129)
__device__ __forceinline__ void unpack12_l4(int (&c)[4], const uint8_t *poly, int lane) {
    const uint16_t *src = reinterpret_cast<const uint16_t *>(poly + 6 * lane);
    const uint32_t h0 = src[0], h1 = src[1], h2 = src[2];
    c[0] = (int)(h0 & 0xfff);
    c[1] = (int)((h0 >> 12) | ((h1 & 0xff) << 4));
    c[2] = (int)((h1 >> 8) | ((h2 & 0xf) << 8));
    c[3] = (int)(h2 >> 4);
}
// 12-bit encode (poly.go:106-117), coefficients in [0,q), layout L4
__device__ __forceinline__ void pack12_l4(uint8_t *poly, const int (&c)[4], int lane) {
    uint16_t *dst = reinterpret_cast<uint16_t *>(poly + 6 * lane);
    dst[0] = (uint16_t)(c[0] | (c[1] << 12));
    dst[1] = (uint16_t)((c[1] >> 4) | (c[2] << 8));
    dst[2] = (uint16_t)((c[2] >> 8) | (c[3] << 4));
}

// D-bit field number n of a little-endian bit stream in global memory (poly.go:170-243 Decompress)
template <int D> __device__ __forceinline__ unsigned get_bits(const uint8_t *p, int n) {
    const int bit = n * D, b = bit >> 3, sh = bit & 7;
    unsigned w = p[b];
    if (sh + D > 8) w |= (unsigned)p[b + 1] << 8;   // never reads past the last byte of the stream
    if (sh + D > 16) w |= (unsigned)p[b + 2] << 16;
    return (w >> sh) & ((1u << D) - 1);
}

// Resident workgroups pull groups of G items from a global ticket counter (zeroed by the host before
// the launch): equal finish times without a static schedule.  work == nullptr means one group per
// workgroup (grid = number of groups).
__device__ __forceinline__ size_t next_group(unsigned *work, int lane, bool first, size_t ngroups) {
    // The first group of a workgroup is its own index, later ones come from the ticket counter (offset by the grid).  A launch
    // with a workgroup per group never touches the counter: ~2000 workgroups hitting one address at the same instant cost a
    // small batch ~40 us (20 ns per same-address atomic, twice per workgroup) -- measured on the small-batch ML-KEM routes.
    if (first) return (size_t)blockIdx.x;                                // wave-uniform
    if (work == nullptr || (size_t)gridDim.x >= ngroups) return ngroups;
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(work, 1u);
    return (size_t)gridDim.x + (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)t);
}

// ---- kernel 2: K-PKE.Encrypt ------------------------------------------------------------------

enum EncryptMode { ENCAPS = 0, REENCRYPT = 1, ENCAPS_LENIENT = 2 };

// ENCAPS   : ek rows of stride EK, canonical check, ciphertext stored, status written.
// REENCRYPT: decapsulation's second half (kyber.go:158-181).  `ek` points at the ek embedded in dk
//            (row stride DK), coefficients >= q are reduced, not rejected (cpapke.go:58-63 Unpack);
//            m = m', r = r' come from the workspace; ct' is compared with ct instead of stored and
//            ss = (ct == ct') ? K' : J(z || ct), both candidates parked in the workspace by
//            mlkem_decaps_hash_kernel.  Items whose status is already non-zero get ss = 0.
// ENCAPS_LENIENT: round-3 Kyber's encapsulation (kem/kyber/kyber768/kyber.go:248-262): as ENCAPS, but
//            coefficients >= q of the key are reduced instead of rejected and ss / status are left alone.
// ABLATE is a profiling aid (tools/ablate.hip): bit 0 skips phase A, bit 1 phase B, bit 2 phase C.
// SCRATCH selects where the sampled matrix lives between phase A and phase C: the workgroup's slice of
// a global scratch (persistent launch: gridDim.x resident workgroups loop over the groups of G items)
// or LDS (one group per workgroup; kept for A/B measurements).
// KM (ENCAPS / REENCRYPT, scratch variant only) = where the key material of an item comes from:
//   KM_ITEM    every item has its own key row (ek_stride = row bytes); A^T is sampled per group of G items.
//   KM_SHARED  every item uses the key at `ek` (ek_stride = 0); A^T is sampled once per workgroup before the group loop
//              and groups are GS items.
//   KM_KEYED   item t uses entry key_idx[t] of a key table at `ek` (row stride ek_stride); A^T of every table entry was
//              expanded beforehand into key_rows (mlkem_expand_keys_kernel: K^2 rows of 256 int16 per entry) and is read
//              with plain cached loads; groups are GS items.  This is the reference's parsed-key cache
//              (kem/mlkem/mlkem768/kyber.go:39-43) for a batch over a handful of distinct keys.
enum KeyMode { KM_ITEM = 0, KM_SHARED = 1, KM_KEYED = 2 };
template <int K, int MODE, int ABLATE_ARG = 0, bool SCRATCH = true, int KM = KM_ITEM>
__global__ void __launch_bounds__(64, SCRATCH ? CIRCL_KEM_WAVES_PER_EU : 1) mlkem_encrypt_kernel(const uint8_t *__restrict__ ek, size_t ek_stride,
                                                          const uint8_t *__restrict__ m, const uint8_t *__restrict__ r_ws,
                                                          uint8_t *__restrict__ ct, uint8_t *__restrict__ ss,
                                                          uint8_t *__restrict__ status, const uint8_t *__restrict__ kbar_ws,
                                                          const uint8_t *__restrict__ ssrej_ws, uint8_t *__restrict__ scratch, unsigned *__restrict__ work, size_t n,
                                                          const KeyIdx key_idx, const int16_t *__restrict__ key_rows) {
    using Gm = Geom<K>;
    using P = Params<K>;
    // ABLATE_ARG bit 3 (8) is not an ablation: the wavefront reads the shader clock at its phase boundaries and leaves, per workgroup, the
    // cycles it spent in { ticket, sampling (matrix + PRF), ring phase } and its item count at key_rows (unused by the per-item form):
    // tools/clocks_kem.hip, profiles/r06_kem_clocks.txt
    constexpr int ABLATE = ABLATE_ARG & 7;
    constexpr bool CLK = (ABLATE_ARG & 8) != 0;
    uint64_t clk[4] = {0, 0, 0, 0}, tprev = 0;
    auto lap = [&](int slot) {
        if constexpr (CLK) {
            const uint64_t t = __builtin_readcyclecounter();
            clk[slot] += t - tprev;
            tprev = t;
        }
    };
    if constexpr (CLK) tprev = __builtin_readcyclecounter();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *lds_a = smem;                                     // LDS variant: matrix buffer; scratch variant: FIFO
    uint8_t *lds_noise = SCRATCH ? smem : smem + Gm::LDS_A;    // the FIFO is dead once phase A is over
    constexpr bool SHARED = KM != KM_ITEM;                      // one PRF pass per group of GS items, no per-group sampling
    static_assert(!SHARED || (SCRATCH && (MODE == ENCAPS || MODE == REENCRYPT) && ABLATE == 0), "shared-key / key-table mode");
    uint8_t *xch = lds_noise + (SHARED ? Gm::LDS_NOISE_SHARED : Gm::LDS_NOISE);
    int16_t *rows = KM == KM_KEYED ? nullptr : reinterpret_cast<int16_t *>(scratch + (size_t)blockIdx.x * Gm::SCRATCH_BYTES);
    const int lane = threadIdx.x;
    constexpr int GI = SHARED ? Gm::GS : Gm::G;  // items per group
    // Shared-key / key-table modes: a launch with at least one workgroup per default group (a small batch on a mostly idle
    // chip) spreads the items over ALL its workgroups -- fewer items per group, down to one: the PRF pass costs the same
    // whether its lanes are full or not, and the ring phases of a group run one after the other.
    const unsigned gi = (SHARED && (size_t)gridDim.x * GI >= n) ? (unsigned)((n + gridDim.x - 1) / gridDim.x) : (unsigned)GI;
    const size_t ngroups = (n + gi - 1) / gi;
    if constexpr (KM == KM_SHARED) {
        sample_matrix_scratch<K, true, 1>(lds_a, rows, ek + 384 * K, 0, 0, 1, lane);  // rows 0 .. K^2 - 1, once
        __threadfence_block();
        __syncthreads();
    }

#pragma unroll 1
  for (size_t grp = next_group(work, lane, true, ngroups); grp < ngroups; grp = next_group(work, lane, false, ngroups)) {
    const size_t item0 = grp * gi;
    lap(0);
    if constexpr (CIRCL_KEM_RING_PRIO != 0) __builtin_amdgcn_s_setprio(0);
    if constexpr (SHARED) {
        __syncthreads();  // phase C of the previous group is done with the noise
        if (gi * Gm::NOISE <= 32) prf_streams_split<K, Gm::NOISE, K>(lds_noise, r_ws, 32, item0, n, lane, (int)gi);  // (wave-uniform)
        else prf_streams<K, Gm::NOISE, K, Gm::GS>(lds_noise, r_ws, 32, item0, n, lane);
        __syncthreads();
    } else if constexpr (SCRATCH && ABLATE == 0 && Gm::HALVES == 1) {
        __syncthreads();  // phase C of the previous group is done with the LDS the FIFO aliases
        sample_matrix_and_prf<K, true, Gm::NOISE, K>(lds_a, lds_noise, xch, rows, ek + 384 * K, ek_stride,
                                                     r_ws, 32, item0, n, lane);
        __threadfence_block();  // the rows are in L2 before anybody loads them
        __syncthreads();
    } else {
        if (!(ABLATE & 1)) {
            if constexpr (SCRATCH) {
                __syncthreads();
                sample_matrix_scratch<K, true>(lds_a, rows, ek + 384 * K, ek_stride, item0, n, lane);
                __threadfence_block();
            } else {
                sample_matrix<K, true>(lds_a, ek + 384 * K, ek_stride, item0, n, lane);
            }
        }
        __syncthreads();
        if constexpr (Gm::HALVES == 1) {
            if (!(ABLATE & 2)) prf_streams<K, Gm::NOISE, K>(lds_noise, r_ws, 32, item0, n, lane);
            __syncthreads();
        }
    }

    // The 15 per-lane twiddle constants are only needed by the ring phase: they are (re)loaded here, behind an opaque copy
    // of the lane index, so that they are not live (or spilled) across the sampling phases' 100-register Keccak rounds.
    int lane_ring = lane;
    asm volatile("" : "+v"(lane_ring));
    if constexpr (CIRCL_KEM_RING_PRIO != 0) __builtin_amdgcn_s_setprio(CIRCL_KEM_RING_PRIO);
    lap(1);
    const kyber::LaneZetas z = kyber::load_lane_zetas(lane_ring);
#pragma unroll 1
    for (int g = 0; g < ((ABLATE & 4) ? 0 : (int)gi); g++) {
        const size_t item = item0 + g;
        if (item >= n) break;  // wave-uniform
        if constexpr (Gm::HALVES > 1 && !SHARED) {
            if (g % Gm::GH == 0) {  // this half's PRF streams replace the consumed ones of the previous half
                __syncthreads();
                if (!(ABLATE & 2)) prf_streams<K, Gm::NOISE, K, Gm::GH>(lds_noise, r_ws, 32, item0, n, lane, g);
                __syncthreads();
            }
        }
        // wave-uniform; a key table without an index vector is the batch's own keys -- or, with stride 0, ONE key (entry 0)
        const size_t kq = KM == KM_KEYED ? (key_idx ? (size_t)key_idx[item] : (ek_stride ? item : size_t(0))) : item;
        const uint8_t *ekp = ek + kq * ek_stride;
        const int16_t *krows = KM == KM_KEYED ? key_rows + kq * (size_t)(K * K * 256) : nullptr;
        const uint8_t *noise = lds_noise + (SHARED ? g : g % Gm::GH) * Gm::NOISE * Gm::NOISE_STRIDE;

        // t-hat (12-bit codec) in layout L4; ENCAPS applies UnpackMLKEM's range check
        int th[K][4];
        bool bad = false;
#pragma unroll
        for (int j = 0; j < K; j++) {
            unpack12_l4(th[j], ekp + 384 * j, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (MODE == ENCAPS) bad |= th[j][r] >= Q;
                else th[j][r] = kyber::csubq(th[j][r]);  // 12-bit value < 2q: Normalize == csubq
            }
        }
        const bool reject = MODE == ENCAPS ? __any(bad) : false;  // ENCAPS_LENIENT and REENCRYPT reduce instead

        // r-hat = NTT(CBD_eta1(PRF(r, j))), layout L4, left lazy (< 8q + 4 < 2^15: the reference Barrett-reduces here,
        // cpapke.go:142-144, but any representative gives the same ciphertext)
        int rh[K][4];
#pragma unroll
        for (int j = 0; j < K; j++) {
#pragma unroll
            for (int r = 0; r < 4; r++) rh[j][r] = cbd_coeff<P::ETA1>(noise + j * Gm::NOISE_STRIDE, kyber::idx_l1(lane, r));
            kyber::ntt(rh[j], z, xch, lane);
        }
        kyber::HatOperand rop[K];
#pragma unroll
        for (int j = 0; j < K; j++) rop[j] = kyber::hat_prepare(rh[j], z.f6, z.f6n);

        uint8_t *ctp = ct + item * Gm::CT;
        bool differs = false;
        // u[i] = InvNTT(sum_j A^T[i][j] * r-hat[j]) + e1[i]  (cpapke.go:150-164), compressed to du bits
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            uint32_t *dst = reinterpret_cast<uint32_t *>(ctp + 32 * P::DU * i);
            RefWords<P::DU> ref;
            if (MODE == REENCRYPT) ref.load(dst, lane);
            int acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < K; j++) {
                uint32_t a01, a23;
                if constexpr (KM == KM_KEYED) AFromCache{krows}.load(a01, a23, i * K + j, lane);
                else if constexpr (SCRATCH) AFromScratch{rows}.load(a01, a23, SHARED ? i * K + j : (g * K + i) * K + j, lane);
                else AFromLds{lds_a, Gm::A_STRIDE}.load(a01, a23, (g * K + i) * K + j, lane);
                kyber::mulhat_acc_packed(acc, a01, a23, rop[j]);
            }
            kyber::mulhat_finish(acc);
            kyber::invntt<kyber::NEG_R32>(acc, z, xch, lane);  // undoes mulhat_finish's -2^-32: plain coefficients in [0, q)
            const uint8_t *e1 = noise + (K + i) * Gm::NOISE_STRIDE;
            unsigned cv[4];
#pragma unroll
            for (int r = 0; r < 4; r++)
                cv[r] = kyber::compress_coeff<P::DU>(acc[r] + cbd_coeff<2>(e1, kyber::idx_l1(lane, r)));  // < 2q + 3: no reduction needed
            uint32_t *stage = reinterpret_cast<uint32_t *>(xch);
            stage_bits_l1<P::DU>(stage, cv, lane);
            if (MODE == REENCRYPT) differs |= ref.differs(stage, lane);
            else store_staged<P::DU>(dst, stage, lane, reject);
        }
        // v = InvNTT(<t-hat, r-hat>) + e2 + Decompress_q(m, 1)  (cpapke.go:167-173), dv bits
        {
            uint32_t *dst = reinterpret_cast<uint32_t *>(ctp + 32 * P::DU * K);
            RefWords<P::DV> ref;
            if (MODE == REENCRYPT) ref.load(dst, lane);
            int acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < K; j++) kyber::mulhat_acc_packed(acc, kyber::pack16(th[j][0], th[j][1]), kyber::pack16(th[j][2], th[j][3]), rop[j]);
            kyber::mulhat_finish(acc);
            kyber::invntt<kyber::NEG_R32>(acc, z, xch, lane);  // undoes mulhat_finish's -2^-32: plain coefficients in [0, q)
            const uint8_t *e2 = noise + 2 * K * Gm::NOISE_STRIDE;
            const uint8_t *mp = m + item * 32;
            unsigned cv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nidx = kyber::idx_l1(lane, r);
                const int mbit = (mp[nidx >> 3] >> (nidx & 7)) & 1;
                cv[r] = kyber::compress_coeff<P::DV>(acc[r] + cbd_coeff<2>(e2, nidx) + (-mbit & ((Q + 1) / 2)));  // < 3q
            }
            uint32_t *stage = reinterpret_cast<uint32_t *>(xch);
            stage_bits_l1<P::DV>(stage, cv, lane);
            if (MODE == REENCRYPT) differs |= ref.differs(stage, lane);
            else store_staged<P::DV>(dst, stage, lane, reject);
        }
        if (MODE == ENCAPS) {
            if (lane == 0) status[item] = reject ? 1 : 0;
            if (reject && lane < 8) reinterpret_cast<uint32_t *>(ss + item * 32)[lane] = 0;
        } else if (MODE == REENCRYPT) {
            // subtle.ConstantTimeCopy(ConstantTimeCompare(ct, ct'), ss2, K')  -- a lane-wise select here
            const bool mismatch = __any(differs);
            const bool dead = status[item] != 0;
            if (lane < 8) {
                const uint32_t kb = reinterpret_cast<const uint32_t *>(kbar_ws + item * 32)[lane];
                const uint32_t rj = reinterpret_cast<const uint32_t *>(ssrej_ws + item * 32)[lane];
                reinterpret_cast<uint32_t *>(ss + item * 32)[lane] = dead ? 0u : (mismatch ? rj : kb);
            }
        }
        if constexpr (CLK) clk[3]++;
    }
    lap(2);
  }
    if constexpr (CLK) {
        lap(0);  // (the last, empty-handed ticket)
        if (lane == 0) {
            uint64_t *prof = reinterpret_cast<uint64_t *>(const_cast<int16_t *>(key_rows)) + (size_t)blockIdx.x * 4;
#pragma unroll
            for (int q = 0; q < 4; q++) prof[q] = clk[q];
        }
    }
}

// ---- key tables: A^T of every table entry, once ---------------------------------------------------
// Single-wave workgroups, G table entries each: lane = (entry, i, j) runs the entry's SHAKE128 stream exactly as phase A
// of the encrypt kernel does, but the rows go to the table's cache (entry e: rows e K^2 .. e K^2 + K^2 - 1 of 256 int16)
// instead of a per-workgroup scratch.  The cache is allocated for a whole number of groups (the lanes of a partial last
// group repeat the last entry into the padding).  rho of entry e is at keys + e * stride + rho_off.
template <int K>
__global__ void __launch_bounds__(64, CIRCL_KEM_WAVES_PER_EU) mlkem_expand_keys_kernel(const uint8_t *__restrict__ keys, size_t stride, size_t rho_off,
                                                                                      int16_t *__restrict__ key_rows, size_t nkeys) {
    using Gm = Geom<K>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const size_t e0 = (size_t)blockIdx.x * Gm::G;
    sample_matrix_scratch<K, true>(smem, key_rows + e0 * (size_t)(K * K * 256), keys + rho_off, stride, e0, nkeys, threadIdx.x);
}

// ---- small batches: hashing and matrix expansion side by side ----------------------------------------------------
// Below ~2^14 items an encapsulation is a latency chain, not a throughput problem: H(ek) || G is ten dependent permutations
// on one lane (~100 us for a lone wavefront at ~5.4 cycles per instruction), and only then do the three matrix blocks, the
// PRF and the ring phase of the encrypt kernel start (another ~60 us), on a chip that is mostly idle.  The matrix does not
// depend on the hashes, so for small batches ONE launch runs both in different workgroups -- [0, nb_hash): H(ek), G(m || H(ek))
// of 64 items each, exactly mlkem_hash_kernel; the rest: A^T of G items each into the key-table cache, exactly
// mlkem_expand_keys_kernel -- and the key-table form of the encrypt kernel (PRF + ring phase, rows from the cache) follows:
// max(hash, expansion) + PRF + ring instead of their sum.  Results are bit-identical (same device functions).
// H(ek), G(m || H(ek)) of items item0 and item0 + 1 by ONE wavefront (keccak_f1600_coop2: 25 lanes per state).  Same values as
// mlkem_hash_kernel: SHA3-256 over the EK / 8 words of ek (rate 17 words, suffix 0x06), SHA3-512 over m || h (one block of
// rate 9 words); K' -> ss, r -> r_ws.
template <int K>
__device__ __forceinline__ void mlkem_hash_coop2(const uint8_t *__restrict__ ek, const uint8_t *__restrict__ m, uint8_t *__restrict__ ss,
                                                 uint8_t *__restrict__ r_ws, size_t item0, size_t n, uint64_t *ws, int lane) {
    using Gm = Geom<K>;
    constexpr int NW = Gm::EK / 8, FULL = NW / 17, REM = NW % 17;
    const CoopLane c = coop_lane(ws, lane);
    uint64_t *hx = ws + 100;  // 2 x 4 words: h on its way from lanes 0..3 to lanes 4..7
    const int half = lane >> 5, j = lane & 31;
    size_t item = item0 + half;
    const bool live = item < n;
    if (!live) item = n - 1;  // both halves run in lock step; the duplicate is not stored
    const uint64_t *ekw = reinterpret_cast<const uint64_t *>(ek + item * Gm::EK);
    uint32_t vlo = 0, vhi = 0;
    uint64_t next = j < 17 ? ekw[j] : 0;  // the block after the current one is requested before the permutation
#pragma unroll 1
    for (int b = 0; b < FULL; b++) {
        vlo ^= (uint32_t)next;
        vhi ^= (uint32_t)(next >> 32);
        next = (j < 17 && 17 * (b + 1) + j < NW) ? ekw[17 * (b + 1) + j] : 0;
        keccak_f1600_coop2(vlo, vhi, c);
    }
    vlo ^= (uint32_t)next;  // the REM words of the last block (zero beyond them)
    vhi ^= (uint32_t)(next >> 32);
    if (j == REM) vlo ^= kDsSha3;
    if (j == 16) vhi ^= 0x80000000u;
    keccak_f1600_coop2(vlo, vhi, c);
    // G: words 0..3 = m, 4..7 = h, word 8 = suffix and final bit
    __syncthreads();
    if (j < 4) hx[4 * half + j] = ((uint64_t)vhi << 32) | vlo;
    __syncthreads();
    uint64_t g = 0;
    if (j < 4) g = reinterpret_cast<const uint64_t *>(m + item * 32)[j];
    else if (j < 8) g = hx[4 * half + j - 4];
    else if (j == 8) g = 0x8000000000000000ull | kDsSha3;
    vlo = (uint32_t)g;
    vhi = (uint32_t)(g >> 32);
    keccak_f1600_coop2(vlo, vhi, c);
    if (live && j < 8) {
        uint64_t *dst = reinterpret_cast<uint64_t *>((j < 4 ? ss : r_ws) + item * 32);
        dst[j & 3] = ((uint64_t)vhi << 32) | vlo;
    }
}

template <int K>
__global__ void __launch_bounds__(64, CIRCL_KEM_WAVES_PER_EU) mlkem_small_pre_kernel(const uint8_t *__restrict__ ek, const uint8_t *__restrict__ m,
                                                                                    uint8_t *__restrict__ ss, uint8_t *__restrict__ r_ws,
                                                                                    int16_t *__restrict__ key_rows, size_t n, unsigned nb_hash, int coop) {
    using Gm = Geom<K>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if (blockIdx.x >= nb_hash) {
        const size_t e0 = (size_t)(blockIdx.x - nb_hash) * Gm::G;
        sample_matrix_scratch<K, true>(smem, key_rows + e0 * (size_t)(K * K * 256), ek + 384 * K, (size_t)Gm::EK, e0, n, threadIdx.x);
        return;
    }
    // The hashing wavefronts are the critical path (ten dependent permutations against three or four): they are dispatched
    // first and issue ahead of the expansion wavefronts that share their SIMD (measured at 2^14 items without the priority: the
    // launch took as long as hashing and expansion one after the other).
    __builtin_amdgcn_s_setprio(3);
    if (coop == 1) {  // very small batches: two items per wavefront, ~2.5 x shorter chain (keccak_f1600_coop2)
        mlkem_hash_coop2<K>(ek, m, ss, r_ws, 2 * (size_t)blockIdx.x, n, reinterpret_cast<uint64_t *>(smem), threadIdx.x);
        return;
    }
    if (coop == 2) {  // the rest of the small-batch range: an item per lane pair, 2/3 of the chain (keccak_f1600_split)
        size_t item = (size_t)blockIdx.x * 32 + (threadIdx.x >> 1);
        const bool live = item < n;
        if (!live) item = n - 1;  // keep the wave converged; the duplicate result is not stored
        mlkem_hash_split<K>(ek, m, ss, r_ws, item, live, threadIdx.x & 1);
        return;
    }
    size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = idx < n;
    if (!live) idx = n - 1;  // keep the wave converged; the duplicate result is not stored
    KeccakState h, g;
    sha3_256_words<Gm::EK / 8>(h, reinterpret_cast<const uint64_t *>(ek + idx * Gm::EK));
    sha3_512_m_h(g, reinterpret_cast<const uint64_t *>(m + idx * 32), h);
    if (live) {
        store_words<0, 4>(reinterpret_cast<uint64_t *>(ss + idx * 32), g);
        store_words<4, 4>(reinterpret_cast<uint64_t *>(r_ws + idx * 32), g);
    }
}

// Small batches under ONE public key (the reference's BenchmarkEncapsulate shape): workgroup 0 expands the key's A^T into
// entry 0 of the row cache; every other workgroup computes H(ek) itself on the cooperative permutation (the same nine
// permutations in every workgroup: no dependence between workgroups, and the chip is idle anyway) and then G for 32 items on
// lane pairs.  The key-table encrypt kernel follows (stride 0: every item takes entry 0) -- instead of H(ek), then G, then a
// kernel whose every workgroup samples A^T first: three dependent stages became max(expansion, H + G) and the ring phase.
template <int K>
__global__ void __launch_bounds__(64, CIRCL_KEM_WAVES_PER_EU) mlkem_small_shared_pre_kernel(const uint8_t *__restrict__ ek, const uint8_t *__restrict__ m,
                                                                                           uint8_t *__restrict__ ss, uint8_t *__restrict__ r_ws,
                                                                                           int16_t *__restrict__ key_rows, size_t n) {
    using Gm = Geom<K>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x;
    if (blockIdx.x == 0) {
        sample_matrix_scratch<K, true, 1>(smem, key_rows, ek + 384 * K, 0, 0, 1, lane);
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    uint64_t *ws = reinterpret_cast<uint64_t *>(smem);
    uint32_t *hx = reinterpret_cast<uint32_t *>(ws + 100);  // 8 dwords: H(ek) on its way from lanes 0..3 to every lane pair
    const CoopLane c = coop_lane(ws, lane);
    const uint64_t *ekw = reinterpret_cast<const uint64_t *>(ek);
    uint32_t vlo, vhi;
    coop_sponge17(vlo, vhi, [&](int k) { return ekw[k]; }, Gm::EK / 8, kDsSha3, c, lane & 31);
    __syncthreads();
    if (lane < 4) { hx[2 * lane] = vlo; hx[2 * lane + 1] = vhi; }
    __syncthreads();
    const int parity = lane & 1;
    size_t item = (size_t)(blockIdx.x - 1) * 32 + (lane >> 1);
    const bool live = item < n;
    if (!live) item = n - 1;
    const uint32_t h[4] = {hx[parity], hx[2 + parity], hx[4 + parity], hx[6 + parity]};
    mlkem_g_split(h, m, ss, r_ws, item, live, parity);
}

// ---- decapsulation ---------------------------------------------------------------------------

// K-PKE.Decrypt (cpapke.go:113-130) of one item by one wavefront: m' -> 32 bytes at `mprime`.
template <int K, bool NW = false>
__device__ __forceinline__ void mlkem_decrypt_item(const uint8_t *__restrict__ dkp, const uint8_t *__restrict__ ctp, uint8_t *mprime,
                                                   uint32_t *xch, int lane) {
    using P = Params<K>;
    const kyber::LaneZetas z = kyber::load_lane_zetas(lane);
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < K; j++) {
        int sh[4], u[4];
        unpack12_l4(sh, dkp + 384 * j, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            sh[r] = kyber::csubq(sh[r]);  // PrivateKey.Unpack normalises (cpapke.go:33-36)
            u[r] = kyber::decompress_coeff<P::DU>(get_bits<P::DU>(ctp + 32 * P::DU * j, kyber::idx_l1(lane, r)));
        }
        kyber::ntt<NW>(u, z, xch, lane);  // < 8q, left lazy
        kyber::mulhat_acc_packed(acc, kyber::pack16(sh[0], sh[1]), kyber::pack16(sh[2], sh[3]), kyber::hat_prepare(u, z.f6, z.f6n));
    }
    kyber::mulhat_finish(acc);
    kyber::invntt<kyber::NEG_R32, NW>(acc, z, xch, lane);  // <s-hat, u-hat> as plain coefficients in [0, q)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int nidx = kyber::idx_l1(lane, r);
        const int v = kyber::decompress_coeff<P::DV>(get_bits<P::DV>(ctp + 32 * P::DU * K, nidx));
        const int d = v - acc[r];  // in (-q, q)
        const unsigned bit = kyber::msg_bit(d + ((d >> 31) & Q));
        const unsigned long long mask = __ballot(bit != 0);  // bits of coefficients 64r .. 64r+63
        if (lane == 0) reinterpret_cast<unsigned long long *>(mprime)[r] = mask;
    }
}
// one item per single-wave workgroup: m' -> workspace
template <int K>
__global__ void __launch_bounds__(64) mlkem_decrypt_kernel(const uint8_t *__restrict__ dk, size_t dk_stride, const uint8_t *__restrict__ ct,
                                                          uint8_t *__restrict__ mprime_ws, size_t n, const KeyIdx key_idx) {
    using Gm = Geom<K>;
    __shared__ __attribute__((aligned(16))) uint32_t xch[256];
    const size_t item = blockIdx.x;
    // dk_stride = 0: one private key for the whole batch; key_idx: item t uses entry key_idx[t] of a key table
    const uint8_t *dkp = dk + (key_idx ? (size_t)key_idx[item] : item) * dk_stride;
    mlkem_decrypt_item<K>(dkp, ct + item * Gm::CT, mprime_ws + item * 32, xch, threadIdx.x);
}

// lane = item: the three sponges of decapsulation.
//   H(ek) over the ek embedded in dk, compared with the stored hash -> status 2 (kyber.go:219-228)
//   (K', r') = G(m' || hpk)   with hpk = the STORED hash (kyber.go:158-162 uses sk.hpk)
//   ss_rej   = J(z || ct) = SHAKE256(z || ct)[:32]  (kyber.go:171-174)
template <int K>
// Shared-key batches (dk_stride = 0) bring the verdict of the key's hash check in *key_status (mlkem_dk_check_kernel).
__global__ void __launch_bounds__(256) mlkem_decaps_hash_kernel(const uint8_t *__restrict__ dk, size_t dk_stride, const uint8_t *__restrict__ ct,
                                                                const uint8_t *__restrict__ mprime_ws, uint8_t *__restrict__ kbar_ws,
                                                                uint8_t *__restrict__ r_ws, uint8_t *__restrict__ ssrej_ws,
                                                                uint8_t *__restrict__ status, size_t n, const uint8_t *__restrict__ key_status,
                                                                const KeyIdx key_idx) {
    using Gm = Geom<K>;
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < n;
    if (!live) idx = n - 1;
    const size_t kq = key_idx ? (size_t)key_idx[idx] : idx;  // key-table batches (key_status then holds one verdict per table entry)
    const uint8_t *dkp = dk + kq * dk_stride;
    const uint64_t *stored = reinterpret_cast<const uint64_t *>(dkp + 768 * K + 32);
    KeccakState h, g;
    bool ok = true;
    if (key_status) {  // kernel-uniform
        ok = key_status[key_idx ? kq : 0] == 0;
    } else {
        sha3_256_words<Gm::EK / 8>(h, reinterpret_cast<const uint64_t *>(dkp + 384 * K));
#pragma unroll
        for (int i = 0; i < 4; i++) ok &= (((uint64_t)h.hi[i] << 32) | h.lo[i]) == stored[i];
    }
    // G(m' || stored hpk)
    keccak_zero(g);
    xor_words<0, 4>(g, reinterpret_cast<const uint64_t *>(mprime_ws + idx * 32));
    xor_words<4, 4>(g, stored);
    g.lo[8] = kDsSha3;
    g.hi[8] = 0x80000000u;
    keccak_f1600(g);
    if (live) {
        store_words<0, 4>(reinterpret_cast<uint64_t *>(kbar_ws + idx * 32), g);
        store_words<4, 4>(reinterpret_cast<uint64_t *>(r_ws + idx * 32), g);
        status[idx] = ok ? 0 : 2;
    }
    // J(z || ct): 4 + CT/8 words through SHAKE256 (rate 17 words)
    constexpr int CTW = Gm::CT / 8, TOTAL = 4 + CTW, FULL = TOTAL / 17, REM = TOTAL % 17;
    const uint64_t *zw = reinterpret_cast<const uint64_t *>(dkp + 768 * K + 64);
    const uint64_t *cw = reinterpret_cast<const uint64_t *>(ct + idx * Gm::CT);
    keccak_zero(h);
    xor_words<0, 4>(h, zw);
    xor_words<4, 13>(h, cw);
    keccak_f1600(h);
#pragma unroll 1
    for (int b = 1; b < FULL; b++) {
        xor_words<0, 17>(h, cw + 17 * b - 4);
        keccak_f1600(h);
    }
    xor_words<0, REM>(h, cw + 17 * FULL - 4);
    h.lo[REM] ^= kDsShake;
    h.hi[16] ^= 0x80000000u;
    keccak_f1600(h);
    if (live) store_words<0, 4>(reinterpret_cast<uint64_t *>(ssrej_ws + idx * 32), h);
}

// ---- small batches: decapsulation ------------------------------------------------------------------------------------
// The same reasoning as mlkem_small_pre_kernel, for decapsulation: per item there are three independent chains -- K-PKE.Decrypt
// followed by G(m' || hpk) (G takes the STORED hash, kyber.go:158-162, so it does not wait for the key's hash check), the check
// H(ek) == hpk itself (9 permutations), J(z || ct) (9 permutations) -- plus, for distinct keys, A^T for the re-encryption; the
// big-batch route runs decrypt, then all three sponges on one lane (19 dependent permutations, ~185 us), then the re-encryption.
// For small batches ONE launch runs them in different workgroups (the long chains first: J and H at raised priority, then the
// expansions; the n short Decrypt + G workgroups fill in around them -- with the expansions dispatched last the launch took 20 %
// longer at 2^13 items, and a one-key batch waited for its single expansion workgroup),
// two sponges per wavefront on the cooperative permutation while the chip has SIMDs to spare (`coop`), and the key-table form
// of the re-encryption follows.  dk_stride = 0: one private key for the batch (its hash check: ONE workgroup, verdict to
// *key_status; the per-item status bytes are filled by mlkem_fill_status_kernel).
template <int K>
__global__ void __launch_bounds__(64, CIRCL_KEM_WAVES_PER_EU) mlkem_small_decaps_pre_kernel(
    const uint8_t *__restrict__ dk, size_t dk_stride, const uint8_t *__restrict__ ct, uint8_t *__restrict__ mprime_ws, uint8_t *__restrict__ kbar_ws,
    uint8_t *__restrict__ r_ws, uint8_t *__restrict__ ssrej_ws, uint8_t *__restrict__ status, uint8_t *__restrict__ key_status,
    int16_t *__restrict__ key_rows, size_t n, unsigned nb_j, unsigned nb_h, int coop) {
    using Gm = Geom<K>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    unsigned b = blockIdx.x;
    constexpr int CTW = Gm::CT / 8;
    if (b < nb_j + nb_h) {
        __builtin_amdgcn_s_setprio(3);
        const bool is_j = b < nb_j;
        if (!is_j) b -= nb_j;
        const size_t per = coop == 1 ? 2 : coop == 2 ? 32 : 64;  // 1: two sponges per wavefront; 2: a sponge per lane pair; 0: per lane
        const size_t nitems = (!is_j && dk_stride == 0) ? 1 : n;  // one key: one hash check
        size_t idx = (size_t)b * per + (coop == 1 ? (size_t)half : coop == 2 ? (size_t)(lane >> 1) : (size_t)lane);
        const int parity = lane & 1;
        const bool live = idx < nitems;
        if (!live) idx = nitems - 1;
        const uint8_t *dkp = dk + idx * dk_stride;
        const uint64_t *stored = reinterpret_cast<const uint64_t *>(dkp + 768 * K + 32);
        if (is_j) {  // ss_rej = J(z || ct) = SHAKE256(z || ct)[:32] (kyber.go:171-174)
            const uint64_t *zw = reinterpret_cast<const uint64_t *>(dkp + 768 * K + 64);
            const uint64_t *cw = reinterpret_cast<const uint64_t *>(ct + idx * Gm::CT);
            if (coop == 1) {
                const CoopLane c = coop_lane(reinterpret_cast<uint64_t *>(smem), lane);
                uint32_t vlo, vhi;
                coop_sponge17(vlo, vhi, [&](int k) { return k < 4 ? zw[k] : cw[k - 4]; }, 4 + CTW, kDsShake, c, j);
                if (live && j < 4) reinterpret_cast<uint64_t *>(ssrej_ws + idx * 32)[j] = ((uint64_t)vhi << 32) | vlo;
            } else if (coop == 2) {
                const uint32_t *z32 = reinterpret_cast<const uint32_t *>(zw) + parity, *c32 = reinterpret_cast<const uint32_t *>(cw) + parity;
                SplitState h;
                split_sponge17_src<4 + CTW>(h, [&](int k) { return k < 4 ? z32[2 * k] : c32[2 * (k - 4)]; }, kDsShake, parity != 0);
                if (live) {
                    uint32_t *dst = reinterpret_cast<uint32_t *>(ssrej_ws + idx * 32) + parity;
#pragma unroll
                    for (int i = 0; i < 4; i++) dst[2 * i] = h.w[i];
                }
            } else {
                constexpr int TOTAL = 4 + CTW, FULL = TOTAL / 17, REM = TOTAL % 17;
                KeccakState h;
                keccak_zero(h);
                xor_words<0, 4>(h, zw);
                xor_words<4, 13>(h, cw);
                keccak_f1600(h);
#pragma unroll 1
                for (int bb = 1; bb < FULL; bb++) {
                    xor_words<0, 17>(h, cw + 17 * bb - 4);
                    keccak_f1600(h);
                }
                xor_words<0, REM>(h, cw + 17 * FULL - 4);
                h.lo[REM] ^= kDsShake;
                h.hi[16] ^= 0x80000000u;
                keccak_f1600(h);
                if (live) store_words<0, 4>(reinterpret_cast<uint64_t *>(ssrej_ws + idx * 32), h);
            }
        } else {  // H(ek) over the ek embedded in dk against the stored hash -> status 2 (kyber.go:219-228)
            const uint64_t *ekw = reinterpret_cast<const uint64_t *>(dkp + 384 * K);
            bool ok = true;
            if (coop == 1) {
                const CoopLane c = coop_lane(reinterpret_cast<uint64_t *>(smem), lane);
                uint32_t vlo, vhi;
                coop_sponge17(vlo, vhi, [&](int k) { return ekw[k]; }, Gm::EK / 8, kDsSha3, c, j);
                const bool mine = j >= 4 || ((((uint64_t)vhi << 32) | vlo) == stored[j & 3]);
                // all four words of the half must match: lanes 0..3 of half 0 are bits 0..3 of the ballot, of half 1 bits 32..35
                const unsigned long long bad = __ballot(!mine);
                ok = ((bad >> (32 * half)) & 0xfull) == 0;
                if (live && j == 0) (dk_stride == 0 ? key_status : status)[dk_stride == 0 ? 0 : idx] = ok ? 0 : 2;
            } else if (coop == 2) {
                SplitState h;
                split_sponge17<Gm::EK / 8>(h, reinterpret_cast<const uint32_t *>(ekw) + parity, kDsSha3, parity != 0);
                const uint32_t *st32 = reinterpret_cast<const uint32_t *>(stored) + parity;
                uint32_t diff = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) diff |= h.w[i] ^ st32[2 * i];
                diff |= split_partner(diff);  // both halves of every word must match
                if (live && parity == 0) (dk_stride == 0 ? key_status : status)[dk_stride == 0 ? 0 : idx] = diff == 0 ? 0 : 2;
            } else {
                KeccakState h;
                sha3_256_words<Gm::EK / 8>(h, ekw);
#pragma unroll
                for (int i = 0; i < 4; i++) ok &= (((uint64_t)h.hi[i] << 32) | h.lo[i]) == stored[i];
                if (live) (dk_stride == 0 ? key_status : status)[dk_stride == 0 ? 0 : idx] = ok ? 0 : 2;
            }
        }
        return;
    }
    b -= nb_j + nb_h;
    // the long chains first (hashes above, then the expansions: ~20 k instructions a wavefront), the short Decrypt + G workgroups
    // (~1.5 k) fill the slots around them and make up the tail
    const unsigned nb_expand = gridDim.x - nb_j - nb_h - (unsigned)n;
    if (b >= nb_expand) {  // K-PKE.Decrypt, then (K', r') = G(m' || hpk) by the same wavefront (one state on the cooperative permutation)
        const size_t item = b - nb_expand;
        const uint8_t *dkp = dk + item * dk_stride;
        uint32_t *xch = reinterpret_cast<uint32_t *>(smem);
        mlkem_decrypt_item<K>(dkp, ct + item * Gm::CT, mprime_ws + item * 32, xch, lane);
        __threadfence_block();
        __syncthreads();  // m' is visible to the lanes that absorb it; the exchange buffer is free
        const uint64_t *stored = reinterpret_cast<const uint64_t *>(dkp + 768 * K + 32);
        const CoopLane c = coop_lane(reinterpret_cast<uint64_t *>(smem), lane);
        uint64_t g = 0;
        if (j < 4) g = reinterpret_cast<const volatile uint64_t *>(mprime_ws + item * 32)[j];
        else if (j < 8) g = stored[j - 4];
        else if (j == 8) g = 0x8000000000000000ull | kDsSha3;
        uint32_t vlo = (uint32_t)g, vhi = (uint32_t)(g >> 32);
        keccak_f1600_coop2(vlo, vhi, c);  // (both halves carry the same state; half 0 stores)
        if (half == 0 && j < 8) reinterpret_cast<uint64_t *>((j < 4 ? kbar_ws : r_ws) + item * 32)[j & 3] = ((uint64_t)vhi << 32) | vlo;
        return;
    }
    const size_t e0 = (size_t)b * Gm::G;  // A^T of G items each into the cache (one key: one such workgroup)
    sample_matrix_scratch<K, true>(smem, key_rows + e0 * (size_t)(K * K * 256), dk + 768 * K, dk_stride, e0, n, lane);
}
// ---- resident keys, small batches: the whole decapsulation of an item as ONE workgroup of two wavefronts -------------------------
// With the key parsed beforehand (a key table: A^T rows, the verdict of the stored-hash check) a decapsulation is two independent
// chains that meet at the final select (kyber.go:144-184):
//   wave 0   m' = K-PKE.Decrypt(dk, ct) -> (K', r') = G(m' || h) [one cooperative permutation] -> the 2K+1 PRF streams [a stream per
//            lane pair, keccak_f1600_split: one permutation] -> ct' = K-PKE.Encrypt(ek, m', r') compared with ct on the fly
//   wave 1   J(z || ct): 9 sequential permutations (ML-KEM-768) on the cooperative form
// and ONE workgroup barrier later wave 0 stores (ct == ct') ? K' : J.  No flags between workgroups, no second and third launch:
// the small-batch route before it ran J / Decrypt+G / re-encryption as three dependent launches (profiles/r04_table_latency.txt).
// Every LDS buffer belongs to one wavefront and every wave-level ordering point inside is the no-wait form (kyber::wave_sync<true>),
// so the two wavefronts never meet at a barrier except the one at the end.  Grid = n workgroups.
// RESIDENT = false: the same for keys that are NOT parsed beforehand (circl_hip_mlkem_decaps on a small batch): item t has its own
// private key at dk + t dk_stride, and two more wavefronts do per item what a key table did once -- wave 2 the stored-hash check
// H(ek) == h (9 permutations, kyber.go:219-228), wave 3 the matrix A^T into LDS (cpapke.go:19-25) -- with one more barrier in front
// of the re-encryption.  Four wavefronts, 256 threads.
template <int K, bool RESIDENT = true>
__global__ void __launch_bounds__(RESIDENT ? 128 : 256) mlkem_decaps_chain_kernel(const uint8_t *__restrict__ dk, size_t dk_stride,
                                                                 const KeyIdx key_idx,
                                                                 const int16_t *__restrict__ key_rows, const uint8_t *__restrict__ key_status,
                                                                 const uint8_t *__restrict__ ct, uint8_t *__restrict__ ss, uint8_t *__restrict__ status,
                                                                 size_t n, const TailFlag tail = TailFlag{nullptr, nullptr, 0}) {
    using Gm = Geom<K>;
    using P = Params<K>;
    __shared__ __attribute__((aligned(16))) uint64_t coopw[RESIDENT ? 2 : 3][100];
    __shared__ __attribute__((aligned(16))) uint8_t lds_a[RESIDENT ? 16 : Gm::PAIRS * Gm::A_STRIDE];
    __shared__ uint32_t verdict_lds;
    __shared__ __attribute__((aligned(16))) uint32_t xch[256];
    __shared__ __attribute__((aligned(16))) uint8_t noise[Gm::NOISE * Gm::NOISE_STRIDE];
    __shared__ __attribute__((aligned(16))) uint64_t mprime[4];
    __shared__ __attribute__((aligned(16))) uint64_t kr[8];      // K' (words 0..3), r' (4..7)
    __shared__ __attribute__((aligned(16))) uint64_t ssrej[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, j = lane & 31;
    const size_t item = blockIdx.x;
    if (item >= n) return;  // (block-uniform)
    const size_t kq = RESIDENT ? (key_idx ? (size_t)key_idx[item] : size_t(0)) : item;
    const uint8_t *dkp = dk + kq * dk_stride;
    const uint8_t *ctp = ct + item * Gm::CT;
    constexpr int CTW = Gm::CT / 8;
    auto handoff = [] {  // what the lanes wrote to LDS (by whatever instruction) is visible to the wavefront's later reads
        __builtin_amdgcn_s_waitcnt(0);
        wave_lds_order();
    };
    bool differs = false;
    if (!RESIDENT && wave == 2) {  // the key's hash check: H(ek) against the stored hash
        const uint64_t *ekw = reinterpret_cast<const uint64_t *>(dkp + 384 * K);
        const uint64_t *stored = reinterpret_cast<const uint64_t *>(dkp + 768 * K + 32);
        const CoopLane c = coop_lane(coopw[RESIDENT ? 0 : 2], lane);
        uint32_t vlo, vhi;
        coop_sponge17<true>(vlo, vhi, [&](int k) { return ekw[k]; }, Gm::EK / 8, kDsSha3, c, j);
        const bool mine = j >= 4 || ((((uint64_t)vhi << 32) | vlo) == stored[j & 3]);
        const unsigned long long bad = __ballot(!mine);
        if (lane == 0) verdict_lds = (bad & 0xfull) ? 2u : 0u;
    } else if (!RESIDENT && wave == 3) {  // A^T of the item's key, a stream per lane (sample_matrix for one item)
        const bool on = lane < Gm::PAIRS;
        const int pi = on ? lane / K : 0, pj = on ? lane % K : 0;
        KeccakState sa;
        keccak_zero(sa);
        xor_words<0, 4>(sa, reinterpret_cast<const uint64_t *>(dkp + 768 * K));  // rho behind t-hat in the embedded ek
        sa.lo[4] = (uint32_t)pi | ((uint32_t)pj << 8) | (kDsShake << 16);
        sa.hi[20] = 0x80000000u;
        int16_t *poly = reinterpret_cast<int16_t *>(lds_a + (on ? lane : 0) * Gm::A_STRIDE);
        int cnt = on ? 0 : 256;
#pragma unroll 1
        for (int blk = 0; blk < 3 || __any(cnt < 256); blk++) {
            keccak_f1600(sa);
            if (on) parse_shake128_block(sa, poly, cnt);
        }
    } else if (wave == 1) {
        const uint64_t *zw = reinterpret_cast<const uint64_t *>(dkp + 768 * K + 64);
        const uint64_t *cw = reinterpret_cast<const uint64_t *>(ctp);
        const CoopLane c = coop_lane(coopw[1], lane);
        uint32_t vlo, vhi;
        coop_sponge17<true>(vlo, vhi, [&](int k) { return k < 4 ? zw[k] : cw[k - 4]; }, 4 + CTW, kDsShake, c, j);
        if (half == 0 && j < 4) ssrej[j] = ((uint64_t)vhi << 32) | vlo;
    } else {
        mlkem_decrypt_item<K, true>(dkp, ctp, reinterpret_cast<uint8_t *>(mprime), xch, lane);
        handoff();
        {   // (K', r') = G(m' || h) with the STORED hash (kyber.go:158-162)
            const uint64_t *stored = reinterpret_cast<const uint64_t *>(dkp + 768 * K + 32);
            const CoopLane c = coop_lane(coopw[0], lane);
            uint64_t g = 0;
            if (j < 4) g = mprime[j];
            else if (j < 8) g = stored[j - 4];
            else if (j == 8) g = 0x8000000000000000ull | kDsSha3;
            uint32_t vlo = (uint32_t)g, vhi = (uint32_t)(g >> 32);
            keccak_f1600_coop2<true>(vlo, vhi, c);
            if (half == 0 && j < 8) kr[j] = ((uint64_t)vhi << 32) | vlo;
        }
        handoff();
        {   // PRF(r', nonce) for the 2K+1 streams, a stream per lane pair (cpapke.go:139-147, sample.go:17-95)
            const int sidx = lane >> 1, parity = lane & 1;
            const bool on = sidx < Gm::NOISE;
            const uint32_t *seed = reinterpret_cast<const uint32_t *>(kr + 4) + parity;
            SplitState s;
#pragma unroll
            for (int w = 0; w < 25; w++) s.w[w] = w < 4 ? seed[2 * w] : 0u;
            if (parity == 0) s.w[4] = (uint32_t)(on ? sidx : 0) | (kDsShake << 8);
            else s.w[16] = 0x80000000u;
            keccak_f1600_split(s, parity != 0);
            uint32_t *out = reinterpret_cast<uint32_t *>(noise + (on ? sidx : 0) * Gm::NOISE_STRIDE) + parity;
            const bool eta2 = P::ETA1 == 2 || sidx >= K;
            if (on) {
                detail::static_for<0, 16>([&](auto ic) {
                    constexpr int w = decltype(ic)::v;
                    out[2 * w] = eta2 ? kyber::cbd2_bias8_word(s.w[w]) : s.w[w];
                });
            }
            if constexpr (P::ETA1 == 3) {  // 192 bytes for the eta1 = 3 streams: word 16 of this block, then 7 more of the next
                if (on && sidx < K) out[32] = s.w[16];
                keccak_f1600_split(s, parity != 0);
                if (on && sidx < K) {
                    detail::static_for<0, 7>([&](auto ic) {
                        constexpr int w = decltype(ic)::v;
                        out[34 + 2 * w] = s.w[w];
                    });
                }
            }
        }
        handoff();
    }
    if constexpr (!RESIDENT) __syncthreads();  // A^T is in LDS (and, as it happens, J and the hash check are done)
    if (wave == 0) {
        // ct' = K-PKE.Encrypt(ek, m', r') against ct (the REENCRYPT / KM_KEYED ring phase of mlkem_encrypt_kernel, one item)
        const kyber::LaneZetas z = kyber::load_lane_zetas(lane);
        const uint8_t *ekp = dkp + 384 * K;
        const int16_t *krows = RESIDENT ? key_rows + kq * (size_t)(K * K * 256) : nullptr;
        int th[K][4];
#pragma unroll
        for (int jj = 0; jj < K; jj++) {
            unpack12_l4(th[jj], ekp + 384 * jj, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) th[jj][r] = kyber::csubq(th[jj][r]);
        }
        kyber::HatOperand rop[K];
#pragma unroll
        for (int jj = 0; jj < K; jj++) {
            int rh[4];
#pragma unroll
            for (int r = 0; r < 4; r++) rh[r] = cbd_coeff<P::ETA1>(noise + jj * Gm::NOISE_STRIDE, kyber::idx_l1(lane, r));
            kyber::ntt<true>(rh, z, xch, lane);
            rop[jj] = kyber::hat_prepare(rh, z.f6, z.f6n);
        }
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            RefWords<P::DU> ref;
            ref.load(reinterpret_cast<const uint32_t *>(ctp + 32 * P::DU * i), lane);
            int acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int jj = 0; jj < K; jj++) {
                uint32_t a01, a23;
                if constexpr (RESIDENT) AFromCache{krows}.load(a01, a23, i * K + jj, lane);
                else AFromLds{lds_a, Gm::A_STRIDE}.load(a01, a23, i * K + jj, lane);
                kyber::mulhat_acc_packed(acc, a01, a23, rop[jj]);
            }
            kyber::mulhat_finish(acc);
            kyber::invntt<kyber::NEG_R32, true>(acc, z, xch, lane);
            const uint8_t *e1 = noise + (K + i) * Gm::NOISE_STRIDE;
            unsigned cv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) cv[r] = kyber::compress_coeff<P::DU>(acc[r] + cbd_coeff<2>(e1, kyber::idx_l1(lane, r)));
            stage_bits_l1<P::DU, true>(xch, cv, lane);
            differs |= ref.differs(xch, lane);
        }
        {
            RefWords<P::DV> ref;
            ref.load(reinterpret_cast<const uint32_t *>(ctp + 32 * P::DU * K), lane);
            int acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int jj = 0; jj < K; jj++) kyber::mulhat_acc_packed(acc, kyber::pack16(th[jj][0], th[jj][1]), kyber::pack16(th[jj][2], th[jj][3]), rop[jj]);
            kyber::mulhat_finish(acc);
            kyber::invntt<kyber::NEG_R32, true>(acc, z, xch, lane);
            const uint8_t *e2 = noise + 2 * K * Gm::NOISE_STRIDE;
            const uint8_t *mp = reinterpret_cast<const uint8_t *>(mprime);
            unsigned cv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nidx = kyber::idx_l1(lane, r);
                const int mbit = (mp[nidx >> 3] >> (nidx & 7)) & 1;
                cv[r] = kyber::compress_coeff<P::DV>(acc[r] + cbd_coeff<2>(e2, nidx) + (-mbit & ((Q + 1) / 2)));
            }
            stage_bits_l1<P::DV, true>(xch, cv, lane);
            differs |= ref.differs(xch, lane);
        }
    }
    __syncthreads();  // the one meeting point of the two chains
    if (wave == 0) {
        // subtle.ConstantTimeCopy(ConstantTimeCompare(ct, ct'), ss2, K') (kyber.go:176-181); a key that failed its hash check: zeros, status 2
        const bool mismatch = __any(differs);
        const uint8_t verdict = RESIDENT ? key_status[kq] : (uint8_t)verdict_lds;
        if (lane < 8) {
            const uint32_t kb = reinterpret_cast<const uint32_t *>(kr)[lane], rj = reinterpret_cast<const uint32_t *>(ssrej)[lane];
            reinterpret_cast<uint32_t *>(ss + item * 32)[lane] = verdict ? 0u : (mismatch ? rj : kb);
        }
        if (lane == 0) status[item] = verdict;
    }
    if constexpr (RESIDENT) tail_signal(tail);  // (the coalescer's completion flag: every wavefront of a resident-key workgroup gets here)
}

// The encapsulation to a resident key, small batches, in ONE launch: a wavefront per item runs (K, r) = G(m || H(ek)) on the
// cooperative permutation, the 2K+1 PRF streams a stream per lane pair, and K-PKE.Encrypt with the table's A^T rows
// (kyber.go:103-137 EncapsulateTo on a parsed key) -- the route before it ran G for all items in one launch (a lane per item)
// and the PRF + ring phase in a second one.  key_h: H(ek) per table entry.  Grid = n single-wave workgroups.
// RESIDENT = false: keys that are not parsed beforehand (circl_hip_mlkem_encaps on a small batch: item t has its own key at ek + t
// ek_stride): wave 0 first hashes the key (H(ek), 9 permutations on the cooperative form), a second wavefront expands A^T into
// LDS meanwhile, and one barrier sits in front of the ring phase.  128 threads.
template <int K, bool RESIDENT = true>
__global__ void __launch_bounds__(RESIDENT ? 64 : 128) mlkem_encaps_chain_kernel(const uint8_t *__restrict__ ek, size_t ek_stride,
                                                                const KeyIdx key_idx,
                                                                const int16_t *__restrict__ key_rows, const uint8_t *__restrict__ key_h,
                                                                const uint8_t *__restrict__ m, uint8_t *__restrict__ ct, uint8_t *__restrict__ ss,
                                                                uint8_t *__restrict__ status, size_t n, const TailFlag tail = TailFlag{nullptr, nullptr, 0}) {
    using Gm = Geom<K>;
    using P = Params<K>;
    __shared__ __attribute__((aligned(16))) uint64_t coopw[100];
    __shared__ __attribute__((aligned(16))) uint32_t xch[256];
    __shared__ __attribute__((aligned(16))) uint8_t noise[Gm::NOISE * Gm::NOISE_STRIDE];
    __shared__ __attribute__((aligned(16))) uint64_t kr[8];      // K (words 0..3), r (4..7)
    __shared__ __attribute__((aligned(16))) uint64_t hk[4];      // H(ek) (RESIDENT = false)
    __shared__ __attribute__((aligned(16))) uint8_t lds_a[RESIDENT ? 16 : Gm::PAIRS * Gm::A_STRIDE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, j = lane & 31;
    const size_t item = blockIdx.x;
    if (item >= n) return;
    const size_t kq = RESIDENT ? (key_idx ? (size_t)key_idx[item] : size_t(0)) : item;
    const uint8_t *ekp = ek + kq * ek_stride;
    const uint8_t *mp = m + item * 32;
    auto handoff = [] {
        __builtin_amdgcn_s_waitcnt(0);
        wave_lds_order();
    };
    if (!RESIDENT && wave == 1) {  // A^T of the item's key, a stream per lane (sample_matrix for one item)
        const bool on = lane < Gm::PAIRS;
        const int pi = on ? lane / K : 0, pj = on ? lane % K : 0;
        KeccakState sa;
        keccak_zero(sa);
        xor_words<0, 4>(sa, reinterpret_cast<const uint64_t *>(ekp + 384 * K));
        sa.lo[4] = (uint32_t)pi | ((uint32_t)pj << 8) | (kDsShake << 16);
        sa.hi[20] = 0x80000000u;
        int16_t *poly = reinterpret_cast<int16_t *>(lds_a + (on ? lane : 0) * Gm::A_STRIDE);
        int cnt = on ? 0 : 256;
#pragma unroll 1
        for (int blk = 0; blk < 3 || __any(cnt < 256); blk++) {
            keccak_f1600(sa);
            if (on) parse_shake128_block(sa, poly, cnt);
        }
    }
    if (wave == 0) {
        if constexpr (!RESIDENT) {  // H(ek) (kyber.go:247-263 caches it in the parsed key)
            const uint64_t *ekw = reinterpret_cast<const uint64_t *>(ekp);
            const CoopLane c = coop_lane(coopw, lane);
            uint32_t vlo, vhi;
            coop_sponge17<true>(vlo, vhi, [&](int k) { return ekw[k]; }, Gm::EK / 8, kDsSha3, c, j);
            if (half == 0 && j < 4) hk[j] = ((uint64_t)vhi << 32) | vlo;
            handoff();
        }
        const CoopLane c = coop_lane(coopw, lane);
        uint64_t g = 0;
        if (j < 4) g = reinterpret_cast<const uint64_t *>(mp)[j];
        else if (j < 8) g = RESIDENT ? reinterpret_cast<const uint64_t *>(key_h + kq * 32)[j - 4] : hk[j - 4];
        else if (j == 8) g = 0x8000000000000000ull | kDsSha3;
        uint32_t vlo = (uint32_t)g, vhi = (uint32_t)(g >> 32);
        keccak_f1600_coop2<true>(vlo, vhi, c);
        if (half == 0 && j < 8) kr[j] = ((uint64_t)vhi << 32) | vlo;
        handoff();
        const int sidx = lane >> 1, parity = lane & 1;
        const bool on = sidx < Gm::NOISE;
        const uint32_t *seed = reinterpret_cast<const uint32_t *>(kr + 4) + parity;
        SplitState s;
#pragma unroll
        for (int w = 0; w < 25; w++) s.w[w] = w < 4 ? seed[2 * w] : 0u;
        if (parity == 0) s.w[4] = (uint32_t)(on ? sidx : 0) | (kDsShake << 8);
        else s.w[16] = 0x80000000u;
        keccak_f1600_split(s, parity != 0);
        uint32_t *out = reinterpret_cast<uint32_t *>(noise + (on ? sidx : 0) * Gm::NOISE_STRIDE) + parity;
        const bool eta2 = P::ETA1 == 2 || sidx >= K;
        if (on) {
            detail::static_for<0, 16>([&](auto ic) {
                constexpr int w = decltype(ic)::v;
                out[2 * w] = eta2 ? kyber::cbd2_bias8_word(s.w[w]) : s.w[w];
            });
        }
        if constexpr (P::ETA1 == 3) {
            if (on && sidx < K) out[32] = s.w[16];
            keccak_f1600_split(s, parity != 0);
            if (on && sidx < K) {
                detail::static_for<0, 7>([&](auto ic) {
                    constexpr int w = decltype(ic)::v;
                    out[34 + 2 * w] = s.w[w];
                });
            }
        }
        handoff();
    }
    if constexpr (!RESIDENT) {
        __syncthreads();  // A^T is in LDS
        if (wave != 0) return;
    }
    const kyber::LaneZetas z = kyber::load_lane_zetas(lane);
    const int16_t *krows = RESIDENT ? key_rows + kq * (size_t)(K * K * 256) : nullptr;
    int th[K][4];
    bool bad = false;
#pragma unroll
    for (int jj = 0; jj < K; jj++) {
        unpack12_l4(th[jj], ekp + 384 * jj, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) bad |= th[jj][r] >= Q;  // UnpackMLKEM's range check (cpapke.go:45-55): kem.ErrPubKey
    }
    const bool reject = __any(bad);
    kyber::HatOperand rop[K];
#pragma unroll
    for (int jj = 0; jj < K; jj++) {
        int rh[4];
#pragma unroll
        for (int r = 0; r < 4; r++) rh[r] = cbd_coeff<P::ETA1>(noise + jj * Gm::NOISE_STRIDE, kyber::idx_l1(lane, r));
        kyber::ntt<true>(rh, z, xch, lane);
        rop[jj] = kyber::hat_prepare(rh, z.f6, z.f6n);
    }
    uint8_t *ctp = ct + item * Gm::CT;
#pragma unroll 1
    for (int i = 0; i < K; i++) {
        int acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int jj = 0; jj < K; jj++) {
            uint32_t a01, a23;
            if constexpr (RESIDENT) AFromCache{krows}.load(a01, a23, i * K + jj, lane);
            else AFromLds{lds_a, Gm::A_STRIDE}.load(a01, a23, i * K + jj, lane);
            kyber::mulhat_acc_packed(acc, a01, a23, rop[jj]);
        }
        kyber::mulhat_finish(acc);
        kyber::invntt<kyber::NEG_R32, true>(acc, z, xch, lane);
        const uint8_t *e1 = noise + (K + i) * Gm::NOISE_STRIDE;
        unsigned cv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) cv[r] = kyber::compress_coeff<P::DU>(acc[r] + cbd_coeff<2>(e1, kyber::idx_l1(lane, r)));
        stage_bits_l1<P::DU, true>(xch, cv, lane);
        store_staged<P::DU>(reinterpret_cast<uint32_t *>(ctp + 32 * P::DU * i), xch, lane, reject);
    }
    {
        int acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int jj = 0; jj < K; jj++) kyber::mulhat_acc_packed(acc, kyber::pack16(th[jj][0], th[jj][1]), kyber::pack16(th[jj][2], th[jj][3]), rop[jj]);
        kyber::mulhat_finish(acc);
        kyber::invntt<kyber::NEG_R32, true>(acc, z, xch, lane);
        const uint8_t *e2 = noise + 2 * K * Gm::NOISE_STRIDE;
        unsigned cv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int nidx = kyber::idx_l1(lane, r);
            const int mbit = (mp[nidx >> 3] >> (nidx & 7)) & 1;
            cv[r] = kyber::compress_coeff<P::DV>(acc[r] + cbd_coeff<2>(e2, nidx) + (-mbit & ((Q + 1) / 2)));
        }
        stage_bits_l1<P::DV, true>(xch, cv, lane);
        store_staged<P::DV>(reinterpret_cast<uint32_t *>(ctp + 32 * P::DU * K), xch, lane, reject);
    }
    if (lane == 0) status[item] = reject ? 1 : 0;
    if (lane < 8) reinterpret_cast<uint32_t *>(ss + item * 32)[lane] = reject ? 0u : reinterpret_cast<const uint32_t *>(kr)[lane];
    if constexpr (RESIDENT) tail_signal(tail);
}

// one key for the batch: every item's status byte is the key's verdict
static __global__ void __launch_bounds__(256) mlkem_fill_status_kernel(uint8_t *__restrict__ status, const uint8_t *__restrict__ key_status, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) status[i] = *key_status;
}

// ---- key generation ---------------------------------------------------------------------------

// lane = item: (rho, sigma) = G(d || K) (cpapke.go:72-79 with the FIPS 203 domain byte,
// pke/kyber/kyber768/kyber.go:77-86) -> workspace (rho 32 B, sigma 32 B per item).
// R3 = round-3 Kyber (kem/kyber/kyber768/kyber.go:60-82): G(d) without the domain byte.
template <int K, bool R3 = false>
__global__ void __launch_bounds__(256) mlkem_keygen_seed_kernel(const uint8_t *__restrict__ seed64, uint8_t *__restrict__ rs_ws, size_t n) {
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    KeccakState g;
    keccak_zero(g);
    xor_words<0, 4>(g, reinterpret_cast<const uint64_t *>(seed64 + idx * 64));
    g.lo[4] = R3 ? kDsSha3 : ((uint32_t)K | (kDsSha3 << 8));
    g.hi[8] = 0x80000000u;
    keccak_f1600(g);
    store_words<0, 8>(reinterpret_cast<uint64_t *>(rs_ws + idx * 64), g);
}

// One wavefront per workgroup, G items: K-PKE.KeyGen (cpapke.go:66-110).
//   t-hat[i] = ToMont(sum_j A[i][j] s-hat[j]) + e-hat[i], normalised; ek = Pack(t-hat) || rho;
//   dk = Pack(s-hat) || ek || (H(ek), z filled in by mlkem_keygen_finish_kernel).
template <int K, bool SCRATCH = true>
__global__ void __launch_bounds__(64, SCRATCH ? CIRCL_KEM_WAVES_PER_EU : 1) mlkem_keygen_kernel(const uint8_t *__restrict__ rs_ws, uint8_t *__restrict__ ek,
                                                         uint8_t *__restrict__ dk, uint8_t *__restrict__ scratch, unsigned *__restrict__ work, size_t n) {
    using Gm = Geom<K>;
    using P = Params<K>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *lds_a = smem;
    uint8_t *lds_noise = SCRATCH ? smem : smem + Gm::LDS_A;
    uint8_t *xch = lds_noise + Gm::LDS_NOISE;
    int16_t *rows = reinterpret_cast<int16_t *>(scratch + (size_t)blockIdx.x * Gm::SCRATCH_BYTES);
    const int lane = threadIdx.x;
    const kyber::LaneZetas z = kyber::load_lane_zetas(lane);
    const size_t ngroups = (n + Gm::G - 1) / Gm::G;

#pragma unroll 1
  for (size_t grp = next_group(work, lane, true, ngroups); grp < ngroups; grp = next_group(work, lane, false, ngroups)) {
    const size_t item0 = grp * Gm::G;
    if constexpr (CIRCL_KEM_RING_PRIO != 0) __builtin_amdgcn_s_setprio(0);  // (as mlkem_encrypt_kernel: the ring phase below gets issue priority)
    if constexpr (SCRATCH && Gm::HALVES == 1) {
        __syncthreads();
        sample_matrix_and_prf<K, false, 2 * K, 2 * K>(lds_a, lds_noise, xch, rows, rs_ws, 64, rs_ws + 32, 64,
                                                      item0, n, lane);
        __threadfence_block();
        __syncthreads();
    } else {
        if constexpr (SCRATCH) {
            __syncthreads();
            sample_matrix_scratch<K, false>(lds_a, rows, rs_ws, 64, item0, n, lane);
            __threadfence_block();
        } else {
            sample_matrix<K, false>(lds_a, rs_ws, 64, item0, n, lane);
        }
        __syncthreads();
        if constexpr (Gm::HALVES == 1) {
            prf_streams<K, 2 * K, 2 * K>(lds_noise, rs_ws + 32, 64, item0, n, lane);
            __syncthreads();
        }
    }

    if constexpr (CIRCL_KEM_RING_PRIO != 0) __builtin_amdgcn_s_setprio(CIRCL_KEM_RING_PRIO);
#pragma unroll 1
    for (int g = 0; g < Gm::G; g++) {
        const size_t item = item0 + g;
        if (item >= n) break;
        if constexpr (Gm::HALVES > 1) {
            if (g % Gm::GH == 0) {
                __syncthreads();
                prf_streams<K, 2 * K, 2 * K, Gm::GH>(lds_noise, rs_ws + 32, 64, item0, n, lane, g);
                __syncthreads();
            }
        }
        const uint8_t *noise = lds_noise + (g % Gm::GH) * (2 * K) * Gm::NOISE_STRIDE;
        uint8_t *ekp = ek + item * Gm::EK, *dkp = dk + item * Gm::DK;
        int sh[K][4];
#pragma unroll
        for (int j = 0; j < K; j++) {
#pragma unroll
            for (int r = 0; r < 4; r++) sh[j][r] = cbd_coeff<P::ETA1>(noise + j * Gm::NOISE_STRIDE, kyber::idx_l1(lane, r));
            kyber::ntt(sh[j], z, xch, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) sh[j][r] = kyber::normalize(sh[j][r]);
            pack12_l4(dkp + 384 * j, sh[j], lane);
        }
        kyber::HatOperand sop[K];
#pragma unroll
        for (int j = 0; j < K; j++) sop[j] = kyber::hat_prepare(sh[j], z.f6, z.f6n);
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            int eh[4], acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; r++) eh[r] = cbd_coeff<P::ETA1>(noise + (K + i) * Gm::NOISE_STRIDE, kyber::idx_l1(lane, r));
            kyber::ntt(eh, z, xch, lane);
#pragma unroll
            for (int j = 0; j < K; j++) {
                uint32_t a01, a23;
                if constexpr (SCRATCH) AFromScratch{rows}.load(a01, a23, (g * K + i) * K + j, lane);
                else AFromLds{lds_a, Gm::A_STRIDE}.load(a01, a23, (g * K + i) * K + j, lane);
                kyber::mulhat_acc_packed(acc, a01, a23, sop[j]);
            }
            kyber::mulhat_finish(acc);
            int t[4];
#pragma unroll
            for (int r = 0; r < 4; r++)  // undo mulhat_finish's -2^-32 (the reference's ToMont, field.go:35-39, undoes its R^-1)
                t[r] = kyber::normalize((int)kyber::mulc((uint32_t)acc[r], kyber::mulc_const(kyber::NEG_R32)) + eh[r]);
            pack12_l4(ekp + 384 * i, t, lane);
            pack12_l4(dkp + 384 * K + 384 * i, t, lane);
        }
        if (lane < 8) {
            const uint32_t w = reinterpret_cast<const uint32_t *>(rs_ws + item * 64)[lane];
            reinterpret_cast<uint32_t *>(ekp + 384 * K)[lane] = w;
            reinterpret_cast<uint32_t *>(dkp + 768 * K)[lane] = w;
        }
    }
  }
}

// lane = item: dk tail = H(ek) || z  (kem/mlkem/mlkem768/kyber.go:69-75, :189-201)
template <int K>
__global__ void __launch_bounds__(256) mlkem_keygen_finish_kernel(const uint8_t *__restrict__ seed64, const uint8_t *__restrict__ ek,
                                                                  uint8_t *__restrict__ dk, size_t n) {
    using Gm = Geom<K>;
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < n;
    if (!live) idx = n - 1;
    KeccakState h;
    sha3_256_words<Gm::EK / 8>(h, reinterpret_cast<const uint64_t *>(ek + idx * Gm::EK));
    if (live) {
        uint64_t *tail = reinterpret_cast<uint64_t *>(dk + idx * Gm::DK + 768 * K + 32);
        store_words<0, 4>(tail, h);
        const uint64_t *zsrc = reinterpret_cast<const uint64_t *>(seed64 + idx * 64 + 32);
#pragma unroll
        for (int i = 0; i < 4; i++) tail[4 + i] = zsrc[i];
    }
}

// The same for small and medium batches, where the nine permutations of H(ek) on a lone lane are the longest stage of a key
// generation: form 1 = two keys per wavefront on the cooperative permutation, form 2 = a key per lane pair (32 per wavefront).
template <int K>
__global__ void __launch_bounds__(64) mlkem_keygen_finish_small_kernel(const uint8_t *__restrict__ seed64, const uint8_t *__restrict__ ek,
                                                                       uint8_t *__restrict__ dk, size_t n, int form) {
    using Gm = Geom<K>;
    __shared__ uint64_t ws[100];
    const int lane = threadIdx.x;
    if (form == 1) {
        const int half = lane >> 5, j = lane & 31;
        size_t idx = 2 * (size_t)blockIdx.x + half;
        const bool live = idx < n;
        if (!live) idx = n - 1;
        const CoopLane c = coop_lane(ws, lane);
        const uint64_t *ekw = reinterpret_cast<const uint64_t *>(ek + idx * Gm::EK);
        uint32_t vlo, vhi;
        coop_sponge17(vlo, vhi, [&](int k) { return ekw[k]; }, Gm::EK / 8, kDsSha3, c, j);
        uint64_t *tail = reinterpret_cast<uint64_t *>(dk + idx * Gm::DK + 768 * K + 32);
        if (live && j < 4) tail[j] = ((uint64_t)vhi << 32) | vlo;
        else if (live && j < 8) tail[j] = reinterpret_cast<const uint64_t *>(seed64 + idx * 64 + 32)[j - 4];
        return;
    }
    const int parity = lane & 1;
    size_t idx = (size_t)blockIdx.x * 32 + (lane >> 1);
    const bool live = idx < n;
    if (!live) idx = n - 1;
    SplitState h;
    split_sponge17<Gm::EK / 8>(h, reinterpret_cast<const uint32_t *>(ek + idx * Gm::EK) + parity, kDsSha3, parity != 0);
    if (live) {
        uint32_t *tail = reinterpret_cast<uint32_t *>(dk + idx * Gm::DK + 768 * K + 32) + parity;
        const uint32_t *zsrc = reinterpret_cast<const uint32_t *>(seed64 + idx * 64 + 32) + parity;
#pragma unroll
        for (int i = 0; i < 4; i++) { tail[2 * i] = h.w[i]; tail[8 + 2 * i] = zsrc[2 * i]; }
    }
}

// Key generation of a small batch in ONE launch (kem/mlkem/mlkem768/kyber.go:57-78 NewKeyFromSeed -> cpapke.go:66-110), two wavefronts
// per key: wave 0 runs (rho, sigma) = G(d || K) on the cooperative permutation, then -- while wave 1 expands A into LDS -- the 2K PRF
// streams a stream per lane pair and the transforms of s; one barrier later t-hat = A s-hat + e-hat, the 12-bit packing (ek is also
// kept in LDS: the sponge of H(ek) absorbs it from there), and H(ek) || z behind it.  The three-launch form before it ran G for all
// keys, then K-PKE.KeyGen, then H(ek) (profiles/r04_latency.txt).  Grid = n workgroups of 128 threads.
template <int K, bool R3 = false>
__global__ void __launch_bounds__(128) mlkem_keygen_chain_kernel(const uint8_t *__restrict__ seed64, uint8_t *__restrict__ ek, uint8_t *__restrict__ dk,
                                                                 size_t n) {
    using Gm = Geom<K>;
    using P = Params<K>;
    __shared__ __attribute__((aligned(16))) uint64_t coopw[100];
    __shared__ __attribute__((aligned(16))) uint32_t xch[256];
    __shared__ __attribute__((aligned(16))) uint8_t noise[2 * K * Gm::NOISE_STRIDE];
    __shared__ __attribute__((aligned(16))) uint64_t rs[8];  // rho (words 0..3), sigma (4..7)
    __shared__ __attribute__((aligned(16))) uint8_t lds_a[Gm::PAIRS * Gm::A_STRIDE];
    __shared__ __attribute__((aligned(16))) uint8_t ekl[Gm::EK];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, j = lane & 31;
    const size_t item = blockIdx.x;
    if (item >= n) return;  // (block-uniform)
    const uint8_t *sd = seed64 + item * 64;
    uint8_t *ekp = ek + item * Gm::EK, *dkp = dk + item * Gm::DK;
    auto handoff = [] {
        __builtin_amdgcn_s_waitcnt(0);
        wave_lds_order();
    };
    if (wave == 0) {  // (rho, sigma) = G(d || K) (cpapke.go:72-79 with the FIPS 203 domain byte; round 3: G(d))
        const CoopLane c = coop_lane(coopw, lane);
        uint64_t g = 0;
        if (j < 4) g = reinterpret_cast<const uint64_t *>(sd)[j];
        else if (j == 4) g = R3 ? (uint64_t)kDsSha3 : ((uint64_t)K | ((uint64_t)kDsSha3 << 8));
        else if (j == 8) g = 0x8000000000000000ull;
        uint32_t vlo = (uint32_t)g, vhi = (uint32_t)(g >> 32);
        keccak_f1600_coop2<true>(vlo, vhi, c);
        if (half == 0 && j < 8) rs[j] = ((uint64_t)vhi << 32) | vlo;
    }
    __syncthreads();  // rho and sigma are in LDS
    kyber::HatOperand sop[K];
    const kyber::LaneZetas z = kyber::load_lane_zetas(lane);
    if (wave == 1) {  // A (not transposed: stream (i, j) from x = j, y = i, mat.go:13-74), a stream per lane
        const bool on = lane < Gm::PAIRS;
        const int pi = on ? lane / K : 0, pj = on ? lane % K : 0;
        KeccakState sa;
        keccak_zero(sa);
        xor_words<0, 4>(sa, rs);
        sa.lo[4] = (uint32_t)pj | ((uint32_t)pi << 8) | (kDsShake << 16);
        sa.hi[20] = 0x80000000u;
        int16_t *poly = reinterpret_cast<int16_t *>(lds_a + (on ? lane : 0) * Gm::A_STRIDE);
        int cnt = on ? 0 : 256;
#pragma unroll 1
        for (int blk = 0; blk < 3 || __any(cnt < 256); blk++) {
            keccak_f1600(sa);
            if (on) parse_shake128_block(sa, poly, cnt);
        }
    } else {
        {   // PRF(sigma, nonce) for the 2K eta1 streams of s and e, a stream per lane pair
            const int sidx = lane >> 1, parity = lane & 1;
            const bool on = sidx < 2 * K;
            const uint32_t *seed = reinterpret_cast<const uint32_t *>(rs + 4) + parity;
            SplitState s;
#pragma unroll
            for (int w = 0; w < 25; w++) s.w[w] = w < 4 ? seed[2 * w] : 0u;
            if (parity == 0) s.w[4] = (uint32_t)(on ? sidx : 0) | (kDsShake << 8);
            else s.w[16] = 0x80000000u;
            keccak_f1600_split(s, parity != 0);
            uint32_t *out = reinterpret_cast<uint32_t *>(noise + (on ? sidx : 0) * Gm::NOISE_STRIDE) + parity;
            if (on) {
                detail::static_for<0, 16>([&](auto ic) {
                    constexpr int w = decltype(ic)::v;
                    out[2 * w] = P::ETA1 == 2 ? kyber::cbd2_bias8_word(s.w[w]) : s.w[w];
                });
            }
            if constexpr (P::ETA1 == 3) {
                if (on) out[32] = s.w[16];
                keccak_f1600_split(s, parity != 0);
                if (on) {
                    detail::static_for<0, 7>([&](auto ic) {
                        constexpr int w = decltype(ic)::v;
                        out[34 + 2 * w] = s.w[w];
                    });
                }
            }
        }
        handoff();
#pragma unroll
        for (int jj = 0; jj < K; jj++) {
            int sh[4];
#pragma unroll
            for (int r = 0; r < 4; r++) sh[r] = cbd_coeff<P::ETA1>(noise + jj * Gm::NOISE_STRIDE, kyber::idx_l1(lane, r));
            kyber::ntt<true>(sh, z, xch, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) sh[r] = kyber::normalize(sh[r]);
            pack12_l4(dkp + 384 * jj, sh, lane);
            sop[jj] = kyber::hat_prepare(sh, z.f6, z.f6n);
        }
    }
    __syncthreads();  // A is in LDS
    if (wave != 0) return;
#pragma unroll 1
    for (int i = 0; i < K; i++) {
        int eh[4], acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; r++) eh[r] = cbd_coeff<P::ETA1>(noise + (K + i) * Gm::NOISE_STRIDE, kyber::idx_l1(lane, r));
        kyber::ntt<true>(eh, z, xch, lane);
#pragma unroll
        for (int jj = 0; jj < K; jj++) {
            uint32_t a01, a23;
            AFromLds{lds_a, Gm::A_STRIDE}.load(a01, a23, i * K + jj, lane);
            kyber::mulhat_acc_packed(acc, a01, a23, sop[jj]);
        }
        kyber::mulhat_finish(acc);
        int t[4];
#pragma unroll
        for (int r = 0; r < 4; r++) t[r] = kyber::normalize((int)kyber::mulc((uint32_t)acc[r], kyber::mulc_const(kyber::NEG_R32)) + eh[r]);
        pack12_l4(ekl + 384 * i, t, lane);
    }
    if (lane < 8) reinterpret_cast<uint32_t *>(ekl + 384 * K)[lane] = reinterpret_cast<const uint32_t *>(rs)[lane];  // rho
    handoff();
    for (int d = lane; d < Gm::EK / 4; d += 64) {  // ek, and its copy inside dk
        const uint32_t w = reinterpret_cast<const uint32_t *>(ekl)[d];
        reinterpret_cast<uint32_t *>(ekp)[d] = w;
        reinterpret_cast<uint32_t *>(dkp + 384 * K)[d] = w;
    }
    {   // dk tail = H(ek) || z (kyber.go:69-75, :189-201)
        const CoopLane c = coop_lane(coopw, lane);
        const uint64_t *ekw = reinterpret_cast<const uint64_t *>(ekl);
        uint32_t vlo, vhi;
        coop_sponge17<true>(vlo, vhi, [&](int k) { return ekw[k]; }, Gm::EK / 8, kDsSha3, c, j);
        uint64_t *tail = reinterpret_cast<uint64_t *>(dkp + 768 * K + 32);
        if (half == 0 && j < 4) tail[j] = ((uint64_t)vhi << 32) | vlo;
        else if (half == 0 && j < 8) tail[j] = reinterpret_cast<const uint64_t *>(sd + 32)[j - 4];
    }
}

}  // namespace mlkem
}  // namespace circl
