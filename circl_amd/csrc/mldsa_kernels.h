// mldsa_kernels.h -- batch ML-DSA verification for gfx950 (included by circl_hip.hip).
//
// sign/mldsa/mldsa65/internal/dilithium.go:273-332 Verify (+ :114-126 PublicKey.Unpack, which the
// reference caches per key but a batch of distinct keys must pay per item) as three launches:
//
//  mldsa_prep_kernel<MODE>    lane = item.  tr = SHAKE256(pk)[:64]; mu = SHAKE256(tr || M')[:64]
//                             with M' = 0 || len(ctx) || ctx || msg (mldsa65/dilithium.go:115-132);
//                             first SHAKE256(c~) block for SampleInBall.  -> workspace.
//  mldsa_verify_kernel<MODE>  persistent single-wavefront workgroups, each pulling groups of
//                             IT = 64 / (K L) items from a ticket counter:
//    phase A  lane = (item, i, j): ExpandA stream SHAKE128(rho || j || i), 23-bit rejection
//             (sample.go:92-123), accepted coefficients through a 16-slot LDS FIFO into the
//             stream's 768-byte row (24-bit coefficients) of the workgroup's global scratch slice (L2 / Infinity Cache).
//    per item, one polynomial per wavefront, everything in registers:
//    phase 1  decode z (norm check), z-hat[j] = NTT(z[j]), strict hint decoding,
//             c-hat = NTT(SampleInBall(c~)).
//    phase 2  w-hat[i] = sum_j A[i][j] o z-hat[j]: the Dilithium NTT splits completely, so MulHat
//             is coefficient-wise (poly.go:88-92); A rows come back as coalesced 16-byte loads.
//    phase 3  w - c-hat * NTT(t1 2^13), inverse NTT, UseHint, w1 bit-packing -> workspace.
//    LDS holds only the FIFOs / the staged signature, the relayout buffer and the hint bitmap
//    (6.6 KB), so occupancy is set by registers (4 waves per SIMD), not by LDS.
//  mldsa_final_kernel<MODE>   lane = item.  c' = SHAKE256(mu || w1)[:len(c~)], ok = (c' == c~)
//                             and no decoding failure.
#pragma once
#include "dilithium_dev.h"
#include "mlkem_kernels.h"

namespace circl {
namespace mldsa {

using dilithium::Q;
using mlkem::rows_acquire;
using mlkem::store_words;
using mlkem::xor_words;

template <int MODE> struct DP;
// MODE 44 / 65 / 87 = ML-DSA (sign/mldsa/mldsa*/internal/params.go); MODE 2 / 3 / 5 = round-3 Dilithium2/3/5
// (sign/dilithium/mode*/internal/params.go): NIST = false, 32-byte tr and c~.  NIST selects the ML-DSA domain
// separation: K, L appended to the key seed (dilithium.go:191-193), rnd in rho'' (:360-362), and the ctx-prefixed
// message of the outer package (mldsa65/dilithium.go:115-132; round 3 hashes the bare message, mode3/dilithium.go:54-75).
template <> struct DP<44> { static constexpr int K = 4, L = 4, ETA = 2, TAU = 39, OMEGA = 80, G1BITS = 17, CT = 32, TR = 64; static constexpr bool NIST = true; static constexpr uint32_t GAMMA2 = 95232; };
template <> struct DP<65> { static constexpr int K = 6, L = 5, ETA = 4, TAU = 49, OMEGA = 55, G1BITS = 19, CT = 48, TR = 64; static constexpr bool NIST = true; static constexpr uint32_t GAMMA2 = 261888; };
template <> struct DP<87> { static constexpr int K = 8, L = 7, ETA = 2, TAU = 60, OMEGA = 75, G1BITS = 19, CT = 64, TR = 64; static constexpr bool NIST = true; static constexpr uint32_t GAMMA2 = 261888; };
template <> struct DP<2> { static constexpr int K = 4, L = 4, ETA = 2, TAU = 39, OMEGA = 80, G1BITS = 17, CT = 32, TR = 32; static constexpr bool NIST = false; static constexpr uint32_t GAMMA2 = 95232; };
template <> struct DP<3> { static constexpr int K = 6, L = 5, ETA = 4, TAU = 49, OMEGA = 55, G1BITS = 19, CT = 32, TR = 32; static constexpr bool NIST = false; static constexpr uint32_t GAMMA2 = 261888; };
template <> struct DP<5> { static constexpr int K = 8, L = 7, ETA = 2, TAU = 60, OMEGA = 75, G1BITS = 19, CT = 32, TR = 32; static constexpr bool NIST = false; static constexpr uint32_t GAMMA2 = 261888; };

template <int MODE> struct DG {
    using P = DP<MODE>;
    static constexpr int K = P::K, L = P::L;
    static constexpr int PK = 32 + 320 * K;
    static constexpr int ZBITS = P::G1BITS + 1;
    static constexpr int ZSZ = 32 * ZBITS;
    static constexpr int SIG = P::CT + L * ZSZ + P::OMEGA + K;
    static constexpr int W1BITS = 23 - P::G1BITS;
    static constexpr int W1SZ = 32 * W1BITS;
    static constexpr uint32_t GAMMA1 = 1u << P::G1BITS;
    static constexpr uint32_t BETA = P::TAU * P::ETA;
    static constexpr int STREAMS = K * L;
    static constexpr int IT = 64 / STREAMS;            // items per workgroup
    static constexpr int MUW1 = 64 + K * W1SZ;         // bytes of mu || w1 per item in the workspace
    static constexpr int LDS_XCH = 4 * dilithium::kXchWords;  // padded exchange buffer (dilithium_dev.h relayout)
    static constexpr int LDS_MISC = 256;               // ball block bytes, positions
    // verify / keygen kernels: sampled matrix in global scratch
    static constexpr int FIFO_STRIDE = 80;             // bytes per lane the FIFO area is sized with (16 dword slots used, slot-major)
    static constexpr int LDS_FIFO = 64 * FIFO_STRIDE;  // phase A; afterwards the staged z || hint bytes
    static constexpr int LDS_HINT1 = K * 32;           // one item's hint bitmap
    static constexpr int LDS_V_TOTAL = LDS_FIFO + LDS_XCH + LDS_HINT1 + LDS_MISC;
    static constexpr int LDS_V_PAIR = LDS_V_TOTAL + LDS_XCH;  // the paired form (ABLATE bit 5, tools/ablate_dsa.hip only): a second exchange buffer behind the rest
    static constexpr int SCRATCH_BYTES = 64 * 768;     // 64 rows of 256 24-bit coefficients per workgroup
};

#ifndef CIRCL_DSA_WAVES_PER_EU
#define CIRCL_DSA_WAVES_PER_EU 4
#endif
// Issue priority of a wavefront of mldsa_verify_kernel while it is in phases 1-3 (0 = the same as phase A's).  Those phases are chains of
// LDS exchanges and loads with a few instructions in between -- 44 and 22 wavefront-cycles per VALU instruction against phase A's 10,
// profiles/r06_verify_clocks.txt --: with priority their instructions need not queue behind three co-resident wavefronts' Keccak rounds.
// Measured (profiles/r06_verify_variants.txt, alternating on one box): priority 1 or 3 alike, phases 1-3 shrink from 115 k to 79 k cycles
// per item, phase A grows from 122 k to 147 k (it now waits for them), the kernel gains 1.6 % (ML-DSA-65) / 2.2 % (ML-DSA-87).
#ifndef CIRCL_DSA_VERIFY_PRIO
#define CIRCL_DSA_VERIFY_PRIO 1
#endif

constexpr size_t kBallStateBytes = 200;

// SHAKE256 / SHA3-256 style absorb of NWORDS 64-bit words (rate 17 words) and final padding.
template <int NWORDS> __device__ __forceinline__ void sponge17_words(KeccakState &s, const uint64_t *p, uint32_t ds) {
    constexpr int FULL = NWORDS / 17, REM = NWORDS % 17;
    keccak_zero(s);
#pragma unroll 1
    for (int b = 0; b < FULL; b++) {
        xor_words<0, 17>(s, p + 17 * b);
        keccak_f1600(s);
    }
    xor_words<0, REM>(s, p + 17 * FULL);
    s.lo[REM] ^= ds;
    s.hi[16] ^= 0x80000000u;
    keccak_f1600(s);
}

// mu = SHAKE256(tr || M')[:64] with M' = 0 || len(ctx) || ctx || msg (mldsa65/dilithium.go:115-132;
// internal = the ACVP interface without the prefix).  On entry words 0..TRW-1 of h hold tr (TRW = 8, or 4 for
// round-3 Dilithium) and the rest is zero; on exit words 0..7 hold mu.  One sponge per lane, message lengths
// may differ per lane.
// The bytes of M' come from three places (prefix, context, message) at arbitrary alignment, so every absorbed word is
// gathered byte by byte.  Doing that inside the statically unrolled xor of the 17 rate words kept ~100 gathered bytes
// live next to the sponge state (204 VGPRs, 2 waves per SIMD; 328 + 156 spilled for round 3): the words of a block are
// therefore gathered in a rolled loop into the lane's row of an LDS staging area (kStageStride dwords apart: an odd
// stride, so the 64 rows of a wave hit distinct banks) and xor-ed into the state from there.
constexpr int kStageStride = 35;

// ---- long messages ------------------------------------------------------------------------------------------------------
// One sponge per lane means one LONG message keeps its lane -- and the 63 idle ones of its wavefront, and the whole launch --
// busy for a block per ~10 us: a 64 KB message among 2^14 32-byte ones made verification 8 x slower (tools/msglen_bench.py).
// Messages whose M' exceeds kLongMsg bytes are therefore listed by a scan kernel and, when there are at most kLongCap of them,
// hashed ahead of the per-lane kernels by mldsa_mu_long_kernel: two messages per wavefront on the two-state cooperative
// permutation (keccak_f1600_coop2, ~2.5 x shorter chain), every long message on its own half-wave in parallel.  The per-lane
// kernels then take mu ready-made for those items.  With more long messages than kLongCap the batch is throughput-bound and
// the per-lane form (64 sponges per wavefront) is the right one: the pre-pass does nothing.
constexpr size_t kLongMsg = 2048;
constexpr uint32_t kLongCap = 4096;
// A small batch is a latency chain like a long message: tr = H(pk) alone is 15 dependent permutations on one lane (~145 us)
// in front of everything else.  Batches of at most kSmallMu items send EVERY item through the pre-pass (tr and mu two items per
// wavefront on the cooperative permutation, ~4 us per block instead of ~10).
constexpr size_t kSmallMu = 1024;
static_assert(kSmallMu <= kLongCap, "a small batch lists every item");
struct LongCtl {
    uint32_t count;          // long messages found by the scan (may exceed kLongCap: then nothing is pre-hashed)
    uint32_t pad[63];
    uint32_t list[kLongCap]; // their item indices
};
__device__ __forceinline__ bool long_premade(const LongCtl *ctl, size_t total, size_t n) {
    return ctl && (total > kLongMsg || n <= kSmallMu) && ctl->count <= kLongCap;
}
__device__ __forceinline__ size_t mprime_total(const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off, int internal, size_t idx) {
    const size_t mlen = (size_t)(msg_off[idx + 1] - msg_off[idx]);
    const size_t clen = (ctx_blob && !internal) ? (size_t)(ctx_off[idx + 1] - ctx_off[idx]) : 0;
    return (internal ? 0 : 2) + clen + mlen;
}
// lane = item: which items have a long M'
static __global__ void __launch_bounds__(256) mldsa_long_scan_kernel(const uint64_t *__restrict__ msg_off, const uint8_t *__restrict__ ctx_blob,
                                                                     const uint64_t *__restrict__ ctx_off, int internal, size_t n, LongCtl *__restrict__ ctl) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    if (n <= kSmallMu) {  // a small batch lists every item: no counting (a thousand same-address atomics are ~20 us)
        ctl->list[idx] = (uint32_t)idx;
        if (idx == 0) ctl->count = (uint32_t)n;
        return;
    }
    if (mprime_total(msg_off, ctx_blob, ctx_off, internal, idx) > kLongMsg) {
        const uint32_t p = atomicAdd(&ctl->count, 1u);
        if (p < kLongCap) ctl->list[p] = (uint32_t)idx;
    }
}

// The eight bytes of M' = [0 || len(ctx) ||] ctx || msg at stream offset `base` (a multiple of 8), with the SHAKE suffix byte
// behind the last one and zeros beyond: aligned 64-bit loads and a funnel shift where the word lies wholly inside the message
// (the second aligned word of a misaligned load holds the word's own last byte, so nothing beyond the message's last 8-byte
// granule is read), byte by byte around the prefix / context and the end.
struct MPrime {
    const uint8_t *mp, *cp;
    size_t clen, total, m_begin;
    int internal;
    const uint64_t *pa;
    unsigned mis;
    __device__ __forceinline__ MPrime(const uint8_t *mp_, size_t mlen, const uint8_t *cp_, size_t clen_, int internal_)
        : mp(mp_), cp(cp_), clen(internal_ ? 0 : clen_), internal(internal_) {
        const size_t pre = internal ? 0 : 2;
        m_begin = pre + clen;
        total = m_begin + mlen;
        const uint8_t *vp = mp - m_begin;  // M' byte k of the message region is vp[k]
        mis = (unsigned)(reinterpret_cast<uintptr_t>(vp) & 7);
        pa = reinterpret_cast<const uint64_t *>(vp - mis);
    }
    __device__ __forceinline__ uint32_t byte(size_t k) const {
        if (k > total) return 0;
        if (k == total) return kDsShake;
        if (!internal) {
            if (k == 0) return 0;
            if (k == 1) return (uint32_t)(clen & 0xff);
            if (k < 2 + clen) return cp[k - 2];
            return mp[k - 2 - clen];
        }
        return mp[k];
    }
    __device__ __forceinline__ void word(size_t base, uint32_t &lo, uint32_t &hi) const {
        if (base >= m_begin && base + 8 <= total) {
            uint64_t v = pa[base >> 3];
            if (mis) v = (v >> (8 * mis)) | (pa[(base >> 3) + 1] << (64 - 8 * mis));
            lo = (uint32_t)v;
            hi = (uint32_t)(v >> 32);
            return;
        }
        lo = hi = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            lo |= byte(base + b) << (8 * b);
            hi |= byte(base + 4 + b) << (8 * b);
        }
    }
};

template <int TRW>
__device__ __forceinline__ void absorb_message_and_squeeze(KeccakState &h, const uint8_t *mp, size_t mlen, const uint8_t *cp,
                                                           size_t clen, int internal, uint32_t *stage) {
    const MPrime mpr(mp, mlen, cp, clen, internal);
    const size_t total = mpr.total;  // length of M'
    size_t pos = 0;   // M' bytes consumed
    int w0 = TRW;     // the first block's words 0..TRW-1 hold tr
    for (;;) {
#pragma unroll 1
        for (int w = w0; w < 17; w++) {
            uint32_t lo, hi;
            mpr.word(pos + 8 * (size_t)(w - w0), lo, hi);
            stage[2 * w] = lo;
            stage[2 * w + 1] = hi;
        }
        detail::static_for<0, 17>([&](auto ic) {
            constexpr int w = decltype(ic)::v;
            if (w >= w0) {
                h.lo[w] ^= stage[2 * w];
                h.hi[w] ^= stage[2 * w + 1];
            }
        });
        const size_t span = 8 * (size_t)(17 - w0);
        const bool last = total < pos + span;
        if (last) h.hi[16] ^= 0x80000000u;
        keccak_f1600(h);
        if (last) break;
        pos += span;
        w0 = 0;
    }
}

// mu = SHAKE256(tr || M')[:64] of the long messages the scan listed, two per wavefront (see kLongMsg above).  tr of item t:
// TRW words at tr_base + q * tr_stride with q = key_idx ? key_idx[t] : t (tr_stride 0: one key for the batch) -- or, when pk
// is given, SHAKE256(pk_t)[:8 TRW] computed here (pk_words 64-bit words at pk + t * pk_stride).  mu -> mu_out + t * mu_stride.
template <int TRW>
__global__ void __launch_bounds__(64) mldsa_mu_long_kernel(const uint8_t *__restrict__ tr_base, size_t tr_stride, const KeyIdx key_idx,
                                                          const uint8_t *__restrict__ pk, size_t pk_stride, int pk_words,
                                                          const uint8_t *__restrict__ msg_blob, const uint64_t *__restrict__ msg_off,
                                                          const uint8_t *__restrict__ ctx_blob, const uint64_t *__restrict__ ctx_off, int internal,
                                                          uint8_t *__restrict__ mu_out, size_t mu_stride, const LongCtl *__restrict__ ctl) {
    __shared__ uint64_t ws[100];
    const uint32_t count = ctl->count;
    if (count == 0 || count > kLongCap) return;
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    const CoopLane c = coop_lane(ws, lane);
#pragma unroll 1
    for (uint32_t pair = blockIdx.x; 2 * pair < count; pair += gridDim.x) {
        const uint32_t e = 2 * pair + (uint32_t)half;
        const bool live = e < count;
        const size_t idx = ctl->list[live ? e : count - 1];  // the odd one out is done twice, stored once
        uint32_t vlo = 0, vhi = 0;
        if (pk) {  // tr = SHAKE256(pk)[:TR] (dilithium.go:123-125)
            const uint64_t *pw = reinterpret_cast<const uint64_t *>(pk + idx * pk_stride);
            const int full = pk_words / 17, rem = pk_words % 17;
            uint64_t next = j < 17 ? pw[j] : 0;
#pragma unroll 1
            for (int b = 0; b < full; b++) {
                vlo ^= (uint32_t)next;
                vhi ^= (uint32_t)(next >> 32);
                next = (j < 17 && 17 * (b + 1) + j < pk_words) ? pw[17 * (b + 1) + j] : 0;
                keccak_f1600_coop2(vlo, vhi, c);
            }
            vlo ^= (uint32_t)next;
            vhi ^= (uint32_t)(next >> 32);
            if (j == rem) vlo ^= kDsShake;
            if (j == 16) vhi ^= 0x80000000u;
            keccak_f1600_coop2(vlo, vhi, c);
            if (j >= TRW) vlo = vhi = 0;  // the mu sponge starts from tr || 0
        } else if (j < TRW) {
            const size_t q = key_idx ? (size_t)key_idx[idx] : idx;
            const uint64_t w = reinterpret_cast<const uint64_t *>(tr_base + q * tr_stride)[j];
            vlo = (uint32_t)w;
            vhi = (uint32_t)(w >> 32);
        }
        const size_t mlen = (size_t)(msg_off[idx + 1] - msg_off[idx]);
        const uint8_t *cp = ctx_blob ? ctx_blob + ctx_off[idx] : nullptr;
        const size_t clen = ctx_blob ? (size_t)(ctx_off[idx + 1] - ctx_off[idx]) : 0;
        const MPrime mpr(msg_blob + msg_off[idx], mlen, cp, clen, internal);
        size_t pos = 0;
        int w0 = TRW;
        bool done = false;  // this half's mu is out (the other half may still have blocks to absorb: it keeps both permuting)
#pragma unroll 1
        for (;;) {
            if (!done && j >= w0 && j < 17) {
                uint32_t lo, hi;
                mpr.word(pos + 8 * (size_t)(j - w0), lo, hi);
                vlo ^= lo;
                vhi ^= hi;
            }
            const size_t span = 8 * (size_t)(17 - w0);
            const bool last = mpr.total < pos + span;
            if (!done && last && j == 16) vhi ^= 0x80000000u;
            keccak_f1600_coop2(vlo, vhi, c);
            if (!done && last) {
                if (live && j < 8) reinterpret_cast<uint64_t *>(mu_out + idx * mu_stride)[j] = ((uint64_t)vhi << 32) | vlo;
                done = true;
            }
            if (!__any(!done)) break;
            pos += span;
            w0 = 0;
        }
    }
}

// tr of the one public key of a shared-key batch (all lanes compute the same sponge)
template <int MODE>
__global__ void __launch_bounds__(64) mldsa_tr_kernel(const uint8_t *__restrict__ pk, uint8_t *__restrict__ tr_out) {
    // one hash per call in front of every mu: on the cooperative permutation (~4 us a block instead of ~10 for a lone lane)
    __shared__ uint64_t ws[100];
    const int lane = threadIdx.x, j = lane & 31;
    const CoopLane c = coop_lane(ws, lane);
    const uint64_t *pkw = reinterpret_cast<const uint64_t *>(pk);
    uint32_t vlo, vhi;
    mlkem::coop_sponge17(vlo, vhi, [&](int k) { return pkw[k]; }, DG<MODE>::PK / 8, kDsShake, c, j);
    if (lane < 8) reinterpret_cast<uint64_t *>(tr_out)[lane] = ((uint64_t)vhi << 32) | vlo;
}

// Key tables (grouped keys): tr of every table entry, lane = entry -> tr_out[j] (64-byte slots).  The reference keeps tr
// and A in the parsed PublicKey (internal/dilithium.go:114-126), i.e. once per key however many signatures it checks.
template <int MODE>
__global__ void __launch_bounds__(256) mldsa_tr_table_kernel(const uint8_t *__restrict__ pk_table, uint8_t *__restrict__ tr_out, size_t nkeys) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= nkeys) return;
    KeccakState s;
    sponge17_words<DG<MODE>::PK / 8>(s, reinterpret_cast<const uint64_t *>(pk_table + j * DG<MODE>::PK), kDsShake);
    store_words<0, 8>(reinterpret_cast<uint64_t *>(tr_out + j * 64), s);
}

// ---- kernel P ---------------------------------------------------------------------------------

template <int MODE>
__global__ void __launch_bounds__(256) mldsa_prep_kernel(const uint8_t *__restrict__ pk, const uint8_t *__restrict__ sig,
                                                         const uint8_t *__restrict__ msg_blob, const uint64_t *__restrict__ msg_off,
                                                         const uint8_t *__restrict__ ctx_blob, const uint64_t *__restrict__ ctx_off,
                                                         int internal, uint8_t *__restrict__ muw1_ws, uint8_t *__restrict__ ball_ws,
                                                         uint8_t *__restrict__ fail_ws, size_t n, const uint8_t *__restrict__ tr_shared,
                                                         const KeyIdx key_idx, const LongCtl *__restrict__ long_ctl,
                                                         size_t tr_item_stride = 0) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    __shared__ uint32_t stage_lds[256 * kStageStride];
    uint32_t *stage = stage_lds + threadIdx.x * kStageStride;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    KeccakState s;
    const bool ctx_unsupported = !P::NIST && !internal && ctx_blob && ctx_off[idx + 1] != ctx_off[idx];  // round 3 has no contexts
    if (!P::NIST) internal = 1;  // round 3: mu = CRH(tr || msg)
    // tr = SHAKE256(pk)[:TR]  (dilithium.go:123-125); shared-key batches bring it ready-made (mldsa_tr_kernel), key-table
    // batches one 64-byte slot per table entry (mldsa_tr_table_kernel), selected by key_idx.  ONE sponge state is live at
    // a time (tr, then mu, then the SampleInBall sponge reuse `s`): two states side by side cost 204 VGPRs = 2 waves per SIMD.
    const uint8_t *mp = msg_blob + msg_off[idx];
    const size_t mlen = (size_t)(msg_off[idx + 1] - msg_off[idx]);
    const uint8_t *cp = ctx_blob ? ctx_blob + ctx_off[idx] : nullptr;
    const size_t clen = ctx_blob ? (size_t)(ctx_off[idx + 1] - ctx_off[idx]) : 0;
    if (!long_premade(long_ctl, mprime_total(msg_off, ctx_blob, ctx_off, internal, idx), n)) {  // (otherwise mu is already there: mldsa_mu_long_kernel)
        if (tr_shared) {  // kernel-uniform
            keccak_zero(s);
            // (tr_item_stride != 0: every item's own tr, made by mldsa_tr_split_kernel -- it may lie in this item's ball slot, read here before that is written)
            xor_words<0, P::TR / 8>(s, reinterpret_cast<const uint64_t *>(tr_shared + (key_idx ? (size_t)key_idx[idx] * 64 : idx * tr_item_stride)));
        } else {
            sponge17_words<G::PK / 8>(s, reinterpret_cast<const uint64_t *>(pk + idx * G::PK), kDsShake);
#pragma unroll
            for (int i = P::TR / 8; i < 25; i++) { s.lo[i] = 0; s.hi[i] = 0; }
        }
        absorb_message_and_squeeze<P::TR / 8>(s, mp, mlen, cp, clen, internal, stage);
        store_words<0, 8>(reinterpret_cast<uint64_t *>(muw1_ws + idx * G::MUW1), s);  // mu
    }
    // SampleInBall's sponge: SHAKE256(c~), first block (sample.go:299-306); the whole state is
    // parked so that the verify kernel can squeeze further blocks in the (rare) case it must.
    const uint8_t *sg = sig + idx * G::SIG;
    keccak_zero(s);
    detail::static_for<0, P::CT / 8>([&](auto ic) {
        constexpr int w = decltype(ic)::v;
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v |= (uint64_t)sg[8 * w + b] << (8 * b);
        s.lo[w] = (uint32_t)v;
        s.hi[w] = (uint32_t)(v >> 32);
    });
    s.lo[P::CT / 8] ^= kDsShake;
    s.hi[16] ^= 0x80000000u;
    keccak_f1600(s);
    store_words<0, 25>(reinterpret_cast<uint64_t *>(ball_ws + idx * kBallStateBytes), s);
    // mldsa65/dilithium.go:116-118: a context longer than 255 bytes never verifies; round-3 Dilithium has no contexts at
    // all (sign.ErrContextNotSupported, sign/dilithium/mode3/dilithium.go:54-75), so a non-empty one never verifies either
    fail_ws[idx] = ((!internal && clen > 255) || ctx_unsupported) ? 1 : 0;
}

// ---- verify kernel helpers ----------------------------------------------------------------------

// D-bit field n of a little-endian bit stream (byte loads; never reads past the stream's end)
template <int D> __device__ __forceinline__ uint32_t get_bits32(const uint8_t *p, int n) {
    const int bit = n * D, b = bit >> 3, sh = bit & 7;
    uint32_t w = p[b];
    if (sh + D > 8) w |= (uint32_t)p[b + 1] << 8;
    if (sh + D > 16) w |= (uint32_t)p[b + 2] << 16;
    uint32_t hi = 0;
    if (sh + D > 24) hi = p[b + 3];
    return (uint32_t)(((((uint64_t)hi << 24) | w) >> sh) & ((1u << D) - 1));
}

// One squeezed SHAKE128 block of ExpandA: 56 candidates of 3 bytes, 23 bits each.
template <class F> __device__ __forceinline__ void for_each_candidate23(const KeccakState &s, F &&f) {
    detail::static_for<0, 56>([&](auto ic) {
        constexpr int c = decltype(ic)::v;
        constexpr int bit = 24 * c, w = bit / 32, sh = bit % 32;
        auto word = [&](int i) -> uint32_t { return (i & 1) ? s.hi[i >> 1] : s.lo[i >> 1]; };
        uint32_t v;
        if constexpr (sh <= 8) v = (word(w) >> sh) & 0x7fffffu;
        else v = alignbit(word(w + 1), word(w), sh) & 0x7fffffu;
        f(v);
    });
}

// Copies `nbytes` starting at the (arbitrarily aligned) global address `src` into LDS as aligned
// dwords: aligned 32-bit loads plus a funnel shift by the (wave-uniform) misalignment.  Never reads
// at or beyond src + nbytes rounded up to the enclosing aligned dword.
__device__ __forceinline__ void stage_unaligned(uint32_t *dst, const uint8_t *src, int nbytes, int lane) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(src) & 3;
    const uint32_t *base = reinterpret_cast<const uint32_t *>(src - a);
    const int ndw = (nbytes + 3) >> 2;                   // dwords to produce
    const int last = (int)((a + nbytes - 1) >> 2);       // last aligned dword that holds payload
    for (int d = lane; d < ndw; d += 64) {
        const uint32_t w0 = base[d];
        const uint32_t w1 = (d + 1 <= last) ? base[d + 1] : 0u;
        dst[d] = a ? alignbit(w1, w0, 8 * (uint32_t)a) : w0;
    }
}
// D-bit field n of a little-endian bit stream held as dwords in LDS (one dword of slack after the
// stream is required: the staging buffer is over-allocated)
template <int D> __device__ __forceinline__ uint32_t lds_bits(const uint32_t *p, int byte_off_dw, int n) {
    const int bit = n * D, w = byte_off_dw + (bit >> 5), sh = bit & 31;
    const uint32_t lo = p[w], hi = p[w + 1];
    return (sh ? alignbit(hi, lo, (uint32_t)sh) : lo) & ((1u << D) - 1);
}

// ExpandA for verification and key generation: lane = (item, i, j) runs the
// stream SHAKE128(rho || LE16((i << 8) + j)) (mat.go:15-49, sample.go:92-123).  Accepted coefficients go
// through a 16-slot LDS FIFO and leave four at a time, so that every global store is a full 16-byte
// segment of the stream's row (row index = lane).  Branch-free acceptance: the candidate is stored at
// slot cnt and cnt advances only if it is < q.
// Polynomials that batch signing parks in HBM and re-reads in every rejection round (the matrix rows, the NTT-domain
// secrets) are stored as 24-bit coefficients, 4 per 12 bytes in layout L4: 768 instead of 1024 bytes per polynomial
// cuts a quarter of that path's HBM traffic (values are < 2^24: matrix coefficients < q, secrets folded).
constexpr int kPackedRowDwords = 192;
__device__ __forceinline__ void pack24(uint32_t (&w)[3], uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    w[0] = c0 | (c1 << 24);
    w[1] = (c1 >> 8) | (c2 << 16);
    w[2] = (c2 >> 16) | (c3 << 8);
}
__device__ __forceinline__ void store_poly24(uint32_t *row, const uint32_t (&c)[4], int lane) {
    uint32_t w[3];
    pack24(w, c[0], c[1], c[2], c[3]);
    uint32_t *p = row + 3 * lane;
    p[0] = w[0]; p[1] = w[1]; p[2] = w[2];
}
__device__ __forceinline__ void load_poly24(uint32_t (&c)[4], const uint32_t *row, int lane) {
    const uint32_t *p = row + 3 * lane;
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
    c[0] = w0 & 0xffffffu;
    c[1] = (w0 >> 24) | ((w1 & 0xffffu) << 8);
    c[2] = (w1 >> 16) | ((w2 & 0xffu) << 16);
    c[3] = w2 >> 8;
}

// The 16-slot FIFOs of a wavefront are stored slot-major, fifo[slot][lane]: the acceptance rate is 0.999, so all 64 lanes
// write the same slot at the same time, and with a row per lane (80-byte stride: 8 distinct banks) every write was an
// 8-way bank conflict (SQ_LDS_BANK_CONFLICT 6.3e8 cycles per 2^18 verifications); slot-major, the lanes of a write hit
// consecutive banks.  A lane's FIFO is fifo[slot * kFifoLanes], with `fifo` already offset by its lane.
constexpr int kFifoLanes = 64;
template <bool TAIL, bool NOSTORE = false, bool PACK24 = false>
__device__ __forceinline__ void parse23_block_fifo(const KeccakState &s, uint32_t *fifo, uint32_t *row, int &cnt, int &flushed) {
    bool live = true;
    detail::static_for<0, 56>([&](auto ic) {
        constexpr int c = decltype(ic)::v;
        constexpr int bit = 24 * c, w = bit / 32, sh = bit % 32;
        auto word = [&](int i) -> uint32_t { return (i & 1) ? s.hi[i >> 1] : s.lo[i >> 1]; };
        if (!TAIL || live) {
            uint32_t v;
            if constexpr (sh <= 8) v = (word(w) >> sh) & 0x7fffffu;
            else v = alignbit(word(w + 1), word(w), sh) & 0x7fffffu;
            fifo[(cnt & 15) * kFifoLanes] = v;
            cnt += v < Q ? 1 : 0;
            if constexpr (c % 4 == 3) {
                cnt = min(cnt, 256);  // capped once per 4 candidates (as mlkem::parse_shake128_block_fifo: the overrun lands in free slots)
                if (cnt - flushed >= 4) {  // at most 7 pending here, so one flush per check suffices
                    const uint32_t *fr = fifo + (flushed & 15) * kFifoLanes;
                    const uint4 d = make_uint4(fr[0], fr[kFifoLanes], fr[2 * kFifoLanes], fr[3 * kFifoLanes]);
                    if constexpr (NOSTORE) { if (d.x == 0x7fffffffu) row[0] = d.y; }  // profiling aid: keep the LDS read, drop the store
                    else if constexpr (PACK24) {
                        uint32_t w[3];
                        pack24(w, d.x, d.y, d.z, d.w);
                        uint32_t *p = row + 3 * (flushed >> 2);
                        p[0] = w[0]; p[1] = w[1]; p[2] = w[2];
                    } else *reinterpret_cast<uint4 *>(row + flushed) = d;
                    flushed += 4;
                }
                if constexpr (TAIL) live = __any(flushed < 256);  // wave-uniform
            }
        }
    });
}

// NONCE_ARG (the unit-level primitive circl_hip_mldsa_sample_uniform only): lane = item, the stream's 16-bit nonce comes
// from nonces[item] instead of (i << 8) + j -- PolyDeriveUniform(p, seed, nonce) for arbitrary nonces, sample.go:92-123.
template <int MODE, bool NOSTORE = false, int GA = DG<MODE>::IT, bool NONCE_ARG = false>
__device__ __forceinline__ void expand_a_scratch(uint8_t *lds_fifo, uint32_t *rows, const uint8_t *__restrict__ rho, size_t rho_stride,
                                                 size_t item0, size_t n, int lane, const uint16_t *__restrict__ nonces = nullptr) {
    using G = DG<MODE>;
    constexpr int L = G::L;
    const bool on = NONCE_ARG ? item0 + lane < n : lane < GA * G::STREAMS;
    const int g = NONCE_ARG ? lane : on ? lane / G::STREAMS : 0, p = on ? lane % G::STREAMS : 0;
    const int i = p / L, j = p % L;
    size_t item = item0 + g;
    if (item >= n) item = n - 1;
    KeccakState s;
    keccak_zero(s);
    xor_words<0, 4>(s, reinterpret_cast<const uint64_t *>(rho + item * rho_stride));
    if constexpr (NONCE_ARG) s.lo[4] = (uint32_t)nonces[item] | (kDsShake << 16);
    else s.lo[4] = (uint32_t)j | ((uint32_t)i << 8) | (kDsShake << 16);
    s.hi[20] = 0x80000000u;
    uint32_t *fifo = reinterpret_cast<uint32_t *>(lds_fifo) + lane;  // slot-major (parse23_block_fifo)
    uint32_t *row = rows + lane * kPackedRowDwords;  // 24-bit packed rows (pack24)
    int cnt = on ? 0 : 256, flushed = cnt;
    // 256 coefficients need at least 5 blocks of 56 candidates; the acceptance rate is q / 2^23 = 0.999
#pragma unroll 1
    for (int blk = 0; blk < 5; blk++) {
        keccak_f1600(s);
        if (on) parse23_block_fifo<false, NOSTORE, true>(s, fifo, row, cnt, flushed);
    }
#pragma unroll 1
    while (__any(flushed < 256)) {  // more than 24 rejections in 280 candidates: essentially never
        keccak_f1600(s);
        parse23_block_fifo<true, NOSTORE, true>(s, fifo, row, cnt, flushed);
    }
}

// One row of the sampled matrix back from scratch, coefficients 4 lane .. 4 lane + 3 (layout L4).
// Plain loads, so that the compiler may batch and hoist them; the caller runs mlkem::rows_acquire() between
// phase A and the first load so that no L1 line left over from the previous group's rows is hit.
__device__ __forceinline__ void load_row_l4(uint32_t (&a)[4], const uint32_t *rows, int stream, int lane) {
    load_poly24(a, rows + stream * kPackedRowDwords, lane);
}
// out = 2^-32 sum_j A[stream0 + j] o vhat[j] mod q, in (0, 2q): the products accumulate un-reduced in 64 bits
// (V_MAD_U64_U32; a < 2^23, vhat < 2^28, L <= 7 terms: below 2^32 q) and are Montgomery-reduced once per coefficient.
// vhat is the plain transform (no 2^32 pre-scaling); callers drop the 2^-32 in their inverse transform (INV256_RR).
// The rows are fetched four at a time ahead of the multiplies, so that their L2 latencies overlap instead of adding up.
template <int L> __device__ __forceinline__ void mac_rows(uint32_t (&out)[4], const uint32_t *rows, int stream0, const uint32_t (&vhat)[L][4],
                                                         int lane) {
    uint64_t acc[4] = {0, 0, 0, 0};
    detail::static_for<0, (L + 3) / 4>([&](auto ic) {
        constexpr int j0 = 4 * decltype(ic)::v, CNT = L - j0 < 4 ? L - j0 : 4;
        uint32_t a[CNT][4];
#pragma unroll
        for (int j = 0; j < CNT; j++) load_row_l4(a[j], rows, stream0 + j0 + j, lane);
#pragma unroll
        for (int j = 0; j < CNT; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[r] += (uint64_t)a[j][r] * vhat[j0 + j][r];
    });
#pragma unroll
    for (int r = 0; r < 4; r++) out[r] = dilithium::mont64(acc[r]);
}
// Two rows of A z-hat at once (rows stream0 and stream1 of the matrix): the 2 L row loads are all issued before the first multiply, so
// that ONE memory round trip covers both dot products (the paired form of mldsa_verify_kernel).
template <int L> __device__ __forceinline__ void mac_rows2(uint32_t (&out0)[4], uint32_t (&out1)[4], const uint32_t *rows, int stream0, int stream1,
                                                          const uint32_t (&vhat)[L][4], int lane) {
    uint64_t acc0[4] = {0, 0, 0, 0}, acc1[4] = {0, 0, 0, 0};
    detail::static_for<0, (L + 3) / 4>([&](auto ic) {
        constexpr int j0 = 4 * decltype(ic)::v, CNT = L - j0 < 4 ? L - j0 : 4;
        uint32_t a[CNT][4], b[CNT][4];
#pragma unroll
        for (int j = 0; j < CNT; j++) { load_row_l4(a[j], rows, stream0 + j0 + j, lane); load_row_l4(b[j], rows, stream1 + j0 + j, lane); }
#pragma unroll
        for (int j = 0; j < CNT; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) { acc0[r] += (uint64_t)a[j][r] * vhat[j0 + j][r]; acc1[r] += (uint64_t)b[j][r] * vhat[j0 + j][r]; }
    });
#pragma unroll
    for (int r = 0; r < 4; r++) { out0[r] = dilithium::mont64(acc0[r]); out1[r] = dilithium::mont64(acc1[r]); }
}
// The positions j_t of SampleInBall the way the reference finds them (sample.go:299-339): step t = 0 .. tau-1 reads bytes
// until one is <= i_t = 256 - tau + t.  Returns j_t in lane t.  One step is three compares + ballots, scalar bit tricks and
// one v_readlane; a further SHAKE256 block is squeezed (keccak_f1600_coop: the wave works on the one state through LDS)
// when the current one runs out.  This sequential form is the fallback of sample_in_ball: the first block practically
// always holds tau acceptable bytes.  `kws` = 440 B of LDS for the sponge, `blk` >= 136 B for the squeezed block.
// NW: called from a workgroup of several wavefronts with buffers private to this wavefront (no workgroup barriers inside)
template <bool NW> __device__ __forceinline__ void ball_sync() {
    if constexpr (NW) { __builtin_amdgcn_s_waitcnt(0); wave_lds_order(); }
    else __syncthreads();
}
template <int MODE, bool NW = false>
__device__ __noinline__ uint32_t sample_in_ball_positions_sequential(const uint8_t *st, uint8_t *blk, uint64_t *kws, int lane) {
    using P = DP<MODE>;
    uint32_t b0 = st[lane], b1 = st[64 + lane], b2 = lane < 8 ? (uint32_t)st[128 + lane] : 0xfffu;
    int off = 8;      // next unread byte of the block
    uint32_t jt = 0;  // lane t keeps j_t
    bool have_state = false;
#pragma unroll 1
    for (int t = 0; t < P::TAU; t++) {
        const uint32_t i = 256 - P::TAU + t;
        int found = -1;
        uint32_t jv = 0;
#pragma unroll 1
        while (found < 0) {
            unsigned long long m0 = __ballot(b0 <= i), m1 = __ballot(b1 <= i), m2 = __ballot(b2 <= i);
            if (off >= 128) { m0 = 0; m1 = 0; m2 &= ~0ull << (off - 128); }
            else if (off >= 64) { m0 = 0; m1 &= ~0ull << (off - 64); }
            else m0 &= ~0ull << off;
            if (m0) { const int p = __ffsll((long long)m0) - 1; found = p; jv = (uint32_t)__builtin_amdgcn_readlane((int)b0, p); }
            else if (m1) { const int p = __ffsll((long long)m1) - 1; found = 64 + p; jv = (uint32_t)__builtin_amdgcn_readlane((int)b1, p); }
            else if (m2) { const int p = __ffsll((long long)m2) - 1; found = 128 + p; jv = (uint32_t)__builtin_amdgcn_readlane((int)b2, p); }
            else {
                // block exhausted (rare): squeeze the next one
                ball_sync<NW>();
                if (!have_state) {
                    if (lane < 25) kws[lane] = reinterpret_cast<const uint64_t *>(st)[lane];
                    have_state = true;
                }
                keccak_f1600_coop<NW>(kws, lane);
                if (lane < 17) reinterpret_cast<uint64_t *>(blk)[lane] = kws[lane];
                ball_sync<NW>();
                b0 = blk[lane]; b1 = blk[64 + lane]; b2 = lane < 8 ? (uint32_t)blk[128 + lane] : 0xfffu;
                off = 0;
            }
        }
        if (lane == t) jt = jv;
        off = found + 1;
    }
    return jt;
}

// SampleInBall (sample.go:299-339) followed by the NTT: c-hat in layout L4, times 2^32 if SCALED (so that
// mont32(x, c-hat) is the plain product) or plain (the product then carries 2^-32, like mac_rows' output).
// `st` = the 200-byte SHAKE256(c~) sponge state after its first permutation (global or LDS):
// 8 sign bytes, then bytes b <= i pick the positions.  `blk` (>= 136 B of LDS) is scratch.
//
// Which bytes the reference's sequential scan accepts is found for the whole block at once: byte p (p >= 8) is taken at
// step t_p = (number of bytes taken before p) iff b_p <= 256 - tau + t_p.  Starting from the bytes <= 256 - tau (taken at
// any step) and re-evaluating every byte against its current rank converges from below to exactly that set in a few
// ballot + popcount passes (a byte's rank only depends on earlier bytes, so the smallest position where the fixed point
// and the sequential scan could differ cannot exist); the t-th taken byte is j_t.  ~60 wave-instructions instead of
// tau dependent steps of ~15.  Fewer than tau taken bytes in the block (essentially never): the sequential fallback.
// FORCE_SEQUENTIAL (a parity-test aid of circl_hip_mldsa_sample_in_ball) always takes the fallback.
template <int MODE, bool SCALED = true, bool FORCE_SEQUENTIAL = false, bool WANT_HAT = true, bool NW = false>
__device__ __forceinline__ void sample_in_ball_hat(uint32_t (&chat)[4], const uint8_t *st, uint8_t *blk, uint32_t *xch,
                                                   const dilithium::LaneZetas &z, int lane) {
    using P = DP<MODE>;
    const unsigned long long signs = *reinterpret_cast<const unsigned long long *>(st);
    uint32_t jt = 0;  // lane t keeps j_t
    if constexpr (FORCE_SEQUENTIAL) {
        jt = sample_in_ball_positions_sequential<MODE, NW>(st, blk, reinterpret_cast<uint64_t *>(xch), lane);
    } else {
        constexpr uint32_t T0 = 256 - P::TAU;
        // positions lane (valid from 8 on), 64 + lane, 128 + lane (valid below 136): 0xfff never passes a test
        const uint32_t b0 = lane >= 8 ? (uint32_t)st[lane] : 0xfffu, b1 = st[64 + lane], b2 = lane < 8 ? (uint32_t)st[128 + lane] : 0xfffu;
        const unsigned long long below = (1ull << lane) - 1;
        unsigned long long m0 = __ballot(b0 <= T0), m1 = __ballot(b1 <= T0), m2 = __ballot(b2 <= T0);
        uint32_t r0, r1, r2;
        for (;;) {  // wave-uniform
            const uint32_t c0 = (uint32_t)__popcll(m0), c1 = (uint32_t)__popcll(m1);
            r0 = (uint32_t)__popcll(m0 & below);
            r1 = c0 + (uint32_t)__popcll(m1 & below);
            r2 = c0 + c1 + (uint32_t)__popcll(m2 & below);
            const unsigned long long n0 = __ballot(b0 <= T0 + r0), n1 = __ballot(b1 <= T0 + r1), n2 = __ballot(b2 <= T0 + r2);
            if (n0 == m0 && n1 == m1 && n2 == m2) break;
            m0 = n0; m1 = n1; m2 = n2;
        }
        if (__popcll(m0) + __popcll(m1) + __popcll(m2) >= P::TAU) {
            ball_sync<NW>();  // earlier users of blk are done
            if (((m0 >> lane) & 1) && r0 < (uint32_t)P::TAU) blk[r0] = (uint8_t)b0;
            if (((m1 >> lane) & 1) && r1 < (uint32_t)P::TAU) blk[r1] = (uint8_t)b1;
            if (((m2 >> lane) & 1) && r2 < (uint32_t)P::TAU) blk[r2] = (uint8_t)b2;
            ball_sync<NW>();
            jt = lane < P::TAU ? (uint32_t)blk[lane] : 0u;
        } else {
            jt = sample_in_ball_positions_sequential<MODE, NW>(st, blk, reinterpret_cast<uint64_t *>(xch), lane);  // (xch is free until the polynomial is built)
        }
    }
    // resolve the Fisher-Yates moves in parallel: the +-1 written at step t sits at j_t until a later
    // step t2 with j_t2 == (its current position) moves it to i_t2
    uint32_t pos = jt;
    for (int t2 = 1; t2 < P::TAU; t2++) {
        const uint32_t j2 = (uint32_t)__builtin_amdgcn_readlane((int)jt, t2);
        if (t2 > lane && j2 == pos) pos = 256 - P::TAU + t2;
    }
    uint32_t *cpoly = xch;
    ball_sync<NW>();
    for (int i = lane; i < 256; i += 64) cpoly[i] = 0;
    ball_sync<NW>();
    if (lane < P::TAU) cpoly[pos] = ((signs >> lane) & 1) ? Q - 1 : 1;
    ball_sync<NW>();
    uint32_t c[4];
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = cpoly[kyber::idx_l1(lane, r)];
    if constexpr (!WANT_HAT) {  // the polynomial itself, layout L1 (circl_hip_mldsa_sample_in_ball)
#pragma unroll
        for (int r = 0; r < 4; r++) chat[r] = c[r];
        return;
    }
    dilithium::ntt<NW>(c, z, xch, lane);
#pragma unroll
    for (int r = 0; r < 4; r++) chat[r] = SCALED ? dilithium::mont32(c[r], dilithium::R32SQ) : c[r];
}

// ---- primitive: PolyDeriveUniformBall (sample.go:299-339) of n challenge seeds, one polynomial per wavefront ----
template <int MODE>
__global__ void __launch_bounds__(64) mldsa_sample_in_ball_kernel(const uint8_t *__restrict__ ctilde, uint32_t *__restrict__ polys, int sequential) {
    using P = DP<MODE>;
    __shared__ __attribute__((aligned(16))) uint32_t xch[dilithium::kXchWords];
    __shared__ __attribute__((aligned(16))) uint8_t st[200];
    __shared__ __attribute__((aligned(16))) uint8_t blk[144];
    const int lane = threadIdx.x;
    const uint8_t *seed = ctilde + (size_t)blockIdx.x * P::CT;
    // SHAKE256(c~), first block, every lane the same sponge (a primitive for parity tests, not a hot path)
    KeccakState s;
    keccak_zero(s);
    detail::static_for<0, P::CT / 8>([&](auto ic) {
        constexpr int w = decltype(ic)::v;
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v |= (uint64_t)seed[8 * w + b] << (8 * b);
        s.lo[w] = (uint32_t)v;
        s.hi[w] = (uint32_t)(v >> 32);
    });
    s.lo[P::CT / 8] ^= kDsShake;
    s.hi[16] ^= 0x80000000u;
    keccak_f1600(s);
    if (lane == 0) store_words<0, 25>(reinterpret_cast<uint64_t *>(st), s);
    __syncthreads();
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);
    uint32_t c[4];
    if (sequential) sample_in_ball_hat<MODE, true, true, false>(c, st, blk, xch, z, lane);
    else sample_in_ball_hat<MODE, true, false, false>(c, st, blk, xch, z, lane);
#pragma unroll
    for (int r = 0; r < 4; r++) polys[(size_t)blockIdx.x * 256 + kyber::idx_l1(lane, r)] = c[r];
}

// ---- kernel V -----------------------------------------------------------------------------------

// ABLATE is a profiling aid (tools/ablate_dsa.hip): bit 0 skips phase A, bit 1 phase 1, bit 2 phases 2+3,
// bit 3 drops the row stores of phase A, bit 4 shrinks the row loads of phase 2 to one row (L2 hits).
// Bit 5 (32) is not an ablation but the PAIRED form of phases 1-3 (round 6's structural attempt, measured at a LOSS of 3 / 10 / 5 % for
// ML-DSA-44 / 65 / 87 -- profiles/r06_verify_ab.txt -- and therefore instantiated by tools/ablate_dsa.hip only, not by the library): the transforms
// run two polynomials at a time (dilithium::ntt2 / invntt2: every LDS exchange wait covers two transforms), the rows of two
// output polynomials are fetched by one batch of loads (mac_rows2), and an odd last z polynomial shares its transform with the
// challenge c.  Same arithmetic, same bytes.
// Bit 6 (64): the wavefront reads the shader clock at every phase boundary of the FULL kernel and leaves, per workgroup, the cycles it
// spent in { ticket, phase A, phase 1, phases 2+3 } and its item count at key_rows (which the per-item form does not use otherwise):
// where the wall time of a wavefront goes when the phases of four co-resident wavefronts overlap (profiles/r06_verify_clocks.txt).
// `scratch` holds gridDim.x slices of DG::SCRATCH_BYTES; `work` is the ticket counter (zeroed by the host)
// or nullptr for one group per workgroup.
// KM = mlkem::KM_SHARED: every item is verified under the ONE public key at `pk` (the reference's cached-key case: A and tr
// live in the parsed PublicKey, internal/dilithium.go:114-126): ExpandA runs once per workgroup before the group loop.
// KM = mlkem::KM_KEYED: item t is verified under entry key_idx[t] of the key table at `pk`; the matrices of all entries
// were expanded beforehand into key_rows (mldsa_expand_keys_kernel: K L packed rows per entry) and are read-only here.
template <int MODE, int ABLATE = 0, int KM = mlkem::KM_ITEM>
__global__ void __launch_bounds__(64, CIRCL_DSA_WAVES_PER_EU)
    mldsa_verify_kernel(const uint8_t *__restrict__ pk, const uint8_t *__restrict__ sig, uint8_t *__restrict__ muw1_ws,
                        const uint8_t *__restrict__ ball_ws, uint8_t *__restrict__ fail_ws, uint8_t *__restrict__ scratch,
                        unsigned *__restrict__ work, size_t n, const KeyIdx key_idx, const uint32_t *__restrict__ key_rows) {
    constexpr bool SHARED = KM == mlkem::KM_SHARED, KEYED = KM == mlkem::KM_KEYED;
    using G = DG<MODE>;
    using P = DP<MODE>;
    constexpr int K = P::K, L = P::L;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *stg = reinterpret_cast<uint32_t *>(smem);  // aliases the FIFOs of phase A
    uint32_t *xch = reinterpret_cast<uint32_t *>(smem + G::LDS_FIFO);
    uint32_t *hintbits = reinterpret_cast<uint32_t *>(smem + G::LDS_FIFO + G::LDS_XCH);
    uint8_t *misc = smem + G::LDS_FIFO + G::LDS_XCH + G::LDS_HINT1;
    uint32_t *xch1 = reinterpret_cast<uint32_t *>(smem + G::LDS_FIFO + G::LDS_XCH + G::LDS_HINT1 + G::LDS_MISC);  // the paired form's second exchange buffer
    constexpr bool PAIR = (ABLATE & 32) != 0;
    const int lane = threadIdx.x;
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);
    uint32_t *rows = reinterpret_cast<uint32_t *>(scratch + (size_t)blockIdx.x * G::SCRATCH_BYTES);
    const size_t ngroups = (n + G::IT - 1) / G::IT;
    constexpr int ZH_BYTES = L * G::ZSZ + P::OMEGA + K;
    static_assert(ZH_BYTES + 8 <= G::LDS_FIFO, "staged z || hint fits in the FIFO area");
    constexpr size_t PK_STRIDE = SHARED ? 0 : G::PK;
    if constexpr (KEYED) rows = nullptr;
    if constexpr (SHARED) {
        expand_a_scratch<MODE, false, 1>(smem, rows, pk, 0, 0, 1, lane);  // rows 0 .. K L - 1, once
        rows_acquire();
    }
    constexpr bool CLK = (ABLATE & 64) != 0;
    uint64_t clk[5] = {0, 0, 0, 0, 0}, tprev = 0;
    auto lap = [&](int slot) {  // the cycles since the previous boundary go to `slot`
        if constexpr (CLK) {
            const uint64_t t = __builtin_readcyclecounter();
            clk[slot] += t - tprev;
            tprev = t;
        }
    };
    if constexpr (CLK) tprev = __builtin_readcyclecounter();

#pragma unroll 1
  for (size_t grp = mlkem::next_group(work, lane, true, ngroups); grp < ngroups; grp = mlkem::next_group(work, lane, false, ngroups)) {
    const size_t item0 = grp * G::IT;
    lap(0);
    if constexpr (!SHARED && !KEYED) {
        // ------------------------------ phase A ------------------------------
        __syncthreads();  // the previous group is done with the LDS the FIFOs alias
        if constexpr (CIRCL_DSA_VERIFY_PRIO != 0) __builtin_amdgcn_s_setprio(0);
        if (!(ABLATE & 1)) expand_a_scratch<MODE, (ABLATE & 8) != 0>(smem, rows, pk, (size_t)G::PK, item0, n, lane);
        rows_acquire();
        if constexpr (CIRCL_DSA_VERIFY_PRIO != 0) __builtin_amdgcn_s_setprio(CIRCL_DSA_VERIFY_PRIO);
        lap(1);
    }

#pragma unroll 1
    for (int g = 0; g < G::IT; g++) {
        const size_t item = item0 + g;
        if (item >= n) break;  // wave-uniform
        const size_t kq = KEYED ? (key_idx ? (size_t)key_idx[item] : size_t(0)) : item;  // wave-uniform; a table without an index vector: entry 0 for every item
        const uint32_t *irows = KEYED ? key_rows + kq * (size_t)(G::STREAMS * kPackedRowDwords) : rows;
        // ------------------------------ phase 1 ------------------------------
        uint32_t zhat[L][4], chat[4] = {0, 0, 0, 0};
        bool bad = false;
        if (!(ABLATE & 2)) {
            const uint8_t *sg = sig + item * G::SIG;
            constexpr bool C_WITH_Z = PAIR && (L & 1);  // the challenge's transform pairs up with the odd last z polynomial
            if constexpr (C_WITH_Z) sample_in_ball_hat<MODE, false, false, false>(chat, ball_ws + item * kBallStateBytes, misc, xch, z, lane);  // c itself, layout L1
            else sample_in_ball_hat<MODE, false>(chat, ball_ws + item * kBallStateBytes, misc, xch, z, lane);  // first: z-hat is not live yet
            __syncthreads();
            if (lane < K * 8) hintbits[lane] = 0;
            stage_unaligned(stg, sg + P::CT, ZH_BYTES, lane);
            if (lane == 0) stg[(ZH_BYTES + 3) >> 2] = 0;  // slack dword for lds_bits
            __syncthreads();
            // z: (gamma1_bits+1)-bit fields, value gamma1 - field (pack.go:146-199); ||z||inf < gamma1 - beta
            auto unpack_z = [&](uint32_t (&c)[4], int j) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t f = lds_bits<G::ZBITS>(stg, j * (G::ZSZ / 4), kyber::idx_l1(lane, r));
                    uint32_t x = G::GAMMA1 - f;
                    x += (uint32_t)((int32_t)x >> 31) & Q;
                    bad |= dilithium::exceeds(x, G::GAMMA1 - G::BETA);
                    c[r] = x;
                }
            };
            if constexpr (PAIR) {
                detail::static_for<0, L / 2>([&](auto ic) {
                    constexpr int j = 2 * decltype(ic)::v;
                    uint32_t c0[4], c1[4];
                    unpack_z(c0, j);
                    unpack_z(c1, j + 1);
                    dilithium::ntt2(c0, c1, z, xch, xch1, lane);
#pragma unroll
                    for (int r = 0; r < 4; r++) { zhat[j][r] = c0[r]; zhat[j + 1][r] = c1[r]; }
                });
                if constexpr (C_WITH_Z) {
                    uint32_t c0[4];
                    unpack_z(c0, L - 1);
                    dilithium::ntt2(c0, chat, z, xch, xch1, lane);  // plain c-hat (the products then carry 2^-32, like mac_rows' output)
#pragma unroll
                    for (int r = 0; r < 4; r++) zhat[L - 1][r] = c0[r];
                }
            } else {
#pragma unroll
                for (int j = 0; j < L; j++) {
                    uint32_t c[4];
                    unpack_z(c, j);
                    dilithium::ntt(c, z, xch, lane);
#pragma unroll
                    for (int r = 0; r < 4; r++) zhat[j][r] = c[r];  // plain z-hat, < 17q
                }
            }
            // hints: strict decoding (pack.go:113-141)
            {
                const uint8_t *hb = reinterpret_cast<const uint8_t *>(stg) + L * G::ZSZ;
                uint32_t sop[K];
#pragma unroll
                for (int i = 0; i < K; i++) sop[i] = hb[P::OMEGA + i];
#pragma unroll
                for (int i = 0; i < K; i++) bad |= sop[i] > (uint32_t)P::OMEGA || (i > 0 && sop[i] < sop[i - 1]);
                for (int j0 = 0; j0 < P::OMEGA; j0 += 64) {
                    const int j = j0 + lane;
                    if (j < P::OMEGA) {
                        int poly = 0;
                        uint32_t start = 0;
#pragma unroll
                        for (int i = 0; i < K; i++)
                            if (sop[i] <= (uint32_t)j) { poly = i + 1; start = sop[i]; }
                        const uint32_t v = hb[j];
                        if (poly < K) {
                            if ((uint32_t)j > start && v <= hb[j - 1]) bad = true;
                            atomicOr(&hintbits[poly * 8 + (v >> 5)], 1u << (v & 31));
                        } else if (v != 0) {
                            bad = true;
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < L; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) zhat[j][r] = (uint32_t)(lane + j + r);
        }
        const bool failed = __any(bad);
        lap(2);

        // -------------------------- phases 2 and 3 --------------------------
        uint8_t *w1out = muw1_ws + item * G::MUW1 + 64;
        // t1 (pack.go:52-66, 10-bit fields): coefficients 4 lane .. 4 lane + 3 are the 5 bytes at 5 lane, fetched as two aligned
        // dwords (the row is 4-byte aligned)
        auto unpack_t1 = [&](uint32_t (&t)[4], int i) {
            const uint32_t *tp = reinterpret_cast<const uint32_t *>(pk + kq * PK_STRIDE + 32 + 320 * i);
            const int b = 5 * lane, d = b >> 2;
            const uint64_t v = (((uint64_t)tp[d + 1] << 32) | tp[d]) >> (8 * (b & 3));
#pragma unroll
            for (int r = 0; r < 4; r++) t[r] = ((uint32_t)(v >> (10 * r)) & 0x3ffu) << dilithium::D;
        };
        auto hinted_w1 = [&](unsigned (&w1v)[4], const uint32_t (&w)[4], int i) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nidx = kyber::idx_l1(lane, r);
                const uint32_t hbit = (hintbits[i * 8 + (nidx >> 5)] >> (nidx & 31)) & 1;
                w1v[r] = dilithium::use_hint<P::GAMMA2>(dilithium::csubq(w[r]), hbit);
            }
        };
        const int row0 = (SHARED || KEYED) ? 0 : g * G::STREAMS;
        if constexpr (PAIR) {
            static_assert(K % 2 == 0, "the paired form takes the rows two at a time");
#pragma unroll 1
            for (int i = 0; i < ((ABLATE & 4) ? 0 : K); i += 2) {
                uint32_t acc0[4], acc1[4], t0[4], t1[4], w0[4], w1[4];
                mac_rows2<L>(acc0, acc1, irows, row0 + i * L, row0 + (i + 1) * L, zhat, lane);  // 2^-32 A z-hat, < 2q
                unpack_t1(t0, i);
                unpack_t1(t1, i + 1);
                dilithium::relayout2<4, 1>(t0, t1, xch, xch1, lane);
                dilithium::ntt2(t0, t1, z, xch, xch1, lane);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    w0[r] = dilithium::fold(acc0[r] + 2 * Q - dilithium::mont32(t0[r], chat[r]));  // as below
                    w1[r] = dilithium::fold(acc1[r] + 2 * Q - dilithium::mont32(t1[r], chat[r]));
                }
                dilithium::invntt2<dilithium::INV256_RR>(w0, w1, z, xch, xch1, lane);
                unsigned v0[4], v1[4];
                hinted_w1(v0, w0, i);
                hinted_w1(v1, w1, i + 1);
                mlkem::stage_bits_l1<G::W1BITS>(xch, v0, lane);
                mlkem::stage_bits_l1<G::W1BITS>(xch1, v1, lane);
                mlkem::store_staged<G::W1BITS>(reinterpret_cast<uint32_t *>(w1out + G::W1SZ * i), xch, lane, false);
                mlkem::store_staged<G::W1BITS>(reinterpret_cast<uint32_t *>(w1out + G::W1SZ * (i + 1)), xch1, lane, false);
            }
        } else {
#pragma unroll 1
        for (int i = 0; i < ((ABLATE & 4) ? 0 : K); i++) {
            uint32_t acc[4] = {0, 0, 0, 0};
            mac_rows<L>(acc, irows, (ABLATE & 16) ? 0 : row0 + i * L, zhat, lane);  // 2^-32 A z-hat, < 2q
            uint32_t t[4], w[4];
            unpack_t1(t, i);
            dilithium::relayout<4, 1>(t, xch, lane);  // to the NTT's input layout
            dilithium::ntt(t, z, xch, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t ct1 = dilithium::mont32(t[r], chat[r]);  // 2^-32 c-hat * t1-hat, < 2q
                w[r] = dilithium::fold(acc[r] + 2 * Q - ct1);           // < 4q, folded below 2q for the inverse transform
            }
            dilithium::invntt<dilithium::INV256_RR>(w, z, xch, lane);  // both terms carried 2^-32
            unsigned w1v[4];
            hinted_w1(w1v, w, i);
            mlkem::stage_bits_l1<G::W1BITS>(xch, w1v, lane);
            mlkem::store_staged<G::W1BITS>(reinterpret_cast<uint32_t *>(w1out + G::W1SZ * i), xch, lane, false);
        }
        }
        if (lane == 0 && failed) fail_ws[item] = 1;
        if constexpr (CLK) { lap(3); clk[4]++; }
    }
  }
    if constexpr (CLK) {
        lap(0);  // (the last, empty-handed ticket)
        if (lane == 0) {
            uint64_t *prof = reinterpret_cast<uint64_t *>(const_cast<uint32_t *>(key_rows)) + (size_t)blockIdx.x * 5;
#pragma unroll
            for (int k = 0; k < 5; k++) prof[k] = clk[k];
        }
    }
}

// Key tables: ExpandA of every table entry, once.  Single-wave workgroups, IT entries each; entry e gets rows
// e K L .. e K L + K L - 1 of the cache (24-bit packed).  The cache holds a whole number of groups.
template <int MODE>
__global__ void __launch_bounds__(64, CIRCL_DSA_WAVES_PER_EU)
    mldsa_expand_keys_kernel(const uint8_t *__restrict__ pk_table, uint32_t *__restrict__ key_rows, size_t nkeys) {
    using G = DG<MODE>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const size_t e0 = (size_t)blockIdx.x * G::IT;
    expand_a_scratch<MODE>(smem, key_rows + e0 * (size_t)(G::STREAMS * kPackedRowDwords), pk_table, (size_t)G::PK, e0, nkeys, threadIdx.x);
}

// c' = SHAKE256(mu || w1)[:CT] against the signature's c~, two items per wavefront on the cooperative permutation: the small-batch
// form of mldsa_final_kernel below (seven dependent permutations on one lane are ~68 us; here ~28)
template <int MODE>
__global__ void __launch_bounds__(64) mldsa_final_coop_kernel(const uint8_t *__restrict__ sig, const uint8_t *__restrict__ muw1_ws,
                                                             const uint8_t *__restrict__ fail_ws, uint8_t *__restrict__ ok, size_t n) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    __shared__ uint64_t ws[100];
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    const CoopLane c = coop_lane(ws, lane);
    size_t idx = 2 * (size_t)blockIdx.x + half;
    const bool live = idx < n;
    if (!live) idx = n - 1;
    const uint64_t *mw = reinterpret_cast<const uint64_t *>(muw1_ws + idx * G::MUW1);
    uint32_t vlo, vhi;
    mlkem::coop_sponge17(vlo, vhi, [&](int k) { return mw[k]; }, G::MUW1 / 8, kDsShake, c, j);
    const uint8_t *sg = sig + idx * G::SIG;
    bool mine = true;
    if (j < P::CT / 8) {
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v |= (uint64_t)sg[8 * j + b] << (8 * b);
        mine = v == (((uint64_t)vhi << 32) | vlo);
    }
    const unsigned long long bad = __ballot(!mine);
    const bool same = ((bad >> (32 * half)) & 0xffffffffull) == 0;
    if (live && j == 0) ok[idx] = (same && fail_ws[idx] == 0) ? 1 : 0;
}

// ---- resident keys, small batches: the whole verification of an item as ONE workgroup of K + 1 wavefronts ----------------------
// With the key parsed beforehand (a key table: the expanded matrix rows and tr) a verification (dilithium.go:273-332) is two chains
// that meet twice:
//   wave K ("hash")  mu = SHAKE256(tr || M') and the first block of SHAKE256(c~) side by side on the two-state cooperative
//                    permutation -> SampleInBall -> c-hat; the strict hint decoding                         | barrier |
//   waves j < L      z_j: decode, norm check, NTT (a polynomial per wavefront)                               | barrier |
//   waves i < K      row i: w1_i = UseHint(InvNTT(sum_j A_ij z-hat_j - c-hat t1-hat_i 2^13)), packed into LDS behind mu | barrier |
//   wave K           c' = SHAKE256(mu || w1) on the cooperative permutation (7 dependent permutations for ML-DSA-65), compare, ok
// ONE launch instead of the five of the small-batch route (scan, mu, prep, verify, final) and a side stream.  Every LDS buffer
// between the barriers belongs to one wavefront and every wave-level ordering point inside is the no-wait form.  Grid = n workgroups.
// Messages of any length (the hash wave walks M' block by block: what mldsa_mu_long_kernel does for a small batch anyway).
// RESIDENT = false: keys that are NOT parsed beforehand (circl_hip_mldsa_verify on a small batch: item t has its own public key at
// pk_table + t PK).  The hash wave first computes tr = SHAKE256(pk) (15 permutations for ML-DSA-65, cooperative), and one more
// wavefront expands the matrix meanwhile (a stream per lane, the rows into the item's part of the workspace's scratch slices --
// dilithium.go:114-126 does both in PublicKey.Unpack); the first barrier also publishes the rows.  K + 2 wavefronts.  pk_stride = 0:
// ONE unparsed key for the whole (small) batch -- every workgroup derives it again, which is shorter than a launch that derives it once.
template <int MODE, bool RESIDENT = true>
__global__ void __launch_bounds__((DP<MODE>::K + (RESIDENT ? 1 : 2)) * 64)
    mldsa_verify_chain_kernel(const uint8_t *__restrict__ pk_table, const KeyIdx key_idx, const uint32_t *__restrict__ key_rows,
                              const uint8_t *__restrict__ key_tr, const uint8_t *__restrict__ sig, const uint8_t *__restrict__ msg_blob,
                              const uint64_t *__restrict__ msg_off, const uint8_t *__restrict__ ctx_blob, const uint64_t *__restrict__ ctx_off,
                              int internal, uint8_t *__restrict__ ok, size_t n, uint8_t *__restrict__ scratch, size_t pk_stride,
                              const TailFlag tail = TailFlag{nullptr, nullptr, 0}) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    constexpr int K = P::K, L = P::L, TRW = P::TR / 8, HB = P::OMEGA + K;
    constexpr int ZST_DW = G::ZSZ / 4 + 2, HST_DW = (HB + 3) / 4 + 2;
    __shared__ __attribute__((aligned(16))) uint64_t coopw[100];
    __shared__ __attribute__((aligned(16))) uint32_t xch_all[K + 1][dilithium::kXchWords];
    __shared__ __attribute__((aligned(16))) uint32_t zst[L][ZST_DW];
    __shared__ __attribute__((aligned(16))) uint32_t hst[HST_DW];
    __shared__ __attribute__((aligned(16))) uint32_t zhat_lds[L][256];
    __shared__ __attribute__((aligned(16))) uint32_t chat_lds[256];
    __shared__ uint32_t hintbits[K * 8];
    __shared__ __attribute__((aligned(16))) uint64_t muw1[G::MUW1 / 8];
    __shared__ __attribute__((aligned(16))) uint64_t ballst[25];
    __shared__ __attribute__((aligned(16))) uint8_t blk[144];
    __shared__ uint32_t bad_w[K + 1];
    __shared__ __attribute__((aligned(16))) uint8_t fifo_lds[RESIDENT ? 16 : G::LDS_FIFO];
    static_assert(G::MUW1 % 8 == 0 && L <= K && G::PK % 8 == 0, "mu || w1 and pk are absorbed as 64-bit words; a wavefront per z polynomial");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t item = blockIdx.x;
    if (item >= n) return;  // (block-uniform)
    const size_t kq = RESIDENT ? (key_idx ? (size_t)key_idx[item] : size_t(0)) : item;
    // the item's matrix rows: the table's, or its own part of a scratch slice (IT items of STREAMS rows per slice)
    uint32_t *my_rows = RESIDENT ? nullptr
                                 : reinterpret_cast<uint32_t *>(scratch + (item / G::IT) * (size_t)G::SCRATCH_BYTES) + (item % G::IT) * (size_t)(G::STREAMS * kPackedRowDwords);
    const uint8_t *sg = sig + item * G::SIG;
    uint32_t *xch = xch_all[wave <= K ? wave : 0];  // (the expanding wavefront of unparsed keys has no transform to run)
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);
    auto handoff = [] {  // what the lanes wrote to LDS (by whatever instruction) is visible to the wavefront's later reads
        __builtin_amdgcn_s_waitcnt(0);
        wave_lds_order();
    };
    bool bad = false;
    if (wave == K) {
        // ---- mu (half 0) and the SampleInBall sponge's first block (half 1) ----
        const int half = lane >> 5, j = lane & 31;
        const CoopLane c = coop_lane(coopw, lane);
        const size_t mlen = (size_t)(msg_off[item + 1] - msg_off[item]);
        const uint8_t *cp = ctx_blob ? ctx_blob + ctx_off[item] : nullptr;
        const size_t clen = ctx_blob ? (size_t)(ctx_off[item + 1] - ctx_off[item]) : 0;
        const bool ctx_unsupported = !P::NIST && !internal && ctx_blob && clen != 0;  // round 3 has no contexts
        const int internal_eff = P::NIST ? internal : 1;                              // round 3: mu = CRH(tr || msg)
        bad = (!internal_eff && clen > 255) || ctx_unsupported;                       // mldsa65/dilithium.go:116-118
        uint32_t vlo = 0, vhi = 0;
        bool first = true;
        auto park_ball = [&] {  // half 1 after ITS first permutation: the whole state (further blocks are squeezed from it)
            if (first && half == 1 && j < 25) ballst[j] = ((uint64_t)vhi << 32) | vlo;
            first = false;
        };
        if (half == 0) {
            if (RESIDENT && j < TRW) {
                const uint64_t w = reinterpret_cast<const uint64_t *>(key_tr + kq * 64)[j];
                vlo = (uint32_t)w;
                vhi = (uint32_t)(w >> 32);
            }
        } else {
            if (j < P::CT / 8) {
                uint64_t v = 0;
                for (int b = 0; b < 8; b++) v |= (uint64_t)sg[8 * j + b] << (8 * b);
                vlo = (uint32_t)v;
                vhi = (uint32_t)(v >> 32);
            }
            if (j == P::CT / 8) vlo ^= kDsShake;
            if (j == 16) vhi ^= 0x80000000u;
        }
        if constexpr (!RESIDENT) {  // tr = SHAKE256(pk)[:TR] (dilithium.go:123-125) on half 0
            const uint64_t *pw = reinterpret_cast<const uint64_t *>(pk_table + kq * pk_stride);
            constexpr int PKW = G::PK / 8, FULL = PKW / 17, REM = PKW % 17;
            uint64_t next = (half == 0 && j < 17) ? pw[j] : 0;
#pragma unroll 1
            for (int b = 0; b < FULL; b++) {
                vlo ^= (uint32_t)next;
                vhi ^= (uint32_t)(next >> 32);
                next = (half == 0 && j < 17 && 17 * (b + 1) + j < PKW) ? pw[17 * (b + 1) + j] : 0;
                keccak_f1600_coop2<true>(vlo, vhi, c);
                park_ball();
            }
            vlo ^= (uint32_t)next;
            vhi ^= (uint32_t)(next >> 32);
            if (half == 0 && j == REM) vlo ^= kDsShake;
            if (half == 0 && j == 16) vhi ^= 0x80000000u;
            keccak_f1600_coop2<true>(vlo, vhi, c);
            park_ball();
            if (half == 0 && j >= TRW) vlo = vhi = 0;  // the mu sponge starts from tr || 0
        }
        const MPrime mpr(msg_blob + msg_off[item], mlen, cp, clen, internal_eff);
        size_t pos = 0;
        int w0 = TRW;
#pragma unroll 1
        for (;;) {
            if (half == 0 && j >= w0 && j < 17) {
                uint32_t lo, hi;
                mpr.word(pos + 8 * (size_t)(j - w0), lo, hi);
                vlo ^= lo;
                vhi ^= hi;
            }
            const size_t span = 8 * (size_t)(17 - w0);
            const bool last = mpr.total < pos + span;  // (wave-uniform: one message)
            if (half == 0 && last && j == 16) vhi ^= 0x80000000u;
            keccak_f1600_coop2<true>(vlo, vhi, c);
            park_ball();
            if (last) break;
            pos += span;
            w0 = 0;
        }
        if (half == 0 && j < 8) muw1[j] = ((uint64_t)vhi << 32) | vlo;
        handoff();
        // ---- c-hat ----
        uint32_t chat[4];
        sample_in_ball_hat<MODE, false, false, true, true>(chat, reinterpret_cast<const uint8_t *>(ballst), blk, xch, z, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) chat_lds[64 * r + lane] = chat[r];
        // ---- hints: strict decoding (pack.go:113-141) ----
        if (lane < K * 8) hintbits[lane] = 0;
        stage_unaligned(hst, sg + P::CT + L * G::ZSZ, HB, lane);
        handoff();
        {
            const uint8_t *hb = reinterpret_cast<const uint8_t *>(hst);
            uint32_t sop[K];
#pragma unroll
            for (int i = 0; i < K; i++) sop[i] = hb[P::OMEGA + i];
#pragma unroll
            for (int i = 0; i < K; i++) bad |= sop[i] > (uint32_t)P::OMEGA || (i > 0 && sop[i] < sop[i - 1]);
            for (int j0 = 0; j0 < P::OMEGA; j0 += 64) {
                const int jj = j0 + lane;
                if (jj < P::OMEGA) {
                    int poly = 0;
                    uint32_t start = 0;
#pragma unroll
                    for (int i = 0; i < K; i++)
                        if (sop[i] <= (uint32_t)jj) { poly = i + 1; start = sop[i]; }
                    const uint32_t v = hb[jj];
                    if (poly < K) {
                        if ((uint32_t)jj > start && v <= hb[jj - 1]) bad = true;
                        atomicOr(&hintbits[poly * 8 + (v >> 5)], 1u << (v & 31));
                    } else if (v != 0) {
                        bad = true;
                    }
                }
            }
        }
    } else if (wave < L) {
        // ---- z_wave: (gamma1_bits + 1)-bit fields, value gamma1 - field (pack.go:146-199); ||z||inf < gamma1 - beta ----
        uint32_t *st = zst[wave];
        stage_unaligned(st, sg + P::CT + wave * G::ZSZ, G::ZSZ, lane);
        if (lane == 0) { st[G::ZSZ / 4] = 0; st[G::ZSZ / 4 + 1] = 0; }  // slack dwords for lds_bits
        handoff();
        uint32_t cz[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t f = lds_bits<G::ZBITS>(st, 0, kyber::idx_l1(lane, r));
            uint32_t x = G::GAMMA1 - f;
            x += (uint32_t)((int32_t)x >> 31) & Q;
            bad |= dilithium::exceeds(x, G::GAMMA1 - G::BETA);
            cz[r] = x;
        }
        dilithium::ntt<true>(cz, z, xch, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) zhat_lds[wave][64 * r + lane] = cz[r];  // plain z-hat, < 17q
    }
    if (!RESIDENT && wave == K + 1) {  // ExpandA of the item's key (mat.go:15-49), a stream per lane
        expand_a_scratch<MODE, false, 1>(fifo_lds, my_rows, pk_table + kq * pk_stride, 0, 0, 1, lane);
    } else if (wave <= K) {
        const bool any_bad = __any(bad);
        if (lane == 0) bad_w[wave] = any_bad ? 1u : 0u;
    }
    // z-hat, c-hat, the hint bitmap and mu are in LDS -- and, for unparsed keys, the rows have left the CU and no stale L1 line of
    // the slice survives (rows_acquire: release, barrier, agent-scope acquire)
    if constexpr (RESIDENT) __syncthreads();
    else rows_acquire();
    if (wave < K) {
        const int i = wave;
        uint32_t zhat[L][4], chat[4];
#pragma unroll
        for (int jj = 0; jj < L; jj++)
#pragma unroll
            for (int r = 0; r < 4; r++) zhat[jj][r] = zhat_lds[jj][64 * r + lane];
#pragma unroll
        for (int r = 0; r < 4; r++) chat[r] = chat_lds[64 * r + lane];
        const uint32_t *irows = RESIDENT ? key_rows + kq * (size_t)(G::STREAMS * kPackedRowDwords) : my_rows;
        uint32_t acc[4] = {0, 0, 0, 0};
        mac_rows<L>(acc, irows, i * L, zhat, lane);  // 2^-32 A z-hat, < 2q
        uint32_t t[4], w[4];
        {
            const uint32_t *tp = reinterpret_cast<const uint32_t *>(pk_table + kq * (RESIDENT ? (size_t)G::PK : pk_stride) + 32 + 320 * i);  // t1, 10-bit fields (pack.go:52-66)
            const int b = 5 * lane, d = b >> 2;
            const uint64_t v = (((uint64_t)tp[d + 1] << 32) | tp[d]) >> (8 * (b & 3));
#pragma unroll
            for (int r = 0; r < 4; r++) t[r] = ((uint32_t)(v >> (10 * r)) & 0x3ffu) << dilithium::D;
            dilithium::relayout<4, 1, true>(t, xch, lane);
        }
        dilithium::ntt<true>(t, z, xch, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t ct1 = dilithium::mont32(t[r], chat[r]);  // 2^-32 c-hat * t1-hat, < 2q
            w[r] = dilithium::fold(acc[r] + 2 * Q - ct1);
        }
        dilithium::invntt<dilithium::INV256_RR, true>(w, z, xch, lane);
        unsigned w1v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int nidx = kyber::idx_l1(lane, r);
            const uint32_t hbit = (hintbits[i * 8 + (nidx >> 5)] >> (nidx & 31)) & 1;
            w1v[r] = dilithium::use_hint<P::GAMMA2>(dilithium::csubq(w[r]), hbit);
        }
        mlkem::stage_bits_l1<G::W1BITS, true>(xch, w1v, lane);
        mlkem::store_staged<G::W1BITS>(reinterpret_cast<uint32_t *>(muw1 + 8) + (G::W1SZ / 4) * i, xch, lane, false);
    }
    __syncthreads();  // w1 is behind mu
    if (wave == K) {
        const int half = lane >> 5, j = lane & 31;
        const CoopLane c = coop_lane(coopw, lane);
        uint32_t vlo, vhi;
        mlkem::coop_sponge17<true>(vlo, vhi, [&](int k) { return muw1[k]; }, G::MUW1 / 8, kDsShake, c, j);
        bool mine = true;
        if (j < P::CT / 8) {
            uint64_t v = 0;
            for (int b = 0; b < 8; b++) v |= (uint64_t)sg[8 * j + b] << (8 * b);
            mine = v == (((uint64_t)vhi << 32) | vlo);
        }
        const unsigned long long diff = __ballot(!mine) & 0xffffffffull;  // half 0 carries the sponge (half 1 mirrors it)
        uint32_t failed = 0;
#pragma unroll
        for (int k = 0; k <= K; k++) failed |= bad_w[k];
        if (half == 0 && j == 0) ok[item] = (diff == 0 && failed == 0) ? 1 : 0;
    }
    if constexpr (RESIDENT) tail_signal(tail);  // (a coalesced batch's completion flag, raised by the launch's last workgroup: keccak_dev.h TailFlag)
}

// ---- kernel F -----------------------------------------------------------------------------------

// The same with an item per lane pair (keccak_f1600_split), for the batches between the cooperative form's range and the
// sizes where a lane per item fills the chip.
template <int MODE>
__global__ void __launch_bounds__(64) mldsa_final_split_kernel(const uint8_t *__restrict__ sig, const uint8_t *__restrict__ muw1_ws,
                                                              const uint8_t *__restrict__ fail_ws, uint8_t *__restrict__ ok, size_t n) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    const int lane = threadIdx.x, parity = lane & 1;
    size_t idx = (size_t)blockIdx.x * 32 + (lane >> 1);
    const bool live = idx < n;
    if (!live) idx = n - 1;
    SplitState s;
    mlkem::split_sponge17<G::MUW1 / 8>(s, reinterpret_cast<const uint32_t *>(muw1_ws + idx * G::MUW1) + parity, kDsShake, parity != 0);
    const uint8_t *sg = sig + idx * G::SIG + 4 * parity;  // (signatures are byte-aligned rows)
    uint32_t diff = 0;
#pragma unroll
    for (int w = 0; w < P::CT / 8; w++) {
        uint32_t v = 0;
        for (int b = 0; b < 4; b++) v |= (uint32_t)sg[8 * w + b] << (8 * b);
        diff |= v ^ s.w[w];
    }
    diff |= split_partner(diff);
    if (live && parity == 0) ok[idx] = (diff == 0 && fail_ws[idx] == 0) ? 1 : 0;
}
// tr = SHAKE256(pk)[:TR] of every item on a lane pair, 64 bytes at tr_out + item * stride: in front of mldsa_prep_kernel for
// the same batches (15 of its 18 dependent permutations are tr)
template <int MODE>
__global__ void __launch_bounds__(64) mldsa_tr_split_kernel(const uint8_t *__restrict__ pk, uint8_t *__restrict__ tr_out, size_t stride, size_t n) {
    using G = DG<MODE>;
    const int lane = threadIdx.x, parity = lane & 1;
    size_t idx = (size_t)blockIdx.x * 32 + (lane >> 1);
    const bool live = idx < n;
    if (!live) idx = n - 1;
    SplitState s;
    mlkem::split_sponge17<G::PK / 8>(s, reinterpret_cast<const uint32_t *>(pk + idx * G::PK) + parity, kDsShake, parity != 0);
    if (live) {
        uint32_t *tr = reinterpret_cast<uint32_t *>(tr_out + idx * stride) + parity;
#pragma unroll
        for (int i = 0; i < 8; i++) tr[2 * i] = s.w[i];
    }
}
template <int MODE>
__global__ void __launch_bounds__(256) mldsa_final_kernel(const uint8_t *__restrict__ sig, const uint8_t *__restrict__ muw1_ws,
                                                          const uint8_t *__restrict__ fail_ws, uint8_t *__restrict__ ok, size_t n) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    KeccakState s;
    sponge17_words<G::MUW1 / 8>(s, reinterpret_cast<const uint64_t *>(muw1_ws + idx * G::MUW1), kDsShake);
    const uint8_t *sg = sig + idx * G::SIG;
    bool same = true;
    detail::static_for<0, P::CT / 8>([&](auto ic) {
        constexpr int w = decltype(ic)::v;
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v |= (uint64_t)sg[8 * w + b] << (8 * b);
        same &= v == (((uint64_t)s.hi[w] << 32) | s.lo[w]);
    });
    ok[idx] = (same && fail_ws[idx] == 0) ? 1 : 0;
}

// ---- key generation (sign/mldsa/mldsa65/internal/dilithium.go:181-267 NewKeyFromSeed) ------------

template <int MODE> struct KG {
    using G = DG<MODE>;
    using P = DP<MODE>;
    static constexpr int ETABITS = P::ETA == 2 ? 3 : 4;              // params.go DoubleEtaBits
    static constexpr int ETASZ = 32 * ETABITS;
    static constexpr int SKHDR = 32 + 32 + P::TR;                     // rho || key || tr
    static constexpr int SK = SKHDR + ETASZ * (G::L + G::K) + 416 * G::K;
    static constexpr int NS = G::L + G::K;                            // secret polynomials per item
    static constexpr int S_STRIDE = 264;                              // int8 row + spill slot
    static constexpr int LDS_S = G::IT * NS * S_STRIDE;
    static constexpr int LDS_SEC = LDS_S > G::LDS_FIFO ? LDS_S : G::LDS_FIFO;  // the secrets alias the FIFOs of phase A
    static constexpr int LDS_TOTAL = LDS_SEC + G::LDS_XCH;
};

// lane = item: (rho, rho', key) = SHAKE256(seed || K || L)[:128]  (dilithium.go:195-206)
template <int MODE>
__global__ void __launch_bounds__(256) mldsa_keygen_seed_kernel(const uint8_t *__restrict__ seed32, uint8_t *__restrict__ es_ws, size_t n) {
    using P = DP<MODE>;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    KeccakState s;
    keccak_zero(s);
    xor_words<0, 4>(s, reinterpret_cast<const uint64_t *>(seed32 + idx * 32));
    s.lo[4] = P::NIST ? ((uint32_t)P::K | ((uint32_t)P::L << 8) | (kDsShake << 16)) : kDsShake;  // dilithium.go:191-193
    s.hi[16] = 0x80000000u;
    keccak_f1600(s);
    store_words<0, 16>(reinterpret_cast<uint64_t *>(es_ws + idx * 128), s);
}

// Persistent single-wavefront workgroups, IT items per group.  phase A: ExpandA into the global scratch
// rows (same code as verification).  phase 1a: lane = (item, secret polynomial) samples s1, s2
// (sample.go:125-175, SHAKE256(rho' || LE16(nonce)), nibble rejection) into LDS as small integers.
// Then per item, in registers: s1-hat = NTT(s1) per polynomial, eta-packing of s1, s2 into sk;
// t-hat[i] = sum_j A[i][j] o s1-hat[j]; t = InvNTT(t-hat) + s2, Power2Round (field.go:35-52), t1 -> pk, t0 -> sk.
// FROM_SK: PrivateKey.Public() (dilithium.go:473-484 -> computeT0andT1 :149-179): `es_ws` is then the array of packed PRIVATE keys
// (rho at the head of each row, stride Kg::SK), the secrets are decoded from them instead of sampled, and only pk = rho || t1 is
// written (sk is not touched).
template <int D> __device__ __forceinline__ uint32_t gbits(const uint32_t *p, int n, int ndwords);
template <int MODE, bool FROM_SK = false>
__global__ void __launch_bounds__(64, CIRCL_DSA_WAVES_PER_EU)
    mldsa_keygen_kernel(const uint8_t *__restrict__ es_ws, uint8_t *__restrict__ pk, uint8_t *__restrict__ sk,
                        uint8_t *__restrict__ scratch, unsigned *__restrict__ work, size_t n) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using Kg = KG<MODE>;
    constexpr int K = P::K, L = P::L, NS = Kg::NS;
    constexpr size_t ES = FROM_SK ? (size_t)Kg::SK : size_t(128);  // stride of the rows that start with rho
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int8_t *sec = reinterpret_cast<int8_t *>(smem);  // aliases the FIFOs of phase A
    uint32_t *xch = reinterpret_cast<uint32_t *>(smem + Kg::LDS_SEC);
    const int lane = threadIdx.x;
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);
    uint32_t *rows = reinterpret_cast<uint32_t *>(scratch + (size_t)blockIdx.x * G::SCRATCH_BYTES);
    const size_t ngroups = (n + G::IT - 1) / G::IT;

#pragma unroll 1
  for (size_t grp = mlkem::next_group(work, lane, true, ngroups); grp < ngroups; grp = mlkem::next_group(work, lane, false, ngroups)) {
    const size_t item0 = grp * G::IT;
    // ---- phase A: the matrix ----
    __syncthreads();
    expand_a_scratch<MODE>(smem, rows, es_ws, ES, item0, n, lane);
    rows_acquire();

    // ---- phase 1a: sample the secrets, one stream per lane ----
    if constexpr (FROM_SK) {  // ... or decode them from the packed private key (pack.go:9-37: field = eta - coefficient)
#pragma unroll 1
        for (int gk = 0; gk < G::IT * NS; gk++) {
            size_t item = item0 + gk / NS;
            if (item >= n) item = n - 1;
            const uint32_t *p = reinterpret_cast<const uint32_t *>(es_ws + item * ES + Kg::SKHDR + Kg::ETASZ * (gk % NS));
            int8_t *row = sec + gk * Kg::S_STRIDE;
#pragma unroll
            for (int r = 0; r < 4; r++) row[lane + 64 * r] = (int8_t)(P::ETA - (int)gbits<Kg::ETABITS>(p, lane + 64 * r, Kg::ETASZ / 4));
        }
    } else {
        const bool on = lane < G::IT * NS;
        const int g = on ? lane / NS : 0, nonce = on ? lane % NS : 0;
        size_t item = item0 + g;
        if (item >= n) item = n - 1;
        KeccakState s;
        keccak_zero(s);
        xor_words<0, 8>(s, reinterpret_cast<const uint64_t *>(es_ws + item * 128 + 32));
        s.lo[8] = (uint32_t)nonce | (kDsShake << 16);
        s.hi[16] = 0x80000000u;
        int8_t *row = sec + (on ? lane : 0) * Kg::S_STRIDE;
        int cnt = on ? 0 : 256;
#pragma unroll 1
        while (__any(cnt < 256)) {
            keccak_f1600(s);
            if (on) {
                detail::static_for<0, 34>([&](auto ic) {
                    constexpr int w = decltype(ic)::v;  // 32-bit word w of the 136-byte block
                    const uint32_t word = (w & 1) ? s.hi[w >> 1] : s.lo[w >> 1];
#pragma unroll
                    for (int nb = 0; nb < 8; nb++) {  // low nibble of each byte first (t1 then t2)
                        uint32_t t = (word >> (4 * nb)) & 15u;
                        bool ok;
                        if constexpr (P::ETA == 2) {
                            ok = t <= 14;
                            t -= ((205 * t) >> 10) * 5;
                        } else {
                            ok = t <= 8;
                        }
                        row[cnt] = (int8_t)(P::ETA - (int)t);
                        cnt = min(cnt + (ok ? 1 : 0), 256);
                    }
                });
            }
        }
    }
    __syncthreads();
    // (issue priority for the ring phases, as in mldsa_verify_kernel, was measured here and not kept: ML-DSA-44 / 65 level, ML-DSA-87 -1.8 %,
    // profiles/r06_keygen_prio_ab.txt)

#pragma unroll 1
    for (int g = 0; g < G::IT; g++) {
        const size_t item = item0 + g;
        if (item >= n) break;
        uint8_t *pkp = pk + item * G::PK, *skp = FROM_SK ? nullptr : sk + item * Kg::SK;
        // ---- phase 1b: NTT(s1), pack s1 and s2 ----
        uint32_t shat[L][4];
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const int8_t *row = sec + (g * NS + k) * Kg::S_STRIDE;
            unsigned fld[4];
            uint32_t c[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int v = row[kyber::idx_l1(lane, r)];   // in [-eta, eta]
                fld[r] = (unsigned)(P::ETA - v);              // pack.go:9-37: field = q + eta - coefficient
                c[r] = v < 0 ? Q + v : (uint32_t)v;
            }
            if constexpr (!FROM_SK) {
                mlkem::stage_bits_l1<Kg::ETABITS>(xch, fld, lane);
                mlkem::store_staged<Kg::ETABITS>(reinterpret_cast<uint32_t *>(skp + Kg::SKHDR + Kg::ETASZ * k), xch, lane, false);
            }
            if (k < L) {
                dilithium::ntt(c, z, xch, lane);
#pragma unroll
                for (int r = 0; r < 4; r++) shat[k < L ? k : 0][r] = c[r];  // plain s1-hat
            }
        }

        // ---- phases 2 and 3: t = InvNTT(A s1-hat) + s2, Power2Round, pack ----
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            uint32_t w[4] = {0, 0, 0, 0};
            mac_rows<L>(w, rows, g * G::STREAMS + i * L, shat, lane);  // 2^-32 A s1-hat, < 2q
            dilithium::invntt<dilithium::INV256_RR>(w, z, xch, lane);
            const int8_t *s2 = sec + (g * NS + L + i) * Kg::S_STRIDE;
            unsigned t1[4], t0[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int v = s2[kyber::idx_l1(lane, r)];
                const uint32_t a = dilithium::csubq(dilithium::fold(w[r] + (v < 0 ? Q + v : (uint32_t)v)));
                uint32_t a0q, a1;
                dilithium::power2round(a, a0q, a1);  // field.go:35-52
                t1[r] = a1;
                t0[r] = ((1u << (dilithium::D - 1)) - (a0q - Q)) & ((1u << dilithium::D) - 1);  // pack.go:23-50 PackT0 field
            }
            mlkem::stage_bits_l1<10>(xch, t1, lane);
            mlkem::store_staged<10>(reinterpret_cast<uint32_t *>(pkp + 32 + 320 * i), xch, lane, false);
            if constexpr (!FROM_SK) {
                mlkem::stage_bits_l1<13>(xch, t0, lane);
                mlkem::store_staged<13>(reinterpret_cast<uint32_t *>(skp + Kg::SKHDR + Kg::ETASZ * NS + 416 * i), xch, lane, false);
            }
        }
        if (lane < 8) {
            const uint32_t r = reinterpret_cast<const uint32_t *>(es_ws + item * ES)[lane];
            reinterpret_cast<uint32_t *>(pkp)[lane] = r;                                                     // rho
            if constexpr (!FROM_SK) {
                reinterpret_cast<uint32_t *>(skp)[lane] = r;
                reinterpret_cast<uint32_t *>(skp + 32)[lane] = reinterpret_cast<const uint32_t *>(es_ws + item * 128 + 96)[lane];  // key
            }
        }
    }
  }
}

// ---- a key pair in ONE launch (small batches) --------------------------------------------------------------------------------
// NewKeyFromSeed (dilithium.go:181-267) of one seed by a workgroup of K wavefronts, instead of fill + seed + keygen + tr as four
// launches with every stage on one wavefront:
//   0  (rho, rho', key) = H(seed || K || L): wavefront 0, one cooperative permutation
//   1  ExpandA (K L streams, a stream per lane, 5 SHAKE128 blocks -- the long pole) on wavefront 1  BESIDE  ExpandS (L + K streams,
//      nibble rejection) on wavefront 0
//   2  s1-hat = NTT(s1) and the eta-packing of s1, s2: a wavefront per secret polynomial
//   3  t_i = InvNTT(A_i s1-hat) + s2_i, Power2Round, t1 -> pk, t0 -> sk: a wavefront per ROW
//   4  tr = H(pk): wavefront 0 on the cooperative permutation (10 / 15 / 20 dependent permutations: what is left)
// with a workgroup barrier between the phases.  Same bytes as the four kernels (their per-polynomial code); the item's matrix rows go
// through its part of a scratch slice as in the verification chain kernel.
template <int MODE>
__global__ void __launch_bounds__(DP<MODE>::K * 64) mldsa_keygen_chain_kernel(const uint8_t *__restrict__ seed32, uint8_t *__restrict__ es_ws,
                                                                              uint8_t *__restrict__ pk, uint8_t *__restrict__ sk,
                                                                              uint8_t *__restrict__ scratch, size_t n) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using Kg = KG<MODE>;
    constexpr int K = P::K, L = P::L, NS = Kg::NS, TRW = P::TR / 8;
    static_assert(K >= 2 && NS <= 2 * K && NS <= 64, "two secret polynomials per wavefront at most; a sampling stream per lane");
    __shared__ __attribute__((aligned(16))) uint8_t fifo_lds[G::LDS_FIFO];
    __shared__ __attribute__((aligned(16))) int8_t sec[NS * Kg::S_STRIDE];
    __shared__ __attribute__((aligned(16))) uint32_t xch_all[K][dilithium::kXchWords];
    __shared__ __attribute__((aligned(16))) uint32_t shat_lds[L][256];
    __shared__ __attribute__((aligned(16))) uint64_t coop_ws[100];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t item = blockIdx.x;
    if (item >= n) return;  // (block-uniform)
    uint32_t *xch = xch_all[wave];
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);
    uint32_t *rows = reinterpret_cast<uint32_t *>(scratch + (item / G::IT) * (size_t)G::SCRATCH_BYTES) + (item % G::IT) * (size_t)(G::STREAMS * kPackedRowDwords);
    uint8_t *es = es_ws + item * 128, *pkp = pk + item * G::PK, *skp = sk + item * Kg::SK;
    // ---- 0: the seed ----
    if (wave == 0) {
        const int j = lane & 31;
        const CoopLane c = coop_lane(coop_ws, lane);
        const uint64_t w = j < 4 ? reinterpret_cast<const uint64_t *>(seed32 + item * 32)[j] : 0ull;
        uint32_t vlo = (uint32_t)w, vhi = (uint32_t)(w >> 32);
        if (j == 4) vlo ^= P::NIST ? ((uint32_t)P::K | ((uint32_t)P::L << 8) | (kDsShake << 16)) : kDsShake;  // dilithium.go:191-193
        if (j == 16) vhi ^= 0x80000000u;
        keccak_f1600_coop2<true>(vlo, vhi, c);
        if (lane < 16) reinterpret_cast<uint64_t *>(es)[lane] = ((uint64_t)vhi << 32) | vlo;
    }
    __threadfence_block();
    __syncthreads();
    // ---- 1: the matrix beside the secrets ----
    if (wave == 1) {
        expand_a_scratch<MODE, false, 1>(fifo_lds, rows, es, 0, 0, 1, lane);
    } else if (wave == 0) {  // sample.go:125-175: SHAKE256(rho' || LE16(nonce)), nibble rejection, a stream per lane
        const bool on = lane < NS;
        KeccakState s;
        keccak_zero(s);
        xor_words<0, 8>(s, reinterpret_cast<const uint64_t *>(es + 32));
        s.lo[8] = (uint32_t)(on ? lane : 0) | (kDsShake << 16);
        s.hi[16] = 0x80000000u;
        int8_t *row = sec + (on ? lane : 0) * Kg::S_STRIDE;
        int cnt = on ? 0 : 256;
#pragma unroll 1
        while (__any(cnt < 256)) {
            keccak_f1600(s);
            if (on) {
                detail::static_for<0, 34>([&](auto ic) {
                    constexpr int w = decltype(ic)::v;  // 32-bit word w of the 136-byte block
                    const uint32_t word = (w & 1) ? s.hi[w >> 1] : s.lo[w >> 1];
#pragma unroll
                    for (int nb = 0; nb < 8; nb++) {  // low nibble of each byte first (t1 then t2)
                        uint32_t t = (word >> (4 * nb)) & 15u;
                        bool ok;
                        if constexpr (P::ETA == 2) {
                            ok = t <= 14;
                            t -= ((205 * t) >> 10) * 5;
                        } else {
                            ok = t <= 8;
                        }
                        row[cnt] = (int8_t)(P::ETA - (int)t);
                        cnt = min(cnt + (ok ? 1 : 0), 256);
                    }
                });
            }
        }
    }
    mlkem::rows_acquire();  // the rows have left the CU and no stale L1 line of the slice survives; the secrets are in LDS
    // ---- 2: a wavefront per secret polynomial: eta-packing into sk, s1-hat ----
#pragma unroll 1
    for (int k = wave; k < NS; k += K) {
        const int8_t *row = sec + k * Kg::S_STRIDE;
        unsigned fld[4];
        uint32_t c[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int v = row[kyber::idx_l1(lane, r)];   // in [-eta, eta]
            fld[r] = (unsigned)(P::ETA - v);              // pack.go:9-37: field = q + eta - coefficient
            c[r] = v < 0 ? Q + v : (uint32_t)v;
        }
        mlkem::stage_bits_l1<Kg::ETABITS, true>(xch, fld, lane);
        mlkem::store_staged<Kg::ETABITS>(reinterpret_cast<uint32_t *>(skp + Kg::SKHDR + Kg::ETASZ * k), xch, lane, false);
        if (k < L) {
            dilithium::ntt<true>(c, z, xch, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) shat_lds[k][64 * r + lane] = c[r];  // plain s1-hat
        }
    }
    __syncthreads();
    // ---- 3: row `wave`: t = InvNTT(A s1-hat) + s2, Power2Round, pack ----
    {
        const int i = wave;
        uint32_t shat[L][4];
#pragma unroll
        for (int jj = 0; jj < L; jj++)
#pragma unroll
            for (int r = 0; r < 4; r++) shat[jj][r] = shat_lds[jj][64 * r + lane];
        uint32_t w[4] = {0, 0, 0, 0};
        mac_rows<L>(w, rows, i * L, shat, lane);  // 2^-32 A s1-hat, < 2q
        dilithium::invntt<dilithium::INV256_RR, true>(w, z, xch, lane);
        const int8_t *s2 = sec + (L + i) * Kg::S_STRIDE;
        unsigned t1[4], t0[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int v = s2[kyber::idx_l1(lane, r)];
            const uint32_t a = dilithium::csubq(dilithium::fold(w[r] + (v < 0 ? Q + v : (uint32_t)v)));
            uint32_t a0q, a1;
            dilithium::power2round(a, a0q, a1);  // field.go:35-52
            t1[r] = a1;
            t0[r] = ((1u << (dilithium::D - 1)) - (a0q - Q)) & ((1u << dilithium::D) - 1);  // pack.go:23-50 PackT0 field
        }
        mlkem::stage_bits_l1<10, true>(xch, t1, lane);
        mlkem::store_staged<10>(reinterpret_cast<uint32_t *>(pkp + 32 + 320 * i), xch, lane, false);
        mlkem::stage_bits_l1<13, true>(xch, t0, lane);
        mlkem::store_staged<13>(reinterpret_cast<uint32_t *>(skp + Kg::SKHDR + Kg::ETASZ * NS + 416 * i), xch, lane, false);
        if (wave == K - 1 && lane < 8) {
            const uint32_t r = reinterpret_cast<const uint32_t *>(es)[lane];
            reinterpret_cast<uint32_t *>(pkp)[lane] = r;                                                   // rho
            reinterpret_cast<uint32_t *>(skp)[lane] = r;
            reinterpret_cast<uint32_t *>(skp + 32)[lane] = reinterpret_cast<const uint32_t *>(es + 96)[lane];  // key
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- 4: tr = SHAKE256(pk)[:TR] -> sk (dilithium.go:257-262) ----
    if (wave == 0) {
        const int j = lane & 31;
        const CoopLane c = coop_lane(coop_ws, lane);
        const uint64_t *pkw = reinterpret_cast<const uint64_t *>(pkp);
        uint32_t vlo, vhi;
        mlkem::coop_sponge17<true>(vlo, vhi, [&](int k) { return pkw[k]; }, G::PK / 8, kDsShake, c, j);
        if (lane < TRW) reinterpret_cast<uint64_t *>(skp + 64)[lane] = ((uint64_t)vhi << 32) | vlo;
    }
}

// lane = item: tr = SHAKE256(pk)[:64] -> sk[64:128]  (dilithium.go:257-262)
template <int MODE>
__global__ void __launch_bounds__(256) mldsa_keygen_finish_kernel(const uint8_t *__restrict__ pk, uint8_t *__restrict__ sk, size_t n) {
    using G = DG<MODE>;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    KeccakState s;
    sponge17_words<G::PK / 8>(s, reinterpret_cast<const uint64_t *>(pk + idx * G::PK), kDsShake);
    store_words<0, DP<MODE>::TR / 8>(reinterpret_cast<uint64_t *>(sk + idx * KG<MODE>::SK + 64), s);
}
// The same for small and medium batches (tr is 10 / 15 / 20 dependent permutations, the longest stage of a key generation):
// form 1 = two keys per wavefront on the cooperative permutation, form 2 = a key per lane pair (keccak_f1600_split).
template <int MODE>
__global__ void __launch_bounds__(64) mldsa_keygen_finish_small_kernel(const uint8_t *__restrict__ pk, uint8_t *__restrict__ sk, size_t n, int form) {
    using G = DG<MODE>;
    constexpr int TRW = DP<MODE>::TR / 8;
    __shared__ uint64_t ws[100];
    const int lane = threadIdx.x;
    if (form == 1) {
        const int half = lane >> 5, j = lane & 31;
        size_t idx = 2 * (size_t)blockIdx.x + half;
        const bool live = idx < n;
        if (!live) idx = n - 1;
        const CoopLane c = coop_lane(ws, lane);
        const uint64_t *pkw = reinterpret_cast<const uint64_t *>(pk + idx * G::PK);
        uint32_t vlo, vhi;
        mlkem::coop_sponge17(vlo, vhi, [&](int k) { return pkw[k]; }, G::PK / 8, kDsShake, c, j);
        if (live && j < TRW) reinterpret_cast<uint64_t *>(sk + idx * KG<MODE>::SK + 64)[j] = ((uint64_t)vhi << 32) | vlo;
        return;
    }
    const int parity = lane & 1;
    size_t idx = (size_t)blockIdx.x * 32 + (lane >> 1);
    const bool live = idx < n;
    if (!live) idx = n - 1;
    SplitState h;
    mlkem::split_sponge17<G::PK / 8>(h, reinterpret_cast<const uint32_t *>(pk + idx * G::PK) + parity, kDsShake, parity != 0);
    if (live) {
        uint32_t *tr = reinterpret_cast<uint32_t *>(sk + idx * KG<MODE>::SK + 64) + parity;
#pragma unroll
        for (int i = 0; i < TRW; i++) tr[2 * i] = h.w[i];
    }
}

// ---- signing (sign/mldsa/mldsa65/internal/dilithium.go:340-470 SignTo) ------
//
// First version, organised for correctness: one wavefront owns one signature through all of its
// rejection iterations (expected 4-7), resident wavefronts pull items from a ticket counter.  The
// expanded matrix and the NTT-domain secrets of the current item live in the wave's slice of a global
// scratch (L2 resident) because they are re-read in every iteration.  The in-loop sponges (ExpandMask:
// L streams, c~ = H(mu || w1): one stream) use few lanes; batching them across items is the obvious
// next optimisation.

template <int MODE> struct SG {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using Kg = KG<MODE>;
    static constexpr int K = P::K, L = P::L;
    static constexpr int A_ROW = 260;                                   // dwords per matrix row (+ spill slot)
    static constexpr int SCRATCH_DW = K * L * A_ROW + (L + 2 * K) * 256; // A rows, then s1-hat, s2-hat, t0-hat
    static constexpr int SCRATCH_BYTES = ((SCRATCH_DW * 4 + 255) / 256) * 256;
    static constexpr int YROW = 696;                                    // 5 blocks of 136 B + slack, per ExpandMask stream
    static constexpr int LDS_Y = L * YROW;
    static constexpr int LDS_W0 = K * 1024;                             // w0 (later w0 - c s2), u32, standard order
    static constexpr int LDS_W1 = K * 256;                              // w1 values, one byte each
    static constexpr int LDS_MUW1 = G::MUW1 + 8;                        // mu || packed w1 (hashed as is)
    static constexpr int LDS_Z = L * G::ZSZ;                            // packed z
    static constexpr int LDS_H = 96;                                    // hint bytes (omega + K <= 84)
    static constexpr int LDS_XCH = 4 * dilithium::kXchWords;  // padded exchange buffer (dilithium_dev.h relayout)
    static constexpr int LDS_MISC = 256;                                // ball block (200 B)
    static constexpr int LDS_TOTAL = LDS_Y + LDS_W0 + LDS_W1 + LDS_MUW1 + LDS_Z + LDS_H + LDS_XCH + LDS_MISC;
    static constexpr int SPEC_STRIDE = (G::SIG + 15) & ~15;                // one parked signature of the speculative tail
};

// lane = item: mu = H(tr || M'), rho'' = H(key || rnd || mu)[:64]  (dilithium.go:355-368) -> workspace
template <int MODE>
__global__ void __launch_bounds__(256) mldsa_sign_prep_kernel(const uint8_t *__restrict__ sk, const uint8_t *__restrict__ msg_blob,
                                                              const uint64_t *__restrict__ msg_off, const uint8_t *__restrict__ ctx_blob,
                                                              const uint64_t *__restrict__ ctx_off, const uint8_t *__restrict__ rnd,
                                                              int internal, uint8_t *__restrict__ mr_ws, size_t n, int shared_key,
                                                              uint8_t *__restrict__ dead_ws, const LongCtl *__restrict__ long_ctl,
                                                              const KeyIdx key_idx, uint32_t *__restrict__ rl_attempts = nullptr,
                                                              uint32_t *__restrict__ rl_best = nullptr, uint32_t *__restrict__ rl_list0 = nullptr,
                                                              uint32_t *__restrict__ rl_ctl = nullptr, unsigned rl_k0 = 1) {
    using Kg = KG<MODE>;
    using P = DP<MODE>;
    __shared__ uint32_t stage_lds[256 * kStageStride];
    uint32_t *stage = stage_lds + threadIdx.x * kStageStride;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    if (rl_attempts) {
        // PREPARED keys (a key table brings A and the transformed secrets): nothing is left of sign_secrets_kernel but the set-up of the
        // round signer's lists and control words (mldsa_sign_batched.h), done here -- one launch less in front of the first round
        rl_attempts[idx] = 0;
        rl_best[idx] = 0xffffffffu;                                                       // kNoSuccess
        for (unsigned o = 0; o < rl_k0; o++) rl_list0[idx * rl_k0 + o] = (uint32_t)idx | (o << 26);  // item | off << kEntryShift
        if (idx == 0) { rl_ctl[0] = (uint32_t)(n * rl_k0); rl_ctl[1] = 0; rl_ctl[2] = rl_k0; rl_ctl[3] = 1; rl_ctl[8] = 0; }  // count[0..1], kk[0..1], chain_done
    }
    {
        // sign.ErrContextTooLong (mldsa65/dilithium.go:63-65) / sign.ErrContextNotSupported (round 3): the host-buffer entry
        // points refuse such a batch up front; device-resident callers get an all-zero signature for the item
        // (mldsa_sign_zero_dead_kernel) instead of one over a truncated length byte or a silently dropped context
        const size_t cl = (ctx_blob && !internal) ? (size_t)(ctx_off[idx + 1] - ctx_off[idx]) : 0;
        dead_ws[idx] = (cl > 255 || (!P::NIST && cl > 0)) ? 1 : 0;
    }
    if (!P::NIST) internal = 1;  // round 3: mu = CRH(tr || msg)
    const uint8_t *skp = sk + (key_idx ? (size_t)key_idx[idx] : shared_key ? size_t(0) : idx) * Kg::SK;  // key_idx: a table of private keys
    KeccakState h;
    keccak_zero(h);
    xor_words<0, P::TR / 8>(h, reinterpret_cast<const uint64_t *>(skp + 64));  // tr
    const uint8_t *mp = msg_blob + msg_off[idx];
    const size_t mlen = (size_t)(msg_off[idx + 1] - msg_off[idx]);
    const uint8_t *cp = ctx_blob ? ctx_blob + ctx_off[idx] : nullptr;
    const size_t clen = ctx_blob ? (size_t)(ctx_off[idx + 1] - ctx_off[idx]) : 0;
    if (long_premade(long_ctl, mprime_total(msg_off, ctx_blob, ctx_off, internal, idx), n)) {  // a long message or a small batch: mu is already there (mldsa_mu_long_kernel)
        keccak_zero(h);
        xor_words<0, 8>(h, reinterpret_cast<const uint64_t *>(mr_ws + idx * 128));
    } else {
        absorb_message_and_squeeze<P::TR / 8>(h, mp, mlen, cp, clen, internal, stage);
        store_words<0, 8>(reinterpret_cast<uint64_t *>(mr_ws + idx * 128), h);  // mu
    }
    // rho'' = H(key || rnd || mu): built in the same state (mu moves up from words 0..7), one sponge live at a time
    constexpr int MU0 = P::NIST ? 8 : 4;                                       // round 3: rho'' = CRH(key || mu), no rnd (dilithium.go:357-364)
#pragma unroll
    for (int i = 7; i >= 0; i--) { h.lo[MU0 + i] = h.lo[i]; h.hi[MU0 + i] = h.hi[i]; }
#pragma unroll
    for (int i = 0; i < MU0; i++) { h.lo[i] = 0; h.hi[i] = 0; }
#pragma unroll
    for (int i = MU0 + 8; i < 25; i++) { h.lo[i] = 0; h.hi[i] = 0; }
    xor_words<0, 4>(h, reinterpret_cast<const uint64_t *>(skp + 32));          // key
    if constexpr (P::NIST) xor_words<4, 4>(h, reinterpret_cast<const uint64_t *>(rnd + idx * 32));  // rnd (zero = deterministic)
    h.lo[MU0 + 8] ^= kDsShake;
    h.hi[16] ^= 0x80000000u;
    keccak_f1600(h);
    store_words<0, 8>(reinterpret_cast<uint64_t *>(mr_ws + idx * 128 + 64), h);  // rho''
}

// The signing front end of a SMALL batch in one launch: what the long-message scan, mldsa_mu_long_kernel and mldsa_sign_prep_kernel do in
// three (plus a fill) -- mu = H(tr || M') and rho'' = H(key || rnd || mu), two items per wavefront on the cooperative permutation (three
// dependent permutations of ~3.5 us for a short message instead of two launches and a lane-form permutation), the "dead" flags, and for
// prepared keys the set-up of the round signer's lists.  Same bytes as the three kernels (tests/test_gpu_round3.py forces both ways).
template <int MODE>
__global__ void __launch_bounds__(64) mldsa_sign_front_kernel(const uint8_t *__restrict__ sk, size_t sk_stride, const KeyIdx key_idx,
                                                             const uint8_t *__restrict__ msg_blob, const uint64_t *__restrict__ msg_off,
                                                             const uint8_t *__restrict__ ctx_blob, const uint64_t *__restrict__ ctx_off,
                                                             const uint8_t *__restrict__ rnd, int internal, uint8_t *__restrict__ mr_ws, size_t n,
                                                             uint8_t *__restrict__ dead_ws, uint32_t *__restrict__ rl_attempts, uint32_t *__restrict__ rl_best,
                                                             uint32_t *__restrict__ rl_list0, uint32_t *__restrict__ rl_ctl, unsigned rl_k0) {
    using P = DP<MODE>;
    constexpr int TRW = P::TR / 8, MU0 = P::NIST ? 8 : 4;  // round 3: rho'' = CRH(key || mu), no rnd (dilithium.go:357-364)
    __shared__ uint64_t ws[100];
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    const CoopLane c = coop_lane(ws, lane);
    const int eff_internal = P::NIST ? internal : 1;  // round 3: mu = CRH(tr || msg)
#pragma unroll 1
    for (size_t pair = blockIdx.x; 2 * pair < n; pair += gridDim.x) {  // block-uniform
        const size_t e = 2 * pair + (size_t)half;
        const bool live = e < n;
        const size_t idx = live ? e : n - 1;  // the odd one out is done twice, stored once
        const size_t q = key_idx ? (size_t)key_idx[idx] : idx;
        const uint8_t *skp = sk + q * sk_stride;
        const size_t mlen = (size_t)(msg_off[idx + 1] - msg_off[idx]);
        const uint8_t *cp = ctx_blob ? ctx_blob + ctx_off[idx] : nullptr;
        const size_t clen = ctx_blob ? (size_t)(ctx_off[idx + 1] - ctx_off[idx]) : 0;
        if (live && j == 0) {  // sign.ErrContextTooLong / ErrContextNotSupported: see mldsa_sign_prep_kernel
            const size_t cl = (ctx_blob && !internal) ? clen : 0;
            dead_ws[idx] = (cl > 255 || (!P::NIST && cl > 0)) ? 1 : 0;
        }
        // ---- mu = SHAKE256(tr || M')[:64] (the loop of mldsa_mu_long_kernel) ----
        uint32_t vlo = 0, vhi = 0, mu_lo = 0, mu_hi = 0;
        if (j < TRW) {
            const uint64_t w = reinterpret_cast<const uint64_t *>(skp + 64)[j];
            vlo = (uint32_t)w;
            vhi = (uint32_t)(w >> 32);
        }
        const MPrime mpr(msg_blob + msg_off[idx], mlen, cp, clen, eff_internal);
        size_t pos = 0;
        int w0 = TRW;
        bool done = false;
#pragma unroll 1
        for (;;) {
            if (!done && j >= w0 && j < 17) {
                uint32_t lo, hi;
                mpr.word(pos + 8 * (size_t)(j - w0), lo, hi);
                vlo ^= lo;
                vhi ^= hi;
            }
            const size_t span = 8 * (size_t)(17 - w0);
            const bool last = mpr.total < pos + span;
            if (!done && last && j == 16) vhi ^= 0x80000000u;
            keccak_f1600_coop2(vlo, vhi, c);
            if (!done && last) {
                mu_lo = vlo;  // (word j of mu in lane j of the half, j < 8)
                mu_hi = vhi;
                done = true;
            }
            if (!__any(!done)) break;
            pos += span;
            w0 = 0;
        }
        // ---- rho'' = SHAKE256(key || rnd || mu)[:64]: mu moves up to words MU0 .. MU0 + 7 ----
        const int src = (lane & 32) | ((j - MU0) & 31);
        const uint32_t m_lo = (uint32_t)__shfl((int)mu_lo, src), m_hi = (uint32_t)__shfl((int)mu_hi, src);
        vlo = vhi = 0;
        if (j < 4) {
            const uint64_t w = reinterpret_cast<const uint64_t *>(skp + 32)[j];  // key
            vlo = (uint32_t)w;
            vhi = (uint32_t)(w >> 32);
        } else if (P::NIST && j < 8) {
            const uint64_t w = reinterpret_cast<const uint64_t *>(rnd + idx * 32)[j - 4];  // rnd (zero = deterministic)
            vlo = (uint32_t)w;
            vhi = (uint32_t)(w >> 32);
        } else if (j >= MU0 && j < MU0 + 8) {
            vlo = m_lo;
            vhi = m_hi;
        }
        if (j == MU0 + 8) vlo ^= kDsShake;
        if (j == 16) vhi ^= 0x80000000u;
        keccak_f1600_coop2(vlo, vhi, c);
        if (live && j < 8) {
            reinterpret_cast<uint64_t *>(mr_ws + idx * 128)[j] = ((uint64_t)mu_hi << 32) | mu_lo;
            reinterpret_cast<uint64_t *>(mr_ws + idx * 128 + 64)[j] = ((uint64_t)vhi << 32) | vlo;
        }
        if (rl_attempts && live && j == 0) {  // prepared keys: the round signer's lists (what is left of sign_secrets_kernel)
            rl_attempts[idx] = 0;
            rl_best[idx] = 0xffffffffu;                                                       // kNoSuccess
            for (unsigned o = 0; o < rl_k0; o++) rl_list0[idx * rl_k0 + o] = (uint32_t)idx | (o << 26);  // item | off << kEntryShift
            if (idx == 0) { rl_ctl[0] = (uint32_t)(n * rl_k0); rl_ctl[1] = 0; rl_ctl[2] = rl_k0; rl_ctl[3] = 1; rl_ctl[8] = 0; }
        }
    }
}

// lane = item: items whose context the scheme refuses (see mldsa_sign_prep_kernel) leave with an all-zero signature
template <int MODE>
__global__ void __launch_bounds__(256) mldsa_sign_zero_dead_kernel(uint8_t *__restrict__ sig, const uint8_t *__restrict__ dead_ws, size_t n) {
    const size_t item = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (item >= n || !dead_ws[item]) return;
    uint8_t *dst = sig + item * DG<MODE>::SIG;
    for (int b = 0; b < DG<MODE>::SIG; b++) dst[b] = 0;
}

// D-bit field n of a little-endian bit stream of aligned dwords in global memory
template <int D> __device__ __forceinline__ uint32_t gbits(const uint32_t *p, int n, int ndwords) {
    const int bit = n * D, w = bit >> 5, sh = bit & 31;
    const uint32_t lo = p[w], hi = (w + 1 < ndwords) ? p[w + 1] : 0u;
    return (sh ? alignbit(hi, lo, (uint32_t)sh) : lo) & ((1u << D) - 1);
}

template <int MODE>
__global__ void __launch_bounds__(64) mldsa_sign_kernel(const uint8_t *__restrict__ sk, const uint8_t *__restrict__ mr_ws,
                                                       uint8_t *__restrict__ sig, uint8_t *__restrict__ scratch,
                                                       unsigned *__restrict__ work, const uint32_t *__restrict__ list,
                                                       const uint32_t *__restrict__ attempts, size_t n, unsigned spec_w,
                                                       uint32_t *__restrict__ best, uint8_t *__restrict__ spec_sig, int shared_key,
                                                       const uint32_t *__restrict__ count_ptr, const KeyIdx key_idx) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using Kg = KG<MODE>;
    using S = SG<MODE>;
    constexpr int K = P::K, L = P::L;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *ybuf = smem;
    uint32_t *w0 = reinterpret_cast<uint32_t *>(smem + S::LDS_Y);
    uint8_t *w1b = smem + S::LDS_Y + S::LDS_W0;
    uint8_t *muw1 = w1b + S::LDS_W1;
    uint8_t *zpk = muw1 + S::LDS_MUW1;
    uint8_t *hbytes = zpk + S::LDS_Z;
    uint32_t *xch = reinterpret_cast<uint32_t *>(hbytes + S::LDS_H);
    uint8_t *misc = reinterpret_cast<uint8_t *>(xch) + S::LDS_XCH;
    uint32_t *arows = reinterpret_cast<uint32_t *>(scratch + (size_t)blockIdx.x * S::SCRATCH_BYTES);
    uint32_t *sec = arows + K * L * S::A_ROW;  // s1-hat (L), s2-hat (K), t0-hat (K), 256 dwords each, standard order
    const int lane = threadIdx.x;
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);

    // `list` (optional) names the n items to sign and `attempts` how many rejection rounds each of them
    // has already been through (the tail of mldsa_sign_batched, n = *count_ptr); otherwise items are 0..n-1 from scratch.
    // spec_w > 1 (tail only): spec_w wavefronts share an item and try its attempts a0 + w, a0 + w + spec_w, ...
    // in parallel.  The signature of the reference is the one of the FIRST successful attempt, so every success
    // lowers best[t] (atomicMin), a wave gives up once its next attempt lies beyond best[t], successful waves park
    // their signature in slot (t, w) of spec_sig, and sign_tail_commit_kernel copies the winner's slot out.
    if (count_ptr) n = *count_ptr;  // the tail of the device-driven batched signer: the list's length lives on the device
    const size_t units = n * spec_w;
#pragma unroll 1
    for (size_t u = mlkem::next_group(work, lane, true, units); u < units; u = mlkem::next_group(work, lane, false, units)) {
        const size_t t = u / spec_w;
        const unsigned spec_class = (unsigned)(u % spec_w);
        const size_t item = list ? (list[t] & 0x03ffffffu) : t;  // (entries carry an attempt offset above bit 26; the tail's are 0)
        const uint8_t *skp = sk + (key_idx ? (size_t)key_idx[item] : shared_key ? size_t(0) : item) * Kg::SK;
        const uint32_t *sk32 = reinterpret_cast<const uint32_t *>(skp);
        __syncthreads();
        // ---- setup 1: ExpandA(rho) into the scratch, lane = (i, j) ----
        {
            const bool on = lane < K * L;
            const int i = on ? lane / L : 0, j = on ? lane % L : 0;
            KeccakState s;
            keccak_zero(s);
            xor_words<0, 4>(s, reinterpret_cast<const uint64_t *>(skp));
            s.lo[4] = (uint32_t)j | ((uint32_t)i << 8) | (kDsShake << 16);
            s.hi[20] = 0x80000000u;
            uint32_t *row = arows + (on ? lane : 0) * S::A_ROW;
            int cnt = on ? 0 : 256;
#pragma unroll 1
            for (int blk = 0; blk < 5 || __any(cnt < 256); blk++) {
                keccak_f1600(s);
                if (on) {
                    for_each_candidate23(s, [&](uint32_t a) {
                        row[cnt] = a;  // rejected values are overwritten; slot 256 is a spill slot
                        cnt = min(cnt + (a < Q ? 1 : 0), 256);
                    });
                }
            }
        }
        // ---- setup 2: NTT of s1, s2, t0 (dilithium.go:149-179 PrivateKey.Unpack) into the scratch ----
#pragma unroll 1
        for (int k = 0; k < L + 2 * K; k++) {
            uint32_t c[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nidx = kyber::idx_l1(lane, r);
                int v;
                if (k < L + K) v = P::ETA - (int)gbits<Kg::ETABITS>(sk32 + (Kg::SKHDR + Kg::ETASZ * k) / 4, nidx, Kg::ETASZ / 4);
                else v = (1 << (dilithium::D - 1)) - (int)gbits<13>(sk32 + (Kg::SKHDR + Kg::ETASZ * (L + K) + 416 * (k - L - K)) / 4, nidx, 104);
                c[r] = v < 0 ? Q + v : (uint32_t)v;
            }
            dilithium::ntt(c, z, xch, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) sec[k * 256 + 4 * lane + r] = dilithium::fold(c[r]);
        }
        // mu into the hashing buffer
        if (lane < 16) reinterpret_cast<uint32_t *>(muw1)[lane] = reinterpret_cast<const uint32_t *>(mr_ws + item * 128)[lane];
        // the scratch rows were written by this wave (stores reach L2) and the same addresses were read for
        // the previous item: drop this CU's possibly stale L1 lines before re-reading them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();

        unsigned attempt = (attempts ? attempts[item] : 0) + spec_class - spec_w;  // pre-decremented (wraps)
        bool accepted = false, lost = false;
        KeccakState cs;  // c~ sponge (uniform across lanes)
#pragma unroll 1
        while (!accepted) {
            attempt += spec_w;
            if (spec_w > 1 && attempt > __hip_atomic_load(&best[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {  // wave-uniform
                lost = true;
                break;
            }
            const unsigned nonce = attempt * L;
            // ---- y = ExpandMask(rho'', nonce) (sample.go:178-196): lane l < L squeezes L streams ----
            {
                const bool on = lane < L;
                KeccakState s;
                keccak_zero(s);
                xor_words<0, 8>(s, reinterpret_cast<const uint64_t *>(mr_ws + item * 128 + 64));
                s.lo[8] = ((nonce + (on ? lane : 0)) & 0xffff) | (kDsShake << 16);
                s.hi[16] = 0x80000000u;
                uint32_t *yrow = reinterpret_cast<uint32_t *>(ybuf + (on ? lane : 0) * S::YROW);
#pragma unroll 1
                for (int blk = 0; blk < 5; blk++) {
                    keccak_f1600(s);
                    if (on) {
                        detail::static_for<0, 17>([&](auto ic) {
                            constexpr int w = decltype(ic)::v;
                            yrow[34 * blk + 2 * w] = s.lo[w];
                            yrow[34 * blk + 2 * w + 1] = s.hi[w];
                        });
                    }
                }
            }
            __syncthreads();
            // ---- y-hat ----
            uint32_t yh[L][4];
#pragma unroll
            for (int l = 0; l < L; l++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    uint32_t x = G::GAMMA1 - lds_bits<G::ZBITS>(reinterpret_cast<const uint32_t *>(ybuf + l * S::YROW), 0, kyber::idx_l1(lane, r));
                    x += (uint32_t)((int32_t)x >> 31) & Q;
                    yh[l][r] = x;
                }
                dilithium::ntt(yh[l], z, xch, lane);  // plain y-hat, < 17q
            }
            // ---- w = InvNTT(A y-hat), Decompose, w1 packing (dilithium.go:385-398) ----
#pragma unroll 1
            for (int i = 0; i < K; i++) {
                uint64_t acc[4] = {0, 0, 0, 0};  // lazy 64-bit dot product, one reduction per coefficient (see mac_rows)
#pragma unroll
                for (int j = 0; j < L; j++) {
                    const uint4 a = *reinterpret_cast<const uint4 *>(arows + (i * L + j) * S::A_ROW + 4 * lane);
                    acc[0] += (uint64_t)a.x * yh[j][0];
                    acc[1] += (uint64_t)a.y * yh[j][1];
                    acc[2] += (uint64_t)a.z * yh[j][2];
                    acc[3] += (uint64_t)a.w * yh[j][3];
                }
                uint32_t w[4];
#pragma unroll
                for (int r = 0; r < 4; r++) w[r] = dilithium::mont64(acc[r]);
                dilithium::invntt<dilithium::INV256_RR>(w, z, xch, lane);
                unsigned w1v[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int nidx = kyber::idx_l1(lane, r);
                    uint32_t a0, a1;
                    dilithium::decompose<P::GAMMA2>(dilithium::csubq(w[r]), a0, a1);
                    w0[i * 256 + nidx] = a0;
                    w1b[i * 256 + nidx] = (uint8_t)a1;
                    w1v[r] = a1;
                }
                mlkem::stage_bits_l1<G::W1BITS>(xch, w1v, lane);
                for (int d = lane; d < 8 * G::W1BITS; d += 64) reinterpret_cast<uint32_t *>(muw1 + 64 + G::W1SZ * i)[d] = xch[d];
            }
            __syncthreads();
            // ---- c~ = H(mu || w1), c-hat (dilithium.go:400-407); every lane computes the same sponge ----
            sponge17_words<G::MUW1 / 8>(cs, reinterpret_cast<const uint64_t *>(muw1), kDsShake);
            uint32_t chat[4];
            {
                KeccakState bs;
                keccak_zero(bs);
#pragma unroll
                for (int i = 0; i < P::CT / 8; i++) { bs.lo[i] = cs.lo[i]; bs.hi[i] = cs.hi[i]; }
                bs.lo[P::CT / 8] ^= kDsShake;
                bs.hi[16] ^= 0x80000000u;
                keccak_f1600(bs);
                __syncthreads();
                if (lane == 0) store_words<0, 25>(reinterpret_cast<uint64_t *>(misc), bs);
                __syncthreads();
                // misc holds the whole sponge state; the packed-z area is free until z is formed below and
                // serves as the (rarely needed) second-block buffer
                sample_in_ball_hat<MODE>(chat, misc, zpk, xch, z, lane);
            }
            bool bad = false;
            // ---- w0 - c s2 (dilithium.go:409-418) ----
#pragma unroll 1
            for (int i = 0; i < K; i++) {
                const uint4 sv = *reinterpret_cast<const uint4 *>(sec + (L + i) * 256 + 4 * lane);
                uint32_t t[4] = {dilithium::fold(dilithium::mont32(sv.x, chat[0])), dilithium::fold(dilithium::mont32(sv.y, chat[1])),
                                 dilithium::fold(dilithium::mont32(sv.z, chat[2])), dilithium::fold(dilithium::mont32(sv.w, chat[3]))};
                dilithium::invntt(t, z, xch, lane);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int nidx = kyber::idx_l1(lane, r);
                    const uint32_t v = dilithium::normalize(w0[i * 256 + nidx] + (2 * Q - t[r]));
                    bad |= dilithium::exceeds(v, P::GAMMA2 - G::BETA);
                    w0[i * 256 + nidx] = v;
                }
                if (__any(bad)) break;  // one polynomial out of range decides the attempt
            }
            if (__any(bad)) continue;
            // ---- z = y + c s1 (dilithium.go:420-429), packed as it will appear in the signature ----
#pragma unroll 1
            for (int l = 0; l < L; l++) {
                const uint4 sv = *reinterpret_cast<const uint4 *>(sec + l * 256 + 4 * lane);
                uint32_t t[4] = {dilithium::fold(dilithium::mont32(sv.x, chat[0])), dilithium::fold(dilithium::mont32(sv.y, chat[1])),
                                 dilithium::fold(dilithium::mont32(sv.z, chat[2])), dilithium::fold(dilithium::mont32(sv.w, chat[3]))};
                dilithium::invntt(t, z, xch, lane);
                unsigned fld[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    uint32_t y = G::GAMMA1 - lds_bits<G::ZBITS>(reinterpret_cast<const uint32_t *>(ybuf + l * S::YROW), 0, kyber::idx_l1(lane, r));
                    y += (uint32_t)((int32_t)y >> 31) & Q;
                    const uint32_t zz = dilithium::normalize(t[r] + y);
                    bad |= dilithium::exceeds(zz, G::GAMMA1 - G::BETA);
                    uint32_t f = G::GAMMA1 - zz;                 // pack.go:202-254 PolyPackLeGamma1
                    f += (uint32_t)((int32_t)f >> 31) & Q;
                    fld[r] = f;
                }
                mlkem::stage_bits_l1<G::ZBITS>(xch, fld, lane);
                for (int d = lane; d < 8 * G::ZBITS; d += 64) reinterpret_cast<uint32_t *>(zpk + G::ZSZ * l)[d] = xch[d];
                if (__any(bad)) break;
            }
            if (__any(bad)) continue;
            // ---- c t0, hints (dilithium.go:431-450) ----
            unsigned pop = 0;
            __syncthreads();
            for (int i = lane; i < (int)(S::LDS_H / 4); i += 64) reinterpret_cast<uint32_t *>(hbytes)[i] = 0;
            __syncthreads();
#pragma unroll 1
            for (int i = 0; i < K; i++) {
                const uint4 sv = *reinterpret_cast<const uint4 *>(sec + (L + K + i) * 256 + 4 * lane);
                uint32_t t[4] = {dilithium::fold(dilithium::mont32(sv.x, chat[0])), dilithium::fold(dilithium::mont32(sv.y, chat[1])),
                                 dilithium::fold(dilithium::mont32(sv.z, chat[2])), dilithium::fold(dilithium::mont32(sv.w, chat[3]))};
                dilithium::invntt(t, z, xch, lane);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int nidx = kyber::idx_l1(lane, r);
                    const uint32_t ct0 = dilithium::csubq(t[r]);
                    bad |= dilithium::exceeds(ct0, P::GAMMA2);
                    const uint32_t v = dilithium::csubq(w0[i * 256 + nidx] + ct0);
                    const uint32_t r1 = w1b[i * 256 + nidx];
                    // rounding.go:55-62 makeHint
                    const bool hbit = dilithium::make_hint<P::GAMMA2>(v, r1);
                    const unsigned long long mask = __ballot(hbit);  // coefficients 64 r .. 64 r + 63, ascending
                    if (hbit) {
                        const unsigned slot = pop + (unsigned)__popcll(mask & ((1ull << lane) - 1));
                        if (slot < (unsigned)P::OMEGA) hbytes[slot] = (uint8_t)nidx;
                    }
                    pop += (unsigned)__popcll(mask);
                }
                if (lane == 0) hbytes[P::OMEGA + i] = (uint8_t)(pop < 255 ? pop : 255);
            }
            if (__any(bad) || pop > (unsigned)P::OMEGA) continue;
            accepted = true;
        }
        if (lost) continue;  // another wave of this item succeeded at an earlier attempt
        if (spec_w > 1 && lane == 0) atomicMin(&best[t], attempt);
        // ---- sig = c~ || z || hints (dilithium.go:84-88), byte-wise because rows are unaligned ----
        __syncthreads();
        uint8_t *sg = spec_w > 1 ? spec_sig + (t * spec_w + spec_class) * S::SPEC_STRIDE : sig + item * G::SIG;
        if (lane == 0) store_words<0, P::CT / 8>(reinterpret_cast<uint64_t *>(misc), cs);
        __syncthreads();
        for (int b = lane; b < P::CT; b += 64) sg[b] = misc[b];
        for (int b = lane; b < L * G::ZSZ; b += 64) sg[P::CT + b] = zpk[b];
        for (int b = lane; b < P::OMEGA + K; b += 64) sg[P::CT + L * G::ZSZ + b] = hbytes[b];
    }
}

}  // namespace mldsa
}  // namespace circl
