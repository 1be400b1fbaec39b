// mldsa_sign_batched.h -- phase-split batch ML-DSA signing (included by circl_hip.hip).
//
// Same arithmetic as mldsa_sign_kernel (mldsa_kernels.h), reorganised so that every sponge runs with
// full lanes: the rejection loop of sign/mldsa/mldsa65/internal/dilithium.go:371-455 is executed as
// rounds over the list of still-unsigned items.  Per-item state lives in the workspace:
//   A rows (K L KB), s1-hat / s2-hat / t0-hat ((L+2K) KB), y bytes, w0, w1, mu || w1, c~ + ball sponge.
// One round = five launches over the active list:
//   mask   lane = (entry, l)            ExpandMask streams, 64 useful lanes per wave
//   w      wave = entry                 y-hat, w = InvNTT(A y-hat), Decompose, w1 packing
//   chal   lane = entry                 c~ = H(mu || w1), first SampleInBall block
//   finish wave = entry                 c s2 / z / c t0 / hints; a success lowers best[item]
//   compact                             next active list, attempt counters
// An ENTRY of the active list is (item, off): attempt number attempts[item] + off of that item.  Early rounds have one
// entry per item.  Once so few items are left that a round is latency-bound (five dependent launches whatever the
// count), the list carries k <= 8 consecutive attempts per item, tried in the same round: the signature is the one of
// the LOWEST successful attempt (atomicMin on best[item]; a commit launch copies it out), exactly the attempt the
// sequential loop of the reference would have stopped at, and the number of rounds shrinks.  The per-attempt buffers
// (y, w0, w1, mu || w1, c~) are indexed by the entry's position in the list, which never exceeds n.
// The host reads the entry count back (one round behind while the rounds are throughput-bound, every round once they
// are latency-bound) and stops when it is zero.
#pragma once
#include "mldsa_kernels.h"

namespace circl {
namespace mldsa {

template <int MODE> struct SB {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using Kg = KG<MODE>;
    static constexpr int K = P::K, L = P::L;
    static constexpr size_t A_BYTES = (size_t)K * L * kPackedRowDwords * 4;          // 24-bit packed rows (pack24)
    static constexpr size_t SEC_BYTES = (size_t)(L + 2 * K) * kPackedRowDwords * 4;
    static constexpr size_t Y_BYTES = (size_t)L * (G::ZSZ + 64);   // ZSZ payload + slack, per polynomial
    static constexpr int YROW_DW = (G::ZSZ + 64) / 4;
    static constexpr size_t W0_BYTES = (size_t)K * 1024;
    static constexpr size_t W1_BYTES = (size_t)K * 256;
    static constexpr size_t MUW1_BYTES = ((G::MUW1 + 63) / 64) * 64;
    static constexpr size_t CB_BYTES = 320;                          // c~ (<= 64 B), 56 B pad, ball state (200 B)
    static constexpr size_t PER_ITEM = A_BYTES + SEC_BYTES + Y_BYTES + W0_BYTES + W1_BYTES + MUW1_BYTES + CB_BYTES + 128 /* mu, rho'' */;
};

struct SignState {          // device pointers into the workspace, passed by value to the kernels
    uint8_t *mr;            // n x 128: mu, rho''
    uint32_t *A;            // n x K L x 256
    uint32_t *sec;          // n x (L + 2K) x 256
    uint32_t *y;            // n x L x YROW_DW
    uint32_t *w0;           // n x K x 256
    uint8_t *w1;            // n x K x 256
    uint8_t *muw1;          // n x MUW1_BYTES
    uint8_t *cb;            // n x 320
    uint32_t *attempts;     // n: attempts already spent on the item
    uint32_t *best;         // n: lowest successful `off` of the current round, kNoSuccess while unsigned
    uint32_t *list[2];      // active lists of entries: item | off << 28
    uint32_t *count;        // [0], [1]: list lengths
    uint32_t shared;        // 1: every item signs with the ONE private key at sk (A and the NTT-domain secrets exist once)
};

constexpr uint32_t kNoSuccess = 0xffffffffu;
constexpr int kEntryShift = 28;
constexpr uint32_t kEntryItemMask = (1u << kEntryShift) - 1;
constexpr unsigned kMaxSpec = 8;

// ---- setup ---------------------------------------------------------------------------------------

// lane = (item, i, j): ExpandA into the per-item A rows (mat.go:15-49), full lanes across items
template <int MODE>
__global__ void __launch_bounds__(256) sign_expand_a_kernel(const uint8_t *__restrict__ sk, SignState st, size_t n) {
    using Kg = KG<MODE>;
    constexpr int K = DP<MODE>::K, L = DP<MODE>::L;
    const size_t sidx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = sidx < n * K * L;
    const size_t item = on ? sidx / (K * L) : n - 1;
    const int p = on ? (int)(sidx % (K * L)) : 0;
    const int i = p / L, j = p % L;
    KeccakState s;
    keccak_zero(s);
    xor_words<0, 4>(s, reinterpret_cast<const uint64_t *>(sk + item * Kg::SK));
    s.lo[4] = (uint32_t)j | ((uint32_t)i << 8) | (kDsShake << 16);
    s.hi[20] = 0x80000000u;
    uint32_t *row = st.A + (item * K * L + p) * kPackedRowDwords;
    // accepted coefficients leave through a per-lane 16-slot LDS FIFO, four at a time (16-byte stores), with the
    // branch-free acceptance of the verify kernel's ExpandA (parse23_block_fifo)
    __shared__ __attribute__((aligned(16))) uint8_t fifo_lds[256 * DG<MODE>::FIFO_STRIDE];
    uint32_t *fifo = reinterpret_cast<uint32_t *>(fifo_lds + threadIdx.x * DG<MODE>::FIFO_STRIDE);
    int cnt = on ? 0 : 256, flushed = cnt;
#pragma unroll 1
    for (int blk = 0; blk < 5; blk++) {
        keccak_f1600(s);
        if (on) parse23_block_fifo<false, false, true>(s, fifo, row, cnt, flushed);
    }
#pragma unroll 1
    while (__any(flushed < 256)) {
        keccak_f1600(s);
        parse23_block_fifo<true, false, true>(s, fifo, row, cnt, flushed);
    }
}

// wave = item: NTT of s1, s2, t0 (dilithium.go:149-179) into the workspace; initial active list
template <int MODE>
__global__ void __launch_bounds__(64) sign_secrets_kernel(const uint8_t *__restrict__ sk, SignState st, size_t n) {
    using P = DP<MODE>;
    using Kg = KG<MODE>;
    constexpr int K = P::K, L = P::L;
    __shared__ __attribute__((aligned(16))) uint32_t xch[256];
    const int lane = threadIdx.x;
    const size_t item = blockIdx.x;
    const uint32_t *sk32 = reinterpret_cast<const uint32_t *>(sk + item * Kg::SK);
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);
    uint32_t *sec = st.sec + item * (L + 2 * K) * kPackedRowDwords;
#pragma unroll 1
    for (int k = 0; k < ((st.shared && item) ? 0 : L + 2 * K); k++) {  // shared key: workgroup 0 transforms the one key
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
        const uint32_t f[4] = {dilithium::fold(c[0]), dilithium::fold(c[1]), dilithium::fold(c[2]), dilithium::fold(c[3])};  // < 2^24
        store_poly24(sec + k * kPackedRowDwords, f, lane);
    }
    if (lane == 0) {
        st.attempts[item] = 0;
        st.best[item] = kNoSuccess;
        st.list[0][item] = (uint32_t)item;  // off = 0
    }
}

// ---- one round -------------------------------------------------------------------------------------

// lane = (active item, l): y = ExpandMask(rho'', L * attempts + l)  (sample.go:178-196)
template <int MODE>
__global__ void __launch_bounds__(256) sign_mask_kernel(SignState st, int cur) {
    using G = DG<MODE>;
    using B = SB<MODE>;
    constexpr int L = DP<MODE>::L;
    const unsigned cnt = st.count[cur];
    const size_t sidx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if ((size_t)(blockIdx.x * 256) >= (size_t)cnt * L) return;  // whole block idle
    const bool on = sidx < (size_t)cnt * L;
    const size_t slot = on ? sidx / L : 0;
    const uint32_t e = st.list[cur][slot];
    const size_t item = e & kEntryItemMask;
    const uint32_t off = e >> kEntryShift;
    const int l = on ? (int)(sidx % L) : 0;
    KeccakState s;
    keccak_zero(s);
    xor_words<0, 8>(s, reinterpret_cast<const uint64_t *>(st.mr + item * 128 + 64));
    s.lo[8] = (((st.attempts[item] + off) * L + l) & 0xffff) | (kDsShake << 16);
    s.hi[16] = 0x80000000u;
    uint32_t *yrow = st.y + (slot * L + l) * B::YROW_DW;
#pragma unroll 1
    for (int blk = 0; blk < 5; blk++) {
        keccak_f1600(s);
        if (on) {
            detail::static_for<0, 17>([&](auto ic) {
                constexpr int w = decltype(ic)::v;
                if (34 * blk + 2 * w < G::ZSZ / 4) {  // only the ZSZ payload bytes are kept
                    yrow[34 * blk + 2 * w] = s.lo[w];
                    yrow[34 * blk + 2 * w + 1] = s.hi[w];
                }
            });
        }
    }
}

// wave = active item: y-hat, w = InvNTT(A y-hat), Decompose, w1 (dilithium.go:376-398)
template <int MODE>
__global__ void __launch_bounds__(64) sign_w_kernel(SignState st, int cur) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using B = SB<MODE>;
    constexpr int K = P::K, L = P::L;
    __shared__ __attribute__((aligned(16))) uint32_t xch[256];
    if (blockIdx.x >= st.count[cur]) return;
    const int lane = threadIdx.x;
    const size_t slot = blockIdx.x;
    const size_t item = st.list[cur][slot] & kEntryItemMask;
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);
    if (lane < 16)  // mu in front of the w1 bytes that follow
        reinterpret_cast<uint32_t *>(st.muw1 + slot * B::MUW1_BYTES)[lane] = reinterpret_cast<const uint32_t *>(st.mr + item * 128)[lane];
    uint32_t yh[L][4];
#pragma unroll
    for (int l = 0; l < L; l++) {
        const uint32_t *yrow = st.y + (slot * L + l) * B::YROW_DW;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t x = G::GAMMA1 - gbits<G::ZBITS>(yrow, kyber::idx_l1(lane, r), G::ZSZ / 4);
            x += (uint32_t)((int32_t)x >> 31) & Q;
            yh[l][r] = x;
        }
        dilithium::ntt(yh[l], z, xch, lane);  // plain y-hat, < 17q
    }
    const uint32_t *arows = st.A + (st.shared ? 0 : item) * K * L * kPackedRowDwords;
#pragma unroll 1
    for (int i = 0; i < K; i++) {
        uint64_t acc[4] = {0, 0, 0, 0};  // lazy 64-bit dot product, one reduction per coefficient (see mac_rows)
#pragma unroll
        for (int j = 0; j < L; j++) {
            uint32_t a[4];
            load_poly24(a, arows + (i * L + j) * kPackedRowDwords, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) acc[r] += (uint64_t)a[r] * yh[j][r];
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
            st.w0[(slot * K + i) * 256 + nidx] = a0;
            st.w1[(slot * K + i) * 256 + nidx] = (uint8_t)a1;
            w1v[r] = a1;
        }
        mlkem::stage_bits_l1<G::W1BITS>(xch, w1v, lane);
        mlkem::store_staged<G::W1BITS>(reinterpret_cast<uint32_t *>(st.muw1 + slot * B::MUW1_BYTES + 64 + G::W1SZ * i), xch, lane, false);
    }
}

// lane = active item: c~ = SHAKE256(mu || w1)[:CT] and the first SampleInBall block (dilithium.go:400-405)
template <int MODE>
__global__ void __launch_bounds__(256) sign_challenge_kernel(SignState st, int cur) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using B = SB<MODE>;
    const size_t a = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= st.count[cur]) return;
    KeccakState s;
    sponge17_words<G::MUW1 / 8>(s, reinterpret_cast<const uint64_t *>(st.muw1 + a * B::MUW1_BYTES), kDsShake);
    uint64_t *cb = reinterpret_cast<uint64_t *>(st.cb + a * B::CB_BYTES);
    store_words<0, P::CT / 8>(cb, s);
    KeccakState bs;
    keccak_zero(bs);
#pragma unroll
    for (int i = 0; i < P::CT / 8; i++) { bs.lo[i] = s.lo[i]; bs.hi[i] = s.hi[i]; }
    bs.lo[P::CT / 8] ^= kDsShake;
    bs.hi[16] ^= 0x80000000u;
    keccak_f1600(bs);
    store_words<0, 25>(cb + 15, bs);  // ball state at byte 120
}

// wave = active item: the three rejection tests, hints, signature (dilithium.go:407-455, :84-88)
template <int MODE>
__global__ void __launch_bounds__(64) sign_finish_kernel(SignState st, int cur, uint8_t *__restrict__ sig, unsigned k) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using B = SB<MODE>;
    constexpr int K = P::K, L = P::L;
    __shared__ __attribute__((aligned(16))) uint32_t xch[256];
    __shared__ __attribute__((aligned(16))) uint8_t zpk[L * G::ZSZ];
    __shared__ __attribute__((aligned(16))) uint8_t hbytes[96];
    __shared__ __attribute__((aligned(16))) uint8_t blk[144];
    if (blockIdx.x >= st.count[cur]) return;
    static_assert((size_t)K * 1024 >= (size_t)G::SIG, "a slot's w0 area can park its signature");
    const int lane = threadIdx.x;
    const size_t slot = blockIdx.x;
    const uint32_t e = st.list[cur][slot];
    const size_t item = e & kEntryItemMask;
    const uint32_t off = e >> kEntryShift;
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);
    const uint8_t *cb = st.cb + slot * B::CB_BYTES;
    uint32_t chat[4];
    sample_in_ball_hat<MODE>(chat, cb + 120, blk, xch, z, lane);
    const uint32_t *sec = st.sec + (st.shared ? 0 : item) * (L + 2 * K) * kPackedRowDwords;
    uint32_t *w0 = st.w0 + slot * K * 256;
    auto mul_c = [&](uint32_t (&t)[4], const uint32_t *row) {
        uint32_t sv[4];
        load_poly24(sv, row, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) t[r] = dilithium::fold(dilithium::mont32(sv[r], chat[r]));
        dilithium::invntt(t, z, xch, lane);
    };
    bool bad = false;
    // w0 - c s2
#pragma unroll 1
    for (int i = 0; i < K; i++) {
        uint32_t t[4];
        mul_c(t, sec + (L + i) * kPackedRowDwords);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int nidx = kyber::idx_l1(lane, r);
            const uint32_t v = dilithium::normalize(w0[i * 256 + nidx] + (2 * Q - t[r]));
            bad |= dilithium::exceeds(v, P::GAMMA2 - G::BETA);
            w0[i * 256 + nidx] = v;
        }
        if (__any(bad)) break;  // one polynomial out of range decides the attempt: the remaining inverse transforms are moot
    }
    bool reject = __any(bad);
    // z = y + c s1
    if (!reject) {
#pragma unroll 1
        for (int l = 0; l < L; l++) {
            uint32_t t[4];
            mul_c(t, sec + l * kPackedRowDwords);
            const uint32_t *yrow = st.y + (slot * L + l) * B::YROW_DW;
            unsigned fld[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint32_t y = G::GAMMA1 - gbits<G::ZBITS>(yrow, kyber::idx_l1(lane, r), G::ZSZ / 4);
                y += (uint32_t)((int32_t)y >> 31) & Q;
                const uint32_t zz = dilithium::normalize(t[r] + y);
                bad |= dilithium::exceeds(zz, G::GAMMA1 - G::BETA);
                uint32_t f = G::GAMMA1 - zz;
                f += (uint32_t)((int32_t)f >> 31) & Q;
                fld[r] = f;
            }
            mlkem::stage_bits_l1<G::ZBITS>(xch, fld, lane);
            for (int d = lane; d < 8 * G::ZBITS; d += 64) reinterpret_cast<uint32_t *>(zpk + G::ZSZ * l)[d] = xch[d];
            if (__any(bad)) break;
        }
        reject = __any(bad);
    }
    // c t0, hints
    unsigned pop = 0;
    if (!reject) {
        __syncthreads();
        for (int i = lane; i < 24; i += 64) reinterpret_cast<uint32_t *>(hbytes)[i] = 0;
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            uint32_t t[4];
            mul_c(t, sec + (L + K + i) * kPackedRowDwords);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nidx = kyber::idx_l1(lane, r);
                const uint32_t ct0 = dilithium::csubq(t[r]);
                bad |= dilithium::exceeds(ct0, P::GAMMA2);
                const uint32_t v = dilithium::csubq(w0[i * 256 + nidx] + ct0);
                const uint32_t r1 = st.w1[(slot * K + i) * 256 + nidx];
                const bool hbit = !(v <= P::GAMMA2 || v > Q - P::GAMMA2 || (v == Q - P::GAMMA2 && r1 == 0));
                const unsigned long long mask = __ballot(hbit);
                if (hbit) {
                    const unsigned slot = pop + (unsigned)__popcll(mask & ((1ull << lane) - 1));
                    if (slot < (unsigned)P::OMEGA) hbytes[slot] = (uint8_t)nidx;
                }
                pop += (unsigned)__popcll(mask);
            }
            if (lane == 0) hbytes[P::OMEGA + i] = (uint8_t)(pop < 255 ? pop : 255);
        }
        reject = __any(bad) || pop > (unsigned)P::OMEGA;
    }
    if (reject) return;  // sign_compact_kernel charges the round's attempts to the item
    __syncthreads();     // every lane is done with w0
    // with one attempt per item the signature goes straight out; with several, this one is parked in the slot's w0
    // area and sign_commit_kernel copies the lowest successful attempt's
    uint8_t *sg = k == 1 ? sig + item * G::SIG : reinterpret_cast<uint8_t *>(w0);
    for (int b = lane; b < P::CT; b += 64) sg[b] = cb[b];
    for (int b = lane; b < L * G::ZSZ; b += 64) sg[P::CT + b] = zpk[b];
    for (int b = lane; b < P::OMEGA + K; b += 64) sg[P::CT + L * G::ZSZ + b] = hbytes[b];
    if (lane == 0) atomicMin(&st.best[item], off);
}

// wave = entry, rounds with several attempts per item only: the lowest successful attempt's parked signature -> sig
template <int MODE>
__global__ void __launch_bounds__(64) sign_commit_kernel(SignState st, int cur, uint8_t *__restrict__ sig) {
    using G = DG<MODE>;
    constexpr int K = DP<MODE>::K;
    const size_t slot = blockIdx.x;
    if (slot >= st.count[cur]) return;
    const uint32_t e = st.list[cur][slot];
    const size_t item = e & kEntryItemMask;
    if (st.best[item] != (e >> kEntryShift)) return;
    const uint32_t *src = st.w0 + slot * K * 256;
    uint8_t *dst = sig + item * G::SIG;   // SIG is not a multiple of 4 for every parameter set: dwords, then the tail bytes
    for (int d = threadIdx.x; d < G::SIG / 4; d += 64) {
        const uint32_t w = src[d];
        dst[4 * d] = (uint8_t)w; dst[4 * d + 1] = (uint8_t)(w >> 8); dst[4 * d + 2] = (uint8_t)(w >> 16); dst[4 * d + 3] = (uint8_t)(w >> 24);
    }
    for (int b = (G::SIG / 4) * 4 + threadIdx.x; b < G::SIG; b += 64) dst[b] = reinterpret_cast<const uint8_t *>(src)[b];
}

// next active list: every item of the current one (k entries each) that is still unsigned has spent k attempts and gets
// k_next entries
__global__ void __launch_bounds__(256) sign_compact_kernel(SignState st, int cur, unsigned k, unsigned k_next) {
    const size_t a = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= st.count[cur]) return;
    const uint32_t e = st.list[cur][a];
    if ((e >> kEntryShift) != 0) return;  // the item's first entry speaks for it
    const uint32_t item = e & kEntryItemMask;
    if (st.best[item] != kNoSuccess) return;
    st.attempts[item] += k;
    const uint32_t base = atomicAdd(&st.count[cur ^ 1], k_next);
    for (unsigned j = 0; j < k_next; j++) st.list[cur ^ 1][base + j] = item | (j << kEntryShift);
}

}  // namespace mldsa
}  // namespace circl
