// mldsa_sign_batched.h -- phase-split batch ML-DSA signing (included by circl_hip.hip).
//
// Same arithmetic as mldsa_sign_kernel (mldsa_kernels.h), reorganised so that every sponge runs with
// full lanes: the rejection loop of sign/mldsa/mldsa65/internal/dilithium.go:371-455 is executed as
// rounds over the list of still-unsigned items.  Per-item state lives in the workspace:
//   A rows (K L KB), s1-hat / s2-hat / t0-hat ((L+2K) KB), y bytes, w0, w1, mu || w1, c~ + ball sponge.
// One round = eight launches over the active list:
//   mask     lane = (entry, l)          ExpandMask streams, 64 useful lanes per wave
//   w        wave = 2 entries           y-hat, w = InvNTT(A y-hat), Decompose, w1 packing; two entries of one item share the matrix reads
//   chal 0/1 lane = entry               c~ = H(mu || w1), first SampleInBall block
//   finish 0/1 wave = entry             c s2 / z / c t0 / hints; a success lowers best[item]
//   commit   wave = entry               (speculative rounds only) the lowest successful attempt's signature -> sig
//   compact  lane = entry               survivors -> next list
// (chal / finish run as two passes: in a LAZY round -- pairs of attempts, sign_next_k -- the second attempt's challenge and norm
// tests run only for the items whose first attempt was rejected; in every other round pass 1 finds nothing to do.)
// An ENTRY of the active list is (item, off): attempt number attempts[item] + off of that item.  Early rounds have one
// entry per item.  Once so few items are left that a round is latency-bound (five dependent launches whatever the
// count), the list carries k <= 64 consecutive attempts per item, tried in the same round: the signature is the one of
// the LOWEST successful attempt (atomicMin on best[item]; the commit launch copies it out), exactly the attempt the
// sequential loop of the reference would have stopped at, and the number of rounds shrinks.  The per-attempt buffers
// (y, w0, w1, mu || w1, c~) are indexed by the entry's position in the list, which never exceeds the entry capacity.
//
// THE DEVICE DRIVES THE LOOP.  The list lengths and the attempts-per-item factor k live in device memory (SignState::count,
// ::kk, double-buffered by round parity); every kernel is a grid-stride / persistent loop bounded by the device-side
// count, and the commit kernel derives the next round's k from the count it sees.  The host therefore enqueues a FIXED
// schedule of rounds (sized so that the probability of an item surviving it is below 2^-40, sign_round_schedule) and
// never reads anything back: circl_hip_mldsa_sign_dev is asynchronous like every other _dev entry point.  Rounds that
// find their list empty cost five near-empty launches.  Whatever is still unsigned after the schedule (practically never)
// is finished by the persistent kernel of mldsa_kernels.h, which takes the final list and its device-side length.
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
    // low parts a0 + q < 2^24 (later r0 < q), four per lane in three dwords (pack24): an internal array, so no coefficient order --
    // 768 B per polynomial instead of 1 KB as uint32 (measured on one box, profiles/r04_sign_ab.txt: +2.1 % signatures/s)
    static constexpr size_t W0_BYTES = (size_t)K * kPackedRowDwords * 4;
    static constexpr int W0_SLOT_DW = K * kPackedRowDwords;
    static constexpr size_t MUW1_BYTES = ((G::MUW1 + 63) / 64) * 64;
    static constexpr size_t CB_BYTES = 320;                          // c~ (<= 64 B), 56 B pad, ball state (200 B)
    static constexpr size_t PER_ITEM = A_BYTES + SEC_BYTES + 128 /* mu, rho'' */;                   // per item (key material)
    static constexpr size_t PER_ENTRY = Y_BYTES + W0_BYTES + MUW1_BYTES + CB_BYTES;                 // per list entry (one attempt)
};

struct SignState {          // device pointers into the workspace, passed by value to the kernels
    uint8_t *mr;            // n x 128: mu, rho''
    uint32_t *A;            // n x K L x 256
    uint32_t *sec;          // n x (L + 2K) x 256
    uint32_t *y;            // n x L x YROW_DW
    uint32_t *w0;           // entries x K x kPackedRowDwords (pack24)
    uint8_t *muw1;          // n x MUW1_BYTES
    uint8_t *cb;            // n x 320
    uint32_t *attempts;     // n: attempts already spent on the item
    uint32_t *best;         // n: lowest successful `off` of the current round, kNoSuccess while unsigned
    uint32_t *list[2];      // active lists of entries: item | off << kEntryShift; round r reads list[r & 1]
    uint32_t *count;        // [0], [1]: list lengths (round r reads count[r & 1], its commit kernel fills count[(r + 1) & 1])
    uint32_t *kk;           // [0], [1]: entries per item of the list with the same parity
    uint32_t shared;        // 1: every item signs with the ONE private key at sk (A and the NTT-domain secrets exist once); 2: ... and they are ready-made (key table)
    uint32_t spec_target;   // rounds speculate (k > 1) once at most this many entries would result
    uint32_t capacity;      // entries the per-attempt buffers and the lists can hold
    uint32_t pair;          // 1: rounds too long to speculate widely still try TWO attempts per item (see sign_next_k)
    KeyIdx key_idx;          // shared == 2 with a table of SEVERAL prepared keys: item i signs with entry key_idx[i] (nullptr: entry 0)
    unsigned *tail_work;    // the persistent tail kernel's ticket counter (64 words), zeroed by the last round's compaction
    unsigned *chain_done;   // workgroups of a one-launch round that are through (sign_round_chain_kernel); zero between rounds
    // which entry of A / sec an item uses
    __device__ __forceinline__ size_t key_of(size_t item) const { return shared ? (key_idx ? (size_t)key_idx[item] : size_t(0)) : item; }
};

constexpr uint32_t kNoSuccess = 0xffffffffu;
constexpr int kEntryShift = 26;  // (mldsa_sign_prep_kernel, which sets the lists up for prepared keys, spells the 26 out)
constexpr uint32_t kEntryItemMask = (1u << kEntryShift) - 1;
constexpr unsigned kMaxSpec = 64;  // 6 bits of `off`
// Small and medium batches speculate widely: a round is four dependent Keccak-latency launches (~250 us) whatever its length, so a
// batch of 2^10 items with room for only 2 attempts per item and round took 1.8 ms in ~8 rounds; with 32 k entries it tries
// 32 attempts per item at once and is done in two rounds.  (11 KB of workspace per entry: 360 MB at most.)
constexpr size_t kMinEntryCapacity = 32768;

// attempts per item of the NEXT list when `items` items may survive into it (the same rule on the host, which sizes the
// schedule, and on the device, which applies it to the real counts).
//   items <= spec_target / 2   k = spec_target / items (<= 64): the rounds are latency-bound, every attempt of a round runs at once
//   otherwise                  k = 2 (`pair`) -- a LAZY PAIR: ExpandMask and w = A y-hat of attempts a and a + 1 are computed
//                              together, because the w kernel is bound by reading the item's 23 KB of matrix rows and one read
//                              then serves two attempts; challenge and the norm tests of attempt a + 1 run in a second pass and
//                              only for the items whose attempt a was rejected.  Expected work per signature (p = success
//                              probability of an attempt, 0.196 for ML-DSA-65): 2 / (1 - (1-p)^2) = 5.65 masks and w products
//                              instead of 1 / p = 5.1, 2.83 matrix reads instead of 5.1, the same 5.1 challenges / finishes,
//                              half the rounds.
__host__ __device__ inline unsigned sign_next_k(unsigned long long items, unsigned spec_target, unsigned pair) {
    if (items == 0) return 1u;
    if (2 * items > spec_target) return pair ? 2u : 1u;
    const unsigned long long k = spec_target / items;
    return (unsigned)(k < 1 ? 1 : k > kMaxSpec ? kMaxSpec : k);
}
// a round is LAZY when its list holds pairs that are too many to run all at once
__host__ __device__ inline bool sign_round_lazy(unsigned long long count, unsigned k, unsigned spec_target) { return k == 2 && count > spec_target; }

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
    // slot-major per wavefront (parse23_block_fifo): wave w owns 16 x 64 dwords
    uint32_t *fifo = reinterpret_cast<uint32_t *>(fifo_lds) + (threadIdx.x >> 6) * (16 * kFifoLanes) + (threadIdx.x & 63);
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

// wave = item: NTT of s1, s2, t0 (dilithium.go:149-179) into the workspace; initial active list with k0 entries per
// item (k0 > 1: a small batch speculates from its first round on) and the loop's control words
template <int MODE>
__global__ void __launch_bounds__(64) sign_secrets_kernel(const uint8_t *__restrict__ sk, SignState st, size_t n, unsigned k0) {
    using P = DP<MODE>;
    using Kg = KG<MODE>;
    constexpr int K = P::K, L = P::L;
    __shared__ __attribute__((aligned(16))) uint32_t xch[dilithium::kXchWords];
    const int lane = threadIdx.x;
    const size_t item = blockIdx.x;
    const uint32_t *sk32 = reinterpret_cast<const uint32_t *>(sk + item * Kg::SK);
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);
    uint32_t *sec = st.sec + item * (L + 2 * K) * kPackedRowDwords;
    // software-pipelined like the finish kernel: the packed words of polynomial k + 1 are requested before the transform of
    // polynomial k (a wave's 17 transforms are otherwise a chain of load -> unpack -> three LDS exchanges -> store)
    struct Raw { uint32_t lo[4], hi[4]; };
    auto fetch = [&](Raw &raw, int k) {
        const bool eta = k < L + K;
        const uint32_t *p = sk32 + (eta ? (Kg::SKHDR + Kg::ETASZ * k) / 4 : (Kg::SKHDR + Kg::ETASZ * (L + K) + 416 * (k - L - K)) / 4);
        const int bits = eta ? Kg::ETABITS : 13, ndw = eta ? Kg::ETASZ / 4 : 104;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int w = (kyber::idx_l1(lane, r) * bits) >> 5;
            raw.lo[r] = p[w];
            raw.hi[r] = (w + 1 < ndw) ? p[w + 1] : 0u;
        }
    };
    auto decode = [&](uint32_t (&c)[4], const Raw &raw, int k) {
        const bool eta = k < L + K;
        const int bits = eta ? Kg::ETABITS : 13;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int sh = (kyber::idx_l1(lane, r) * bits) & 31;
            const uint32_t fld = (sh ? alignbit(raw.hi[r], raw.lo[r], (uint32_t)sh) : raw.lo[r]) & ((1u << bits) - 1);
            const int v = eta ? P::ETA - (int)fld : (1 << (dilithium::D - 1)) - (int)fld;
            c[r] = v < 0 ? Q + v : (uint32_t)v;
        }
    };
    // shared key: workgroup 0 transforms the one key; shared == 2: A and the transformed secrets come ready-made from a key table
    // that lives across calls (keytable.h) -- nothing to transform, only the lists to set up
    const int npoly = (st.shared == 2 || (st.shared && item)) ? 0 : L + 2 * K;
    Raw raw;
    if (npoly) fetch(raw, 0);
#pragma unroll 1
    for (int k = 0; k < npoly; k++) {
        Raw next = raw;
        if (k + 1 < npoly) fetch(next, k + 1);
        uint32_t c[4];
        decode(c, raw, k);
        dilithium::ntt(c, z, xch, lane);
        const uint32_t f[4] = {dilithium::fold(c[0]), dilithium::fold(c[1]), dilithium::fold(c[2]), dilithium::fold(c[3])};  // < 2^24
        store_poly24(sec + k * kPackedRowDwords, f, lane);
        raw = next;
    }
    if (lane == 0) {
        st.attempts[item] = 0;
        st.best[item] = kNoSuccess;
    }
    if ((unsigned)lane < k0) st.list[0][item * k0 + lane] = (uint32_t)item | ((uint32_t)lane << kEntryShift);
    if (item == 0 && lane == 0) {
        st.count[0] = (uint32_t)(n * k0);
        st.count[1] = 0;
        st.kk[0] = k0;
        st.kk[1] = 1;
        *st.chain_done = 0;
    }
}

// ---- one round -------------------------------------------------------------------------------------
// Every kernel of a round takes the round's parity `cur` and bounds itself with the device-side list length
// st.count[cur]; grids are fixed-size (the host does not know the count), work is distributed by grid-stride loops.

// lane = (entry, l): y = ExpandMask(rho'', L * attempts + l)  (sample.go:178-196).  Also clears the next list's length.
template <int MODE, bool SPLIT = false>
__global__ void __launch_bounds__(256) sign_mask_kernel(SignState st, int cur) {
    using G = DG<MODE>;
    using B = SB<MODE>;
    constexpr int L = DP<MODE>::L;
    // SPLIT: a stream per lane PAIR (keccak_f1600_split) for the rounds whose streams would leave the SIMDs at or below one
    // wavefront each: 2/3 of the five-permutation chain.  The output is the raw stream, so each lane stores its own dwords.
    constexpr int PER = SPLIT ? 128 : 256;
    const size_t total = (size_t)st.count[cur] * L;
    if (blockIdx.x == 0 && threadIdx.x == 0) st.count[cur ^ 1] = 0;  // filled by this round's compact kernel, several launches later
    const int parity = threadIdx.x & 1;
#pragma unroll 1
    for (size_t base = (size_t)blockIdx.x * PER; base < total; base += (size_t)gridDim.x * PER) {  // block-uniform
        const size_t sidx = base + (SPLIT ? threadIdx.x >> 1 : threadIdx.x);
        const bool on = sidx < total;
        const size_t slot = on ? sidx / L : 0;
        const uint32_t e = st.list[cur][slot];
        const size_t item = e & kEntryItemMask;
        const uint32_t off = e >> kEntryShift;
        const int l = on ? (int)(sidx % L) : 0;
        const uint32_t nonce = (((st.attempts[item] + off) * L + l) & 0xffff) | (kDsShake << 16);
        uint32_t *yrow = st.y + (slot * L + l) * B::YROW_DW;
        if constexpr (SPLIT) {
            SplitState s;
            const uint32_t *seed = reinterpret_cast<const uint32_t *>(st.mr + item * 128 + 64) + parity;
#pragma unroll
            for (int w = 0; w < 25; w++) s.w[w] = w < 8 ? seed[2 * w] : 0;
            if (parity == 0) s.w[8] = nonce;
            else s.w[16] = 0x80000000u;
#pragma unroll 1
            for (int blk = 0; blk < 5; blk++) {
                keccak_f1600_split(s, parity != 0);
                if (on) {
                    detail::static_for<0, 17>([&](auto ic) {
                        constexpr int w = decltype(ic)::v;
                        if (34 * blk + 2 * w + parity < G::ZSZ / 4) yrow[34 * blk + 2 * w + parity] = s.w[w];  // only the ZSZ payload bytes are kept
                    });
                }
            }
        } else {
            KeccakState s;
            keccak_zero(s);
            xor_words<0, 8>(s, reinterpret_cast<const uint64_t *>(st.mr + item * 128 + 64));
            s.lo[8] = nonce;
            s.hi[16] = 0x80000000u;
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
    }
}

// The same for the shortest rounds (a few thousand streams at most: a small batch, or the tail of a large one): TWO streams per
// wavefront on the cooperative permutation (keccak_f1600_coop2, 25 lanes per state) -- the five-permutation chain of a stream is
// the round's first latency, and the cooperative form is the shortest chain there is (~3.5 us per permutation against ~6.5 on a lane pair).
template <int MODE>
__global__ void __launch_bounds__(64) sign_mask_coop_kernel(SignState st, int cur) {
    using G = DG<MODE>;
    using B = SB<MODE>;
    constexpr int L = DP<MODE>::L;
    __shared__ __attribute__((aligned(16))) uint64_t ws[100];
    const size_t total = (size_t)st.count[cur] * L;
    if (blockIdx.x == 0 && threadIdx.x == 0) st.count[cur ^ 1] = 0;  // (as sign_mask_kernel)
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    const CoopLane c = coop_lane(ws, lane);
#pragma unroll 1
    for (size_t base = (size_t)blockIdx.x * 2; base < total; base += (size_t)gridDim.x * 2) {  // block-uniform
        const size_t sidx = base + half;
        const bool on = sidx < total;
        const size_t slot = on ? sidx / L : 0;
        const uint32_t e = st.list[cur][slot];
        const size_t item = e & kEntryItemMask;
        const uint32_t off = e >> kEntryShift;
        const int l = on ? (int)(sidx % L) : 0;
        const uint32_t nonce = (((st.attempts[item] + off) * L + l) & 0xffff) | (kDsShake << 16);
        const uint64_t *seed = reinterpret_cast<const uint64_t *>(st.mr + item * 128 + 64);
        const uint64_t w0 = j < 8 ? seed[j] : j == 8 ? (uint64_t)nonce : j == 16 ? 0x8000000000000000ull : 0ull;
        uint32_t vlo = (uint32_t)w0, vhi = (uint32_t)(w0 >> 32);
        uint32_t *yrow = st.y + (slot * L + l) * B::YROW_DW;
#pragma unroll 1
        for (int blk = 0; blk < 5; blk++) {
            keccak_f1600_coop2<true>(vlo, vhi, c);
            const int d = 34 * blk + 2 * j;
            if (on && j < 17) {  // only the ZSZ payload bytes are kept
                if (d < G::ZSZ / 4) yrow[d] = vlo;
                if (d + 1 < G::ZSZ / 4) yrow[d + 1] = vhi;
            }
        }
    }
}

// wave = one or two entries: y-hat, w = InvNTT(A y-hat), Decompose, w1 (dilithium.go:376-398).
// Two consecutive list entries that belong to the SAME item (attempts a, a + 1 of a lazy pair, or neighbours of a speculative
// round) are processed by one wavefront and every matrix row read (K L rows of 768 bytes, 23 KB for ML-DSA-65) serves both.
// The kernel is a chain of L + K transforms per attempt, each with three exchanges through LDS whose latency a lone wavefront
// waits out (measured: 6.7 cycles per VALU instruction, 1.83 ms per 2^18 ML-DSA-65 attempts, whether the matrix reads are
// shared or not): transforms therefore run TWO AT A TIME (dilithium::ntt2 / invntt2) -- both attempts' polynomial l, both
// attempts' output polynomial i; a single entry pairs its own polynomials (l, l + 1) and (i, i + 1).
// The transformed masks do not stay in registers (up to 2 L polynomials): each lane parks its 4 coefficients of every y-hat,
// folded to 24 bits, in its own 12 bytes of LDS (no other lane touches them).
template <int MODE, int T>
__device__ __forceinline__ void sign_w_entries(const SignState &st, size_t slot, size_t item, uint32_t *xch0, uint32_t *xch1, uint32_t *yl,
                                               const dilithium::LaneZetas &z, int lane) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using B = SB<MODE>;
    constexpr int K = P::K, L = P::L;
    // the exchange buffers belong to this wavefront alone: its LDS instructions execute in order, so the reads of an exchange need
    // no s_waitcnt behind the writes (dilithium_dev.h xch_sync; +1 % signatures/s measured)
    constexpr bool NW = true;
    auto put_w0 = [&](size_t sl, int i, const uint32_t (&a0v)[4]) { store_poly24(st.w0 + sl * B::W0_SLOT_DW + i * kPackedRowDwords, a0v, lane); };
    static_assert(K % 2 == 0, "output polynomials are processed in pairs");
#pragma unroll
    for (int t = 0; t < T; t++)
        if (lane < 16)  // mu in front of the w1 bytes that follow
            reinterpret_cast<uint32_t *>(st.muw1 + (slot + t) * B::MUW1_BYTES)[lane] = reinterpret_cast<const uint32_t *>(st.mr + item * 128)[lane];
    if constexpr (T == 1) {
        // One entry alone (the long rounds by default): y-hat in registers, single transforms -- measured 1.83 ms per 2^18
        // ML-DSA-65 attempts against 1.94 ms for the paired-transform form below on (l, l + 1) / (i, i + 1): with one attempt
        // per wavefront the kernel sits on its matrix reads (4.4 TB/s, profiles/r03_sign_pmc.txt), not on exchange latency.
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
            dilithium::ntt<NW>(yh[l], z, xch0, lane);  // plain y-hat, < 17q
        }
        const uint32_t *arows1 = st.A + st.key_of(item) * K * L * kPackedRowDwords;
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            uint64_t acc[4] = {0, 0, 0, 0};  // lazy 64-bit dot product, one reduction per coefficient
#pragma unroll
            for (int j = 0; j < L; j++) {
                uint32_t a[4];
                load_poly24(a, arows1 + (i * L + j) * kPackedRowDwords, lane);
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] += (uint64_t)a[r] * yh[j][r];
            }
            uint32_t w[4];
#pragma unroll
            for (int r = 0; r < 4; r++) w[r] = dilithium::mont64(acc[r]);
            dilithium::invntt<dilithium::INV256_RR, NW>(w, z, xch0, lane);
            unsigned w1v[4];
            uint32_t a0v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint32_t a1;
                dilithium::decompose<P::GAMMA2>(dilithium::csubq(w[r]), a0v[r], a1);
                w1v[r] = a1;
            }
            put_w0(slot, i, a0v);
            mlkem::stage_bits_l1<G::W1BITS>(xch0, w1v, lane);
            mlkem::store_staged<G::W1BITS>(reinterpret_cast<uint32_t *>(st.muw1 + slot * B::MUW1_BYTES + 64 + G::W1SZ * i), xch0, lane, false);
        }
        __syncthreads();  // xch is reused by the next entry
        return;
    }
    auto load_y = [&](uint32_t (&yh)[4], size_t sl, int l) {
        const uint32_t *yrow = st.y + (sl * L + l) * B::YROW_DW;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t x = G::GAMMA1 - gbits<G::ZBITS>(yrow, kyber::idx_l1(lane, r), G::ZSZ / 4);
            x += (uint32_t)((int32_t)x >> 31) & Q;
            yh[r] = x;
        }
    };
    auto park = [&](const uint32_t (&yh)[4], int row) {  // plain y-hat < 17q, folded below 2^24
        const uint32_t f[4] = {dilithium::fold(yh[0]), dilithium::fold(yh[1]), dilithium::fold(yh[2]), dilithium::fold(yh[3])};
        store_poly24(yl + row * kPackedRowDwords, f, lane);
    };
    if constexpr (T == 2) {
#pragma unroll 1
        for (int l = 0; l < L; l++) {
            uint32_t ya[4], yb[4];
            load_y(ya, slot, l);
            load_y(yb, slot + 1, l);
            dilithium::ntt2<NW>(ya, yb, z, xch0, xch1, lane);
            park(ya, l);
            park(yb, L + l);
        }
    } else {
#pragma unroll 1
        for (int l = 0; l + 1 < L; l += 2) {
            uint32_t ya[4], yb[4];
            load_y(ya, slot, l);
            load_y(yb, slot, l + 1);
            dilithium::ntt2<NW>(ya, yb, z, xch0, xch1, lane);
            park(ya, l);
            park(yb, l + 1);
        }
        if constexpr (L % 2 == 1) {
            uint32_t ya[4];
            load_y(ya, slot, L - 1);
            dilithium::ntt<NW>(ya, z, xch0, lane);
            park(ya, L - 1);
        }
    }
    const uint32_t *arows = st.A + st.key_of(item) * K * L * kPackedRowDwords;
    // Decompose, the low part to w0, the high part packed behind mu (PackW1, pack.go:256-270: the challenge hash absorbs it from
    // there and the hint computation of the finish kernel reads its fields back)
    auto emit = [&](uint32_t (&w)[4], size_t sl, int i, uint32_t *stage) {
        unsigned w1v[4];
        uint32_t a0v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t a1;
            dilithium::decompose<P::GAMMA2>(dilithium::csubq(w[r]), a0v[r], a1);
            w1v[r] = a1;
        }
        put_w0(sl, i, a0v);
        mlkem::stage_bits_l1<G::W1BITS>(stage, w1v, lane);
        mlkem::store_staged<G::W1BITS>(reinterpret_cast<uint32_t *>(st.muw1 + sl * B::MUW1_BYTES + 64 + G::W1SZ * i), stage, lane, false);
    };
    // lazy 64-bit dot products (a < 2^23, y-hat < 2^24, L <= 7 terms), one reduction per coefficient
#pragma unroll 1
    for (int i = 0; i < K; i += (T == 2 ? 1 : 2)) {
        uint64_t acc0[4] = {0, 0, 0, 0}, acc1[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < L; j++) {
            uint32_t a[4], b[4], y0[4], y1[4];
            load_poly24(a, arows + (i * L + j) * kPackedRowDwords, lane);
            load_poly24(y0, yl + j * kPackedRowDwords, lane);
            if constexpr (T == 2) load_poly24(y1, yl + (L + j) * kPackedRowDwords, lane);
            else load_poly24(b, arows + ((i + 1) * L + j) * kPackedRowDwords, lane);
            // Keep y opaque.  With BOTH factors visibly below 2^24 (two load_poly24 results) the AMDGPU backend of ROCm 7.2
            // selects 24-bit multiplies, drops the `& 0xffffff` of the unpacking as implied by them -- and then fuses the
            // products into V_MAD_U64_U32 on the UNMASKED dwords (seen in the ISA of this loop: v_mad_u64_u32 on the raw
            // ds_read / global_load registers): wrong products for every term but the first.  Found by the signing parity
            // tests; an operand of unknown width keeps the masks and the full 32 x 32 -> 64 multiply-add.
            asm volatile("" : "+v"(y0[0]), "+v"(y0[1]), "+v"(y0[2]), "+v"(y0[3]));
            if constexpr (T == 2) asm volatile("" : "+v"(y1[0]), "+v"(y1[1]), "+v"(y1[2]), "+v"(y1[3]));
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc0[r] += (uint64_t)a[r] * y0[r];
                if constexpr (T == 2) acc1[r] += (uint64_t)a[r] * y1[r];
                else acc1[r] += (uint64_t)b[r] * y0[r];
            }
        }
        uint32_t wa[4], wb[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { wa[r] = dilithium::mont64(acc0[r]); wb[r] = dilithium::mont64(acc1[r]); }
        dilithium::invntt2<dilithium::INV256_RR, NW>(wa, wb, z, xch0, xch1, lane);
        emit(wa, slot, i, xch0);
        if constexpr (T == 2) emit(wb, slot + 1, i, xch1);
        else emit(wb, slot, i + 1, xch1);
    }
    __syncthreads();  // the exchange buffers are reused by the next entry
}
// PAIRS = false: the form for the LONG rounds, whose lists hold one entry per item (the host knows from its schedule; a pair that turns
// up anyway is two single entries: same bytes).  Without the paired path the kernel needs neither its registers nor the 2 L rows of
// parked y-hat: ML-DSA-87 / Dilithium5 (L = 7) run 4 wavefronts per SIMD instead of 3 -- the general kernel is held at 3 both by its
// 155 VGPRs and by 13 KB of LDS per wavefront (profiles/r05_sign_ab.txt).
template <int MODE, bool PAIRS = true>
__global__ void __launch_bounds__(64, (PAIRS && DP<MODE>::L > 5) ? 3 : 4) sign_w_kernel(SignState st, int cur) {
    constexpr int L = DP<MODE>::L;
    __shared__ __attribute__((aligned(16))) uint32_t xch0[dilithium::kXchWords];
    __shared__ __attribute__((aligned(16))) uint32_t xch1[dilithium::kXchWords];
    __shared__ __attribute__((aligned(16))) uint32_t yl[PAIRS ? 2 * L * kPackedRowDwords : 4];
    const int lane = threadIdx.x;
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);
    const size_t count = st.count[cur];
#pragma unroll 1
    for (size_t s0 = 2 * (size_t)blockIdx.x; s0 < count; s0 += 2 * (size_t)gridDim.x) {
        const size_t item0 = st.list[cur][s0] & kEntryItemMask;
        const bool two = s0 + 1 < count;
        const size_t item1 = two ? (size_t)(st.list[cur][s0 + 1] & kEntryItemMask) : item0;
        if (PAIRS && two && item1 == item0) {
            if constexpr (PAIRS) sign_w_entries<MODE, 2>(st, s0, item0, xch0, xch1, yl, z, lane);
        } else {
            sign_w_entries<MODE, 1>(st, s0, item0, xch0, xch1, yl, z, lane);
            if (two) sign_w_entries<MODE, 1>(st, s0 + 1, item1, xch0, xch1, yl, z, lane);
        }
    }
}

// Which entries a pass of the challenge / finish kernels handles.  A LAZY round (pairs, sign_round_lazy) runs them twice: pass 0
// takes the first attempt of every pair, pass 1 the second -- and only of the items whose first attempt was rejected (best is
// still kNoSuccess); every other round does everything in pass 0.  Pairs start at even slots (the compaction reserves k
// entries per survivor from a counter that starts at zero), so in a lazy round the slot's parity is the entry's `off`.
struct PassMap {
    size_t nwork;   // entries of this pass
    bool lazy;
    int pass;
    __device__ __forceinline__ PassMap(const SignState &st, int cur, int pass_) : pass(pass_) {
        const size_t count = st.count[cur];
        lazy = sign_round_lazy(count, st.kk[cur], st.spec_target);
        nwork = lazy ? (count + 1 - (size_t)pass) / 2 : (pass == 0 ? count : 0);
    }
    __device__ __forceinline__ size_t slot(size_t a) const { return lazy ? 2 * a + (size_t)pass : a; }
};

// lane = entry: c~ = SHAKE256(mu || w1)[:CT] and the first SampleInBall block (dilithium.go:400-405)
template <int MODE, bool SPLIT = false>
__global__ void __launch_bounds__(256) sign_challenge_kernel(SignState st, int cur, int pass) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using B = SB<MODE>;
    const PassMap pm(st, cur, pass);
    constexpr int PER = SPLIT ? 128 : 256;  // SPLIT: an entry per lane pair (see sign_mask_kernel)
    const int parity = threadIdx.x & 1;
#pragma unroll 1
    for (size_t base = (size_t)blockIdx.x * PER; base < pm.nwork; base += (size_t)gridDim.x * PER) {
        const size_t w = base + (SPLIT ? threadIdx.x >> 1 : threadIdx.x);
        if (w >= pm.nwork) continue;  // (both lanes of a pair take the same branches)
        const size_t a = pm.slot(w);
        const uint32_t e = st.list[cur][a];
        if (st.best[e & kEntryItemMask] < (e >> kEntryShift)) continue;  // a lower attempt of the item has already succeeded
        if constexpr (SPLIT) {
            SplitState s;
            mlkem::split_sponge17<G::MUW1 / 8>(s, reinterpret_cast<const uint32_t *>(st.muw1 + a * B::MUW1_BYTES) + parity, kDsShake, parity != 0);
            uint32_t *cb = reinterpret_cast<uint32_t *>(st.cb + a * B::CB_BYTES) + parity;
#pragma unroll
            for (int i = 0; i < P::CT / 8; i++) cb[2 * i] = s.w[i];
#pragma unroll
            for (int i = P::CT / 8; i < 25; i++) s.w[i] = 0;  // the SampleInBall sponge absorbs c~: same state
            if (parity == 0) s.w[P::CT / 8] ^= kDsShake;
            else s.w[16] ^= 0x80000000u;
            keccak_f1600_split(s, parity != 0);
#pragma unroll
            for (int i = 0; i < 25; i++) cb[30 + 2 * i] = s.w[i];  // ball state at byte 120
        } else {
            KeccakState s;
            sponge17_words<G::MUW1 / 8>(s, reinterpret_cast<const uint64_t *>(st.muw1 + a * B::MUW1_BYTES), kDsShake);
            uint64_t *cb = reinterpret_cast<uint64_t *>(st.cb + a * B::CB_BYTES);
            store_words<0, P::CT / 8>(cb, s);
#pragma unroll
            for (int i = P::CT / 8; i < 25; i++) { s.lo[i] = 0; s.hi[i] = 0; }  // the SampleInBall sponge absorbs c~: same state
            s.lo[P::CT / 8] ^= kDsShake;
            s.hi[16] ^= 0x80000000u;
            keccak_f1600(s);
            store_words<0, 25>(cb + 15, s);  // ball state at byte 120
        }
    }
}

// ... and two entries per wavefront on the cooperative permutation for the shortest rounds (see sign_mask_coop_kernel): the 7-9
// permutations of c~ = H(mu || w1) and the SampleInBall block are the round's longest chain.
template <int MODE>
__global__ void __launch_bounds__(64) sign_challenge_coop_kernel(SignState st, int cur, int pass) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using B = SB<MODE>;
    static_assert(G::MUW1 % 8 == 0, "mu || w1 is absorbed as 64-bit words");
    __shared__ __attribute__((aligned(16))) uint64_t ws[100];
    const PassMap pm(st, cur, pass);
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    const CoopLane c = coop_lane(ws, lane);
#pragma unroll 1
    for (size_t base = (size_t)blockIdx.x * 2; base < pm.nwork; base += (size_t)gridDim.x * 2) {  // block-uniform
        const size_t w = base + half;
        const bool on = w < pm.nwork;
        const size_t a = pm.slot(on ? w : base);
        const uint32_t e = st.list[cur][a];
        const bool live = on && !(st.best[e & kEntryItemMask] < (e >> kEntryShift));  // a lower attempt of the item has already succeeded
        const uint64_t *src = reinterpret_cast<const uint64_t *>(st.muw1 + a * B::MUW1_BYTES);
        uint32_t vlo, vhi;
        mlkem::coop_sponge17<true>(vlo, vhi, [&](int k) { return src[k]; }, G::MUW1 / 8, kDsShake, c, j);
        uint64_t *cb = reinterpret_cast<uint64_t *>(st.cb + a * B::CB_BYTES);
        if (live && j < P::CT / 8) cb[j] = ((uint64_t)vhi << 32) | vlo;
        if (j >= P::CT / 8) { vlo = 0; vhi = 0; }  // the SampleInBall sponge absorbs c~: same state
        if (j == P::CT / 8) vlo ^= kDsShake;
        if (j == 16) vhi ^= 0x80000000u;
        keccak_f1600_coop2<true>(vlo, vhi, c);
        if (live && j < 25) cb[15 + j] = ((uint64_t)vhi << 32) | vlo;  // ball state at byte 120
    }
}

// wave = entry: the three rejection tests, hints, signature (dilithium.go:407-455, :84-88).
// One entry's work (sign_finish_kernel below runs it inlined for a workgroup's first entry and through a non-inlined copy for
// any further one).
// WS (wave sync): the body runs on ONE wavefront of a multi-wavefront workgroup (sign_round_chain_kernel) whose buffers here are private to
// it: every ordering point is the wave-level no-wait form instead of a workgroup barrier.
template <int MODE, bool WS = false>
__device__ __forceinline__ void sign_finish_body(const SignState &st, int cur, uint8_t *__restrict__ sig, size_t slot, bool direct, uint32_t *xch,
                                                 uint8_t *zpk, uint8_t *hbytes, uint8_t *blk) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using B = SB<MODE>;
    constexpr int K = P::K, L = P::L;
    const int lane = threadIdx.x & 63;
    auto fsync = [] {
        if constexpr (WS) { __builtin_amdgcn_s_waitcnt(0); wave_lds_order(); }
        else __syncthreads();
    };
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);
    const uint32_t e = st.list[cur][slot];
    const size_t item = e & kEntryItemMask;
    const uint32_t off = e >> kEntryShift;
    if (st.best[item] < off) return;  // a lower attempt of the item has already succeeded (the challenge kernel skipped it too)
    const uint8_t *cb = st.cb + slot * B::CB_BYTES;
    uint32_t chat[4];
    sample_in_ball_hat<MODE, true, false, true, WS>(chat, cb + 120, blk, xch, z, lane);
    const uint32_t *sec = st.sec + st.key_of(item) * (L + 2 * K) * kPackedRowDwords;
    constexpr bool NW = true;  // (wave-private exchange buffer: see sign_w_entries)
    uint32_t *w0 = st.w0 + slot * B::W0_SLOT_DW;
    uint32_t *best = st.best;
    // Every loop below is software-pipelined by hand: the packed row of polynomial i + 1 (and its w0 / w1 words) is requested
    // before the inverse transform of polynomial i, so the wave has loads in flight while it computes (the attempts are
    // latency-bound chains: load, product, three LDS exchanges, compare).
    auto load_row = [&](uint32_t (&raw)[3], const uint32_t *row) {
        const uint32_t *p = row + 3 * lane;
        raw[0] = p[0]; raw[1] = p[1]; raw[2] = p[2];
    };
    auto mul_c = [&](uint32_t (&t)[4], const uint32_t (&raw)[3]) {
        const uint32_t sv[4] = {raw[0] & 0xffffffu, (raw[0] >> 24) | ((raw[1] & 0xffffu) << 8), (raw[1] >> 16) | ((raw[2] & 0xffu) << 16), raw[2] >> 8};
#pragma unroll
        for (int r = 0; r < 4; r++) t[r] = dilithium::fold(dilithium::mont32(sv[r], chat[r]));
        dilithium::invntt<dilithium::INV256_R, NW>(t, z, xch, lane);
    };
    bool bad = false;
    // The three norm tests decide together and their order is free; the reference's (r0, z, hints: dilithium.go:409-450) is also
    // the cheap one: for ML-DSA-65 the r0 test rejects 68 % of the attempts (1536 coefficients against gamma2 - beta), the z
    // test 38 % (1280 against gamma1 - beta), so r0 first costs ~6.4 inverse transforms per attempt, z first ~7.8 (measured:
    // 14.7 vs 16.3 ms of finish kernels per 2^18 signatures).
    // r0 = w0 - c s2.  The differences are parked in LDS (the area that will hold the packed z: not live yet) and go back to the
    // slot's w0 area only when ALL of them are in range -- a third of the attempts; the hint computation reads them there.
    {
        uint32_t *r0l = reinterpret_cast<uint32_t *>(zpk);
        uint32_t raw[3], wraw[3];
        load_row(raw, sec + L * kPackedRowDwords);
        load_row(wraw, w0);
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            uint32_t raw_n[3] = {0, 0, 0}, wraw_n[3] = {0, 0, 0};
            if (i + 1 < K) {
                load_row(raw_n, sec + (L + i + 1) * kPackedRowDwords);
                load_row(wraw_n, w0 + (i + 1) * kPackedRowDwords);
            }
            uint32_t t[4];
            mul_c(t, raw);
            const uint32_t wv[4] = {wraw[0] & 0xffffffu, (wraw[0] >> 24) | ((wraw[1] & 0xffffu) << 8), (wraw[1] >> 16) | ((wraw[2] & 0xffu) << 16), wraw[2] >> 8};
            uint32_t v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                v[r] = dilithium::normalize(wv[r] + (2 * Q - t[r]));
                bad |= dilithium::exceeds(v[r], P::GAMMA2 - G::BETA);
            }
            if (__any(bad)) break;  // one polynomial out of range decides the attempt: the remaining inverse transforms are moot
            store_poly24(r0l + i * kPackedRowDwords, v, lane);
#pragma unroll
            for (int r = 0; r < 3; r++) { raw[r] = raw_n[r]; wraw[r] = wraw_n[r]; }
        }
        if (__any(bad)) return;
        fsync();
        for (int d = lane; d < K * kPackedRowDwords; d += 64) w0[d] = r0l[d];
        fsync();  // ... before z is packed into the same area
    }
    if (__any(bad)) return;
    // z = y + c s1.  A polynomial that passes its norm test is bit-packed at once as it will appear in the signature (pack.go:202-254)
    // into the LDS area the parked r0 has just left: nothing of z stays in registers, and the loop stays rolled.
    {
        auto load_y = [&](uint32_t (&yv)[4], int l) {
            const uint32_t *yrow = st.y + (slot * L + l) * B::YROW_DW;
#pragma unroll
            for (int r = 0; r < 4; r++) yv[r] = gbits<G::ZBITS>(yrow, kyber::idx_l1(lane, r), G::ZSZ / 4);
        };
        uint32_t raw[3], yv[4];
        load_row(raw, sec);
        load_y(yv, 0);
#pragma unroll 1
        for (int l = 0; l < L; l++) {
            uint32_t raw_n[3] = {0, 0, 0}, yv_n[4] = {0, 0, 0, 0};
            if (l + 1 < L) {
                load_row(raw_n, sec + (l + 1) * kPackedRowDwords);
                load_y(yv_n, l + 1);
            }
            uint32_t t[4];
            mul_c(t, raw);
            unsigned f[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint32_t y = G::GAMMA1 - yv[r];
                y += (uint32_t)((int32_t)y >> 31) & Q;
                const uint32_t zz = dilithium::normalize(t[r] + y);
                bad |= dilithium::exceeds(zz, G::GAMMA1 - G::BETA);
                uint32_t fr = G::GAMMA1 - zz;
                fr += (uint32_t)((int32_t)fr >> 31) & Q;
                f[r] = fr;
            }
            if (__any(bad)) break;
            mlkem::stage_bits_l1<G::ZBITS, WS>(xch, f, lane);
            for (int d = lane; d < 8 * G::ZBITS; d += 64) reinterpret_cast<uint32_t *>(zpk + G::ZSZ * l)[d] = xch[d];
#pragma unroll
            for (int r = 0; r < 3; r++) raw[r] = raw_n[r];
#pragma unroll
            for (int r = 0; r < 4; r++) yv[r] = yv_n[r];
        }
    }
    if (__any(bad)) return;
    // c t0, hints
    unsigned pop = 0;
    fsync();
    for (int i = lane; i < 24; i += 64) reinterpret_cast<uint32_t *>(hbytes)[i] = 0;
    fsync();
    {
        uint32_t raw[3];
        load_row(raw, sec + (L + K) * kPackedRowDwords);
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            uint32_t raw_n[3] = {0, 0, 0}, wv[4], r1v[4];
            if (i + 1 < K) load_row(raw_n, sec + (L + K + i + 1) * kPackedRowDwords);
            load_poly24(wv, w0 + i * kPackedRowDwords, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                r1v[r] = get_bits32<G::W1BITS>(st.muw1 + slot * B::MUW1_BYTES + 64 + G::W1SZ * i, kyber::idx_l1(lane, r));  // w1 as packed by the w kernel
            }
            uint32_t t[4];
            mul_c(t, raw);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nidx = kyber::idx_l1(lane, r);
                const uint32_t ct0 = dilithium::csubq(t[r]);
                bad |= dilithium::exceeds(ct0, P::GAMMA2);
                const uint32_t v = dilithium::csubq(wv[r] + ct0);
                const uint32_t r1 = r1v[r];
                const bool hbit = dilithium::make_hint<P::GAMMA2>(v, r1);
                const unsigned long long mask = __ballot(hbit);
                if (hbit) {
                    const unsigned hs = pop + (unsigned)__popcll(mask & ((1ull << lane) - 1));
                    if (hs < (unsigned)P::OMEGA) hbytes[hs] = (uint8_t)nidx;
                }
                pop += (unsigned)__popcll(mask);
            }
            if (lane == 0) hbytes[P::OMEGA + i] = (uint8_t)(pop < 255 ? pop : 255);
#pragma unroll
            for (int r = 0; r < 3; r++) raw[r] = raw_n[r];
        }
    }
    if (__any(bad) || pop > (unsigned)P::OMEGA) return;
    fsync();       // every lane is done with w0
    // `direct`: this attempt is the lowest one of its item that can still succeed (one attempt per item, or a lazy pair,
    // whose second attempt only runs after the first was rejected): the signature goes straight out.  Otherwise it is parked
    // in the slot's w0 area and the commit kernel copies the lowest successful attempt's.
    uint8_t *sg = direct ? sig + item * G::SIG : reinterpret_cast<uint8_t *>(w0);
    for (int b = lane; b < P::CT; b += 64) sg[b] = cb[b];
    for (int b = lane; b < L * G::ZSZ; b += 64) sg[P::CT + b] = zpk[b];
    for (int b = lane; b < P::OMEGA + K; b += 64) sg[P::CT + L * G::ZSZ + b] = hbytes[b];
    if (lane == 0) atomicMin(&best[item], off);
}
// The grid is (an upper estimate of) one workgroup per entry of the pass, so a workgroup normally handles exactly one: that one
// runs inlined in the kernel.  Entries beyond the grid -- only when the schedule's estimate was too small -- go through a
// non-inlined copy: inlined into a grid-stride loop the body made the loop's register allocation grow from 102 to 164 VGPRs,
// and as a function it reaches its arrays through generic pointers (FLAT instructions, which also count against the LDS
// counter: every wait for an LDS exchange then waits for the global loads in flight -- 8.2 cycles per VALU instruction).
template <int MODE>
__device__ __noinline__ void sign_finish_entry(const SignState &st, int cur, uint8_t *__restrict__ sig, size_t slot, bool direct, uint32_t *xch,
                                               uint8_t *zpk, uint8_t *hbytes, uint8_t *blk) {
    sign_finish_body<MODE>(st, cur, sig, slot, direct, xch, zpk, hbytes, blk);
}
template <int MODE>
__global__ void __launch_bounds__(64, 4) sign_finish_kernel(SignState st, int cur, int pass, uint8_t *__restrict__ sig) {
    using G = DG<MODE>;
    constexpr int K = DP<MODE>::K, L = DP<MODE>::L;
    __shared__ __attribute__((aligned(16))) uint32_t xch[dilithium::kXchWords];
    __shared__ __attribute__((aligned(16))) uint8_t zpk[K * kPackedRowDwords * 4 > L * G::ZSZ ? K * kPackedRowDwords * 4 : L * G::ZSZ];  // packed z; before that, r0
    __shared__ __attribute__((aligned(16))) uint8_t hbytes[96];
    __shared__ __attribute__((aligned(16))) uint8_t blk[144];
    static_assert(SB<MODE>::W0_BYTES >= (size_t)G::SIG, "a slot's w0 area can park its signature");
    const PassMap pm(st, cur, pass);
    const bool direct = pm.lazy || st.kk[cur] == 1;
    if (blockIdx.x >= pm.nwork) return;
    sign_finish_body<MODE>(st, cur, sig, pm.slot(blockIdx.x), direct, xch, zpk, hbytes, blk);
    if ((size_t)blockIdx.x + gridDim.x >= pm.nwork) return;
    const SignState escaped = st;  // a copy for the call by reference: taking st's own address would turn every pointer field
                                   // of the inlined path above into a generic pointer as well
#pragma unroll 1
    for (size_t a = (size_t)blockIdx.x + gridDim.x; a < pm.nwork; a += gridDim.x) {
        __syncthreads();  // the previous entry is done with the LDS buffers
        sign_finish_entry<MODE>(escaped, cur, sig, pm.slot(a), direct, xch, zpk, hbytes, blk);
    }
}

// The norm tests, hints and the signature of ONE entry by a workgroup of K wavefronts (sign_round_chain_kernel): what sign_finish_body does
// on one wavefront -- 2K + L inverse transforms one after the other -- with a wavefront per polynomial: r0 = w0 - c s2 row by row, z = y + c s1
// polynomial by polynomial, c t0 and the hints row by row, a workgroup barrier and a shared verdict between the three tests (their order
// is free: all three must pass), the rows' hint counts prefix-summed through LDS.  Every wavefront builds c-hat itself (SampleInBall + one
// transform, ~3 us side by side instead of a broadcast).  Same bytes as sign_finish_body.  `fin`: [0] / [K + 1] / [K + 2] the shared verdicts of
// the r0, z and hint tests, [1 + i] row i's hint count.  ONE WORD PER TEST: each is zeroed by the caller before the phases
// start, written only during its own phase and read only behind that phase's barrier -- with a single word, a wavefront that was slow
// to read it after the r0 barrier could pick up a sibling's veto from the NEXT phase, leave early and mis-pair every barrier after it.
// Every branch on `fin` is therefore workgroup-uniform.
template <int MODE>
__device__ __forceinline__ void sign_finish_rows(const SignState &st, uint8_t *__restrict__ sig, size_t slot, size_t item, uint32_t off, bool direct,
                                                 int wave, int lane, uint32_t *xch, const dilithium::LaneZetas &z, uint8_t *zpk, uint8_t *hbytes,
                                                 uint8_t *blk, unsigned *fin) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using B = SB<MODE>;
    constexpr int K = P::K, L = P::L;
    constexpr bool NW = true;
    const uint8_t *cb = st.cb + slot * B::CB_BYTES;
    uint32_t chat[4];
    sample_in_ball_hat<MODE, true, false, true, true>(chat, cb + 120, blk, xch, z, lane);
    const uint32_t *sec = st.sec + st.key_of(item) * (L + 2 * K) * kPackedRowDwords;
    uint32_t *w0 = st.w0 + slot * B::W0_SLOT_DW;
    auto mul_c = [&](uint32_t (&t)[4], const uint32_t *row) {
        uint32_t sv[4];
        load_poly24(sv, row, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) t[r] = dilithium::fold(dilithium::mont32(sv[r], chat[r]));
        dilithium::invntt<dilithium::INV256_R, NW>(t, z, xch, lane);
    };
    auto veto = [&](bool bad, int word) {
        if (__any(bad) && lane == 0) atomicOr(&fin[word], 1u);
    };
    // ---- r0 = w0 - c s2, row `wave`; parked in LDS (the area that will hold the packed z) and written back only when every row is in range ----
    uint32_t *r0l = reinterpret_cast<uint32_t *>(zpk);
    {
        const int i = wave;
        uint32_t t[4], wv[4], v[4];
        load_poly24(wv, w0 + i * kPackedRowDwords, lane);
        mul_c(t, sec + (L + i) * kPackedRowDwords);
        bool bad = false;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[r] = dilithium::normalize(wv[r] + (2 * Q - t[r]));
            bad |= dilithium::exceeds(v[r], P::GAMMA2 - G::BETA);
        }
        store_poly24(r0l + i * kPackedRowDwords, v, lane);
        veto(bad, 0);
    }
    if (threadIdx.x < 24) reinterpret_cast<uint32_t *>(hbytes)[threadIdx.x] = 0;
    __syncthreads();
    if (fin[0]) return;
    for (int d = threadIdx.x; d < K * kPackedRowDwords; d += K * 64) w0[d] = r0l[d];
    __threadfence_block();
    __syncthreads();  // ... before z is packed into the same area; the hint phase reads r0 back from the slot
    // ---- z = y + c s1, polynomial `wave`, bit-packed as it will appear in the signature (pack.go:202-254) ----
    if (wave < L) {
        const int l = wave;
        const uint32_t *yrow = st.y + (slot * L + l) * B::YROW_DW;
        uint32_t yv[4], t[4];
#pragma unroll
        for (int r = 0; r < 4; r++) yv[r] = gbits<G::ZBITS>(yrow, kyber::idx_l1(lane, r), G::ZSZ / 4);
        mul_c(t, sec + l * kPackedRowDwords);
        unsigned f[4];
        bool bad = false;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t y = G::GAMMA1 - yv[r];
            y += (uint32_t)((int32_t)y >> 31) & Q;
            const uint32_t zz = dilithium::normalize(t[r] + y);
            bad |= dilithium::exceeds(zz, G::GAMMA1 - G::BETA);
            uint32_t fr = G::GAMMA1 - zz;
            fr += (uint32_t)((int32_t)fr >> 31) & Q;
            f[r] = fr;
        }
        mlkem::stage_bits_l1<G::ZBITS, NW>(xch, f, lane);
        for (int d = lane; d < 8 * G::ZBITS; d += 64) reinterpret_cast<uint32_t *>(zpk + G::ZSZ * l)[d] = xch[d];
        veto(bad, K + 1);
    }
    __syncthreads();
    if (fin[K + 1]) return;
    // ---- c t0 and the hints of row `wave` (dilithium.go:431-450): the row's hint bits as four ballots, its count to LDS ----
    unsigned long long hmask[4];
    unsigned count = 0;
    {
        const int i = wave;
        uint32_t t[4], wv[4];
        load_poly24(wv, w0 + i * kPackedRowDwords, lane);
        mul_c(t, sec + (L + K + i) * kPackedRowDwords);
        bool bad = false;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t r1 = get_bits32<G::W1BITS>(st.muw1 + slot * B::MUW1_BYTES + 64 + G::W1SZ * i, kyber::idx_l1(lane, r));  // w1 as packed in phase B2
            const uint32_t ct0 = dilithium::csubq(t[r]);
            bad |= dilithium::exceeds(ct0, P::GAMMA2);
            const uint32_t v = dilithium::csubq(wv[r] + ct0);
            hmask[r] = __ballot(dilithium::make_hint<P::GAMMA2>(v, r1));
            count += (unsigned)__popcll(hmask[r]);
        }
        if (lane == 0) fin[1 + i] = count;
        veto(bad, K + 2);
    }
    __syncthreads();
    if (fin[K + 2]) return;
    unsigned before = 0, total = 0;
#pragma unroll
    for (int i = 0; i < K; i++) {
        const unsigned ci = fin[1 + i];
        if (i < wave) before += ci;
        total += ci;
    }
    if (total > (unsigned)P::OMEGA) return;  // (workgroup-uniform: every wavefront sums the same counts)
    {
        unsigned pop = before;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if ((hmask[r] >> lane) & 1) hbytes[pop + (unsigned)__popcll(hmask[r] & ((1ull << lane) - 1))] = (uint8_t)kyber::idx_l1(lane, r);
            pop += (unsigned)__popcll(hmask[r]);
        }
        if (lane == 0) hbytes[P::OMEGA + wave] = (uint8_t)pop;
    }
    __syncthreads();
    // `direct`: the signature goes straight out; otherwise it is parked in the slot's w0 area and the commit step copies the lowest
    // successful attempt's (sign_finish_body)
    uint8_t *sg = direct ? sig + item * G::SIG : reinterpret_cast<uint8_t *>(w0);
    const int nthr = K * 64;
    for (int b = threadIdx.x; b < P::CT; b += nthr) sg[b] = cb[b];
    for (int b = threadIdx.x; b < L * G::ZSZ; b += nthr) sg[P::CT + b] = zpk[b];
    for (int b = threadIdx.x; b < P::OMEGA + K; b += nthr) sg[P::CT + L * G::ZSZ + b] = hbytes[b];
    if (threadIdx.x == 0) atomicMin(&st.best[item], off);
}

template <int MODE>
__device__ __forceinline__ void sign_commit_body(const SignState &st, int cur, uint8_t *__restrict__ sig, size_t first, size_t stride);
__device__ __forceinline__ void sign_compact_body(const SignState &st, int cur, int last, unsigned block, unsigned nblocks, unsigned nthreads);

// ---- a short round in ONE launch -------------------------------------------------------------------------------------------
// mask -> w -> challenge -> finish of one list entry by one workgroup of K wavefronts, for rounds of a few hundred entries at most
// (one signature with a prepared key: 64 speculative attempts).  Such a round is latency from end to end -- four dependent launches
// of a few microseconds of work each -- and a single wavefront per entry walks L + K transforms one after the other.  Here:
//   A   ExpandMask: two streams per wavefront on the cooperative permutation, ceil(L / 2) wavefronts side by side (as sign_mask_coop_kernel)
//   B1  y-hat_l = NTT(y_l): a wavefront per l, parked in LDS (pack24)
//   B2  w_i = InvNTT(sum_j A_ij y-hat_j), Decompose, w0 / w1: a wavefront per ROW i
//   C   c~ = H(mu || w1) and the first SampleInBall block: wavefront 0 on the cooperative permutation (7-9 dependent permutations)
//   D   the norm tests, hints and the signature: a wavefront per polynomial again (sign_finish_rows)
// with a workgroup barrier between the phases (data crosses wavefronts there: y through global memory -- the finish phase wants it
// there anyway --, y-hat through LDS, w0 / w1 through the entry's global slots).  Same bytes as the four kernels (the per-entry
// arithmetic is theirs); commit and compact follow as separate launches.
template <int MODE>
__global__ void __launch_bounds__(DP<MODE>::K * 64) sign_round_chain_kernel(SignState st, int cur, uint8_t *__restrict__ sig, int last_round) {
    using G = DG<MODE>;
    using P = DP<MODE>;
    using B = SB<MODE>;
    constexpr int K = P::K, L = P::L, MW = (L + 1) / 2;
    static_assert(K >= L, "a wavefront per row also gives a wavefront per mask polynomial");
    __shared__ __attribute__((aligned(16))) uint32_t xch_all[K][dilithium::kXchWords];
    __shared__ __attribute__((aligned(16))) uint32_t yhat[L * kPackedRowDwords];
    __shared__ __attribute__((aligned(16))) uint64_t coop_ws[MW][100];
    __shared__ __attribute__((aligned(16))) uint8_t zpk[K * kPackedRowDwords * 4 > L * G::ZSZ ? K * kPackedRowDwords * 4 : L * G::ZSZ];
    __shared__ __attribute__((aligned(16))) uint8_t hbytes[96];
    __shared__ __attribute__((aligned(16))) uint8_t blk_all[K][144];
    __shared__ unsigned fin[3 + K];  // (sign_finish_rows: three verdict words and K hint counts)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t *xch = xch_all[wave];
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);
    const size_t count = st.count[cur];
    if (blockIdx.x == 0 && threadIdx.x == 0) st.count[cur ^ 1] = 0;  // filled by this round's compact kernel (the mask kernel's duty otherwise)
    const bool direct = st.kk[cur] == 1;  // (lazy rounds never come here)
    constexpr bool NW = true;
#pragma unroll 1
    for (size_t slot = blockIdx.x; slot < count; slot += gridDim.x) {  // block-uniform
        const uint32_t e = st.list[cur][slot];
        const size_t item = e & kEntryItemMask;
        const uint32_t off = e >> kEntryShift;
        // ---- A: y = ExpandMask(rho'', L * attempt + l), two polynomials per wavefront ----
        if (wave < MW) {
            const int half = lane >> 5, j = lane & 31;
            const CoopLane c = coop_lane(coop_ws[wave], lane);
            const int l = 2 * wave + half;
            const bool on = l < L;
            const uint32_t nonce = (((st.attempts[item] + off) * L + (on ? l : 0)) & 0xffff) | (kDsShake << 16);
            const uint64_t *seed = reinterpret_cast<const uint64_t *>(st.mr + item * 128 + 64);
            const uint64_t w0 = j < 8 ? seed[j] : j == 8 ? (uint64_t)nonce : j == 16 ? 0x8000000000000000ull : 0ull;
            uint32_t vlo = (uint32_t)w0, vhi = (uint32_t)(w0 >> 32);
            uint32_t *yrow = st.y + (slot * L + (on ? l : 0)) * B::YROW_DW;
#pragma unroll 1
            for (int blk_i = 0; blk_i < 5; blk_i++) {
                keccak_f1600_coop2<NW>(vlo, vhi, c);
                const int d = 34 * blk_i + 2 * j;
                if (on && j < 17) {  // only the ZSZ payload bytes are kept
                    if (d < G::ZSZ / 4) yrow[d] = vlo;
                    if (d + 1 < G::ZSZ / 4) yrow[d + 1] = vhi;
                }
            }
        }
        if (wave == K - 1 && lane < 16)  // mu in front of the w1 bytes that follow (an idle wavefront's job)
            reinterpret_cast<uint32_t *>(st.muw1 + slot * B::MUW1_BYTES)[lane] = reinterpret_cast<const uint32_t *>(st.mr + item * 128)[lane];
        __threadfence_block();
        __syncthreads();
        // ---- B1: y-hat_l, a wavefront per polynomial ----
        if (wave < L) {
            const uint32_t *yrow = st.y + (slot * L + wave) * B::YROW_DW;
            uint32_t yh[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint32_t x = G::GAMMA1 - gbits<G::ZBITS>(yrow, kyber::idx_l1(lane, r), G::ZSZ / 4);
                x += (uint32_t)((int32_t)x >> 31) & Q;
                yh[r] = x;
            }
            dilithium::ntt<NW>(yh, z, xch, lane);  // plain y-hat, < 17q
            const uint32_t f[4] = {dilithium::fold(yh[0]), dilithium::fold(yh[1]), dilithium::fold(yh[2]), dilithium::fold(yh[3])};
            store_poly24(yhat + wave * kPackedRowDwords, f, lane);
        }
        __syncthreads();
        // ---- B2: row `wave` of w = A y-hat, Decompose ----
        {
            const int i = wave;
            const uint32_t *arow = st.A + st.key_of(item) * K * L * kPackedRowDwords + (size_t)i * L * kPackedRowDwords;
            uint64_t acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < L; j++) {
                uint32_t a[4], y0[4];
                load_poly24(a, arow + j * kPackedRowDwords, lane);
                load_poly24(y0, yhat + j * kPackedRowDwords, lane);
                asm volatile("" : "+v"(y0[0]), "+v"(y0[1]), "+v"(y0[2]), "+v"(y0[3]));  // keep y opaque: see sign_w_entries (24-bit multiply selection)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] += (uint64_t)a[r] * y0[r];
            }
            uint32_t w[4];
#pragma unroll
            for (int r = 0; r < 4; r++) w[r] = dilithium::mont64(acc[r]);
            dilithium::invntt<dilithium::INV256_RR, NW>(w, z, xch, lane);
            unsigned w1v[4];
            uint32_t a0v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint32_t a1;
                dilithium::decompose<P::GAMMA2>(dilithium::csubq(w[r]), a0v[r], a1);
                w1v[r] = a1;
            }
            store_poly24(st.w0 + slot * B::W0_SLOT_DW + i * kPackedRowDwords, a0v, lane);
            mlkem::stage_bits_l1<G::W1BITS, NW>(xch, w1v, lane);
            mlkem::store_staged<G::W1BITS>(reinterpret_cast<uint32_t *>(st.muw1 + slot * B::MUW1_BYTES + 64 + G::W1SZ * i), xch, lane, false);
        }
        __threadfence_block();
        __syncthreads();
        // ---- C: the challenge, wavefront 0 alone (a chain of dependent permutations) ----
        if (threadIdx.x == 0) fin[0] = fin[K + 1] = fin[K + 2] = 0;  // (the barrier behind phase C publishes them; nobody reads them before)
        if (wave == 0) {
            const int j = lane & 31;
            const CoopLane c = coop_lane(coop_ws[0], lane);
            const uint64_t *src = reinterpret_cast<const uint64_t *>(st.muw1 + slot * B::MUW1_BYTES);
            uint32_t vlo, vhi;
            mlkem::coop_sponge17<true>(vlo, vhi, [&](int k) { return src[k]; }, G::MUW1 / 8, kDsShake, c, j);  // (both halves carry the same sponge)
            uint64_t *cb = reinterpret_cast<uint64_t *>(st.cb + slot * B::CB_BYTES);
            if (lane < P::CT / 8) cb[lane] = ((uint64_t)vhi << 32) | vlo;
            if (j >= P::CT / 8) { vlo = 0; vhi = 0; }  // the SampleInBall sponge absorbs c~: same state
            if (j == P::CT / 8) vlo ^= kDsShake;
            if (j == 16) vhi ^= 0x80000000u;
            keccak_f1600_coop2<true>(vlo, vhi, c);
            if (lane < 25) cb[15 + lane] = ((uint64_t)vhi << 32) | vlo;  // ball state at byte 120
        }
        __threadfence_block();  // (every wavefront reads c~ and the ball state back from the slot)
        __syncthreads();
        // ---- D: the norm tests, hints and the signature, a wavefront per polynomial ----
        sign_finish_rows<MODE>(st, sig, slot, item, off, direct, wave, lane, xch, z, zpk, hbytes, blk_all[wave], fin);
        __syncthreads();  // the LDS buffers are reused by the workgroup's next entry
    }
    // ---- the LAST workgroup to get here commits the round's lowest successful attempts and builds the next list itself (two launches
    // less): every workgroup releases its writes (signatures parked in its slots, best[]) at agent scope and takes a ticket; the one
    // that draws the last ticket acquires and does what sign_commit_kernel / sign_compact_kernel do, on a list of a few hundred entries ----
    __shared__ int is_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned ticket = atomicAdd(st.chain_done, 1u);
        is_last = ticket == gridDim.x - 1;
        if (is_last) __threadfence();
    }
    __syncthreads();
    if (!is_last) return;
    sign_commit_body<MODE>(st, cur, sig, (size_t)wave, (size_t)K);
    __threadfence_block();
    __syncthreads();
    sign_compact_body(st, cur, last_round, 0u, 1u, (unsigned)(K * 64));
    if (threadIdx.x == 0) *st.chain_done = 0;  // for the next one-launch round (kernels of a stream run one after the other)
}

// wave = entry, rounds with several attempts per item only (k == 1: every block leaves at once): the lowest successful
// attempt's parked signature -> sig
// (first / stride: which entries this wavefront takes -- blockIdx.x / gridDim.x for sign_commit_kernel; the wavefront's number / the
// workgroup's wavefronts when the last workgroup of a one-launch round commits itself)
template <int MODE>
__device__ __forceinline__ void sign_commit_body(const SignState &st, int cur, uint8_t *__restrict__ sig, size_t first, size_t stride) {
    using G = DG<MODE>;
    const size_t count = st.count[cur];
    if (st.kk[cur] == 1 || sign_round_lazy(count, st.kk[cur], st.spec_target)) return;  // those rounds wrote their signatures directly
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (size_t slot = first; slot < count; slot += stride) {
        const uint32_t e = st.list[cur][slot];
        const uint32_t item = e & kEntryItemMask;
        if (st.best[item] != (e >> kEntryShift)) continue;
        const uint32_t *src = st.w0 + slot * SB<MODE>::W0_SLOT_DW;
        uint8_t *dst = sig + (size_t)item * G::SIG;   // SIG is not a multiple of 4 for every parameter set: dwords, then the tail bytes
        for (int d = lane; d < G::SIG / 4; d += 64) {
            const uint32_t w = src[d];
            dst[4 * d] = (uint8_t)w; dst[4 * d + 1] = (uint8_t)(w >> 8); dst[4 * d + 2] = (uint8_t)(w >> 16); dst[4 * d + 3] = (uint8_t)(w >> 24);
        }
        for (int b = (G::SIG / 4) * 4 + lane; b < G::SIG; b += 64) dst[b] = reinterpret_cast<const uint8_t *>(src)[b];
    }
}
template <int MODE>
__global__ void __launch_bounds__(64) sign_commit_kernel(SignState st, int cur, uint8_t *__restrict__ sig) {
    sign_commit_body<MODE>(st, cur, sig, blockIdx.x, gridDim.x);
}

// lane = entry: the next active list.  Every item of the current one that is still unsigned has spent k attempts and gets
// k_next entries; k_next follows from the current list's item count (an upper bound on the survivors), the same for every
// lane.  One atomic per wavefront reserves the survivors' entries.  `last`: the list is for the persistent tail kernel, which
// wants one entry per item.
// (block / nblocks / nthreads: the launch's own for sign_compact_kernel; 0 / 1 / the workgroup's size when the last workgroup of a
// one-launch round does the compaction itself)
__device__ __forceinline__ void sign_compact_body(const SignState &st, int cur, int last, unsigned block, unsigned nblocks, unsigned nthreads) {
    const size_t count = st.count[cur];
    const unsigned k = st.kk[cur];
    const unsigned k_next = last ? 1u : sign_next_k((count + k - 1) / k, st.spec_target, st.pair);
    if (block == 0 && threadIdx.x == 0) st.kk[cur ^ 1] = k_next;
    if (last && block == 0 && threadIdx.x < 64) st.tail_work[threadIdx.x] = 0;  // (instead of a memset in front of the tail kernel: one launch less)
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (size_t base = (size_t)block * nthreads; base < count; base += (size_t)nblocks * nthreads) {
        const size_t a = base + threadIdx.x;
        uint32_t item = 0;
        bool survivor = false;
        if (a < count) {
            const uint32_t e = st.list[cur][a];
            item = e & kEntryItemMask;
            survivor = (e >> kEntryShift) == 0 && st.best[item] == kNoSuccess;  // the item's first entry speaks for it
        }
        const unsigned long long m = __ballot(survivor);
        if (m == 0) continue;  // wave-uniform
        uint32_t wbase = 0;
        if (lane == (int)(__ffsll((long long)m) - 1)) wbase = atomicAdd(&st.count[cur ^ 1], (uint32_t)__popcll(m) * k_next);
        wbase = (uint32_t)__builtin_amdgcn_readlane((int)wbase, __ffsll((long long)m) - 1);
        if (survivor) {
            st.attempts[item] += k;
            const uint32_t at = wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1)) * k_next;
            for (unsigned j = 0; j < k_next; j++) st.list[cur ^ 1][at + j] = item | (j << kEntryShift);
        }
    }
}
__global__ void __launch_bounds__(256) sign_compact_kernel(SignState st, int cur, int last) { sign_compact_body(st, cur, last, blockIdx.x, gridDim.x, 256); }

// two ranges zeroed by one launch (the end of a small signing call: one launch instead of two fills); 16-byte aligned, sizes multiples of 16
__global__ void __launch_bounds__(256) sign_wipe2_kernel(uint4 *__restrict__ a, size_t na, uint4 *__restrict__ b, size_t nb) {
    const uint4 z = {0u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += (size_t)gridDim.x * 256) {
        if (i < na) a[i] = z;
        else b[i - na] = z;
    }
}

// The fixed schedule: how many rounds the host enqueues for n items.  The survivors of a round are at most the items
// that entered it; with success probability p per attempt an item survives k attempts with (1 - p)^k.  The schedule follows
// the EXPECTED survivor count (with a pessimistic p and a safety margin on the count that drives k) until it is below
// 2^-40; the persistent tail kernel behind the schedule makes the result independent of that estimate.
template <int MODE>
inline int sign_round_schedule(size_t n, unsigned k0, unsigned spec_target, unsigned pair, size_t *entries_upper, bool *lazy, int max_rounds,
                               double eps = 9.1e-13) {
    // expected attempts per signature 4.25 / 5.1 / 3.85 (FIPS 204 table 1): success probability per attempt, times 0.85
    const double p = 0.85 * (DP<MODE>::K == 4 ? 0.235 : DP<MODE>::K == 6 ? 0.196 : 0.26);
    double items = (double)n;
    unsigned k = k0;
    int rounds = 0;
    while (items > eps && rounds < max_rounds) {
        // an upper estimate of the round's entries sizes its grids (one workgroup per entry: the hardware then balances the
        // unevenly long attempts; workgroups beyond the real count leave at once, and the kernels' grid-stride loops keep a
        // too-small estimate correct)
        entries_upper[rounds] = (size_t)(items * k * 1.03) + 64;
        lazy[rounds] = sign_round_lazy((unsigned long long)(items * k), k, spec_target);
        const double survive = __builtin_pow(1.0 - p, (double)k);
        // the device derives the next k from the items that ENTERED this round
        k = sign_next_k((unsigned long long)(items * 1.25 + 8.0), spec_target, pair);
        items *= survive;
        rounds++;
    }
    return rounds;
}

}  // namespace mldsa
}  // namespace circl
