// mlkem_kernels.h -- batch ML-KEM kernels for gfx950 (included by circl_hip.hip).
//
// Two launches per batch of encapsulations, both over the same HBM-resident arrays:
//
//  mlkem_hash_kernel<K>      lane = item.  H(ek) = SHA3-256 over the packed key (9 absorb
//                            blocks for ML-KEM-768), then (K,r) = SHA3-512(m || H(ek))
//                            (kem/mlkem/mlkem768/kyber.go:126-131, :258-260).  Writes the
//                            shared secret K and parks r (32 B / item) in the workspace.
//                            64 independent sponges per wavefront, states in registers.
//
//  mlkem_encrypt_kernel<K>   one wavefront per workgroup, G = 64 / K^2 items per workgroup:
//    phase A  lane = (item, i, j): SHAKE128(rho || i || j) rejection sampling of the K^2
//             matrix polynomials of G items at once (sample.go:192-236, mat.go:13-74 with
//             transpose=true), accepted coefficients streamed into LDS.  A^T never touches
//             HBM and lives only until phase C of the same workgroup consumed it.
//    phase B  lane = (item, nonce): SHAKE256(r || nonce) PRF blocks for the 2K+1 noise
//             polynomials (sample.go:31-95), raw bytes into LDS.
//    phase C  the wave walks its G items; per item it is K-PKE.Encrypt (cpapke.go:137-181)
//             with one polynomial per wavefront: CBD, 3 forward NTTs, K(K+1) lazy MulHat
//             accumulations, K+1 inverse NTTs, compress and bit-pack straight to HBM.
//             It also decodes t-hat from ek and applies UnpackMLKEM's canonical check
//             (cpapke.go:45-55): an item with a coefficient >= q gets status 1 and zeroed
//             outputs.
#pragma once
#include "kyber_dev.h"

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
    static constexpr int NOISE = 2 * K + 1;              // PRF streams per item
    static constexpr int A_STRIDE = 520;                 // bytes per sampled polynomial (+1 spill slot, 8-B aligned)
    static constexpr int NOISE_BYTES = 64 * P::ETA1;     // eta1 stream length (eta2 streams use 128)
    static constexpr int NOISE_STRIDE = NOISE_BYTES + 8; // breaks the power-of-two bank stride
    static constexpr int LDS_A = A_STREAMS * A_STRIDE;
    static constexpr int LDS_NOISE = G * NOISE * NOISE_STRIDE;
    static constexpr int LDS_XCH = 512;
    static constexpr int LDS_TOTAL = LDS_A + LDS_NOISE + LDS_XCH;
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

template <int FIRST, int COUNT> __device__ __forceinline__ void store_words(uint64_t *p, const KeccakState &s) {
    detail::static_for<0, COUNT>([&](auto ic) {
        constexpr int i = decltype(ic)::v;
        p[i] = ((uint64_t)s.hi[FIRST + i] << 32) | s.lo[FIRST + i];
    });
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

template <int K>
__device__ __forceinline__ void sample_matrix(uint8_t *lds_a, const uint8_t *__restrict__ ek, size_t item0, size_t n, int lane) {
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
    xor_words<0, 4>(s, reinterpret_cast<const uint64_t *>(ek + item * Gm::EK + 384 * K));
    s.lo[4] = (uint32_t)i | ((uint32_t)j << 8) | (kDsShake << 16);
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

// ---- phase B: PRF ---------------------------------------------------------------------------

// SHAKE256(r || nonce) -> NB bytes (128 for eta=2, 192 for eta=3) written to LDS as 64-bit words.
template <int K>
__device__ __forceinline__ void prf_streams(uint8_t *lds_noise, const uint8_t *__restrict__ r_ws, size_t item0, size_t n, int lane) {
    using Gm = Geom<K>;
    constexpr int STREAMS = Gm::G * Gm::NOISE;
#pragma unroll 1
    for (int base = 0; base < STREAMS; base += 64) {
        const int sidx = base + lane;
        const bool on = sidx < STREAMS;
        const int g = on ? sidx / Gm::NOISE : 0, nonce = on ? sidx % Gm::NOISE : 0;
        size_t item = item0 + g;
        if (item >= n) item = n - 1;
        KeccakState s;
        keccak_zero(s);
        xor_words<0, 4>(s, reinterpret_cast<const uint64_t *>(r_ws + item * 32));
        s.lo[4] = (uint32_t)nonce | (kDsShake << 8);
        s.hi[16] = 0x80000000u;
        keccak_f1600(s);
        uint32_t *out = reinterpret_cast<uint32_t *>(lds_noise + (on ? sidx : 0) * Gm::NOISE_STRIDE);
        if (on) {
            detail::static_for<0, 16>([&](auto ic) {
                constexpr int w = decltype(ic)::v;
                out[2 * w] = s.lo[w];
                out[2 * w + 1] = s.hi[w];
            });
        }
        if constexpr (Params<K>::ETA1 == 3) {
            // 192 bytes needed for eta1 = 3 streams (nonce < K): word 16 of this block, then 7 more
            if (on && nonce < K) { out[32] = s.lo[16]; out[33] = s.hi[16]; }
            if (__any(on && nonce < K)) {
                keccak_f1600(s);
                if (on && nonce < K) {
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

// ---- phase C helpers -------------------------------------------------------------------------

// CBD sample of coefficient n from a PRF byte string in LDS (sample.go:31-95)
template <int ETA> __device__ __forceinline__ int cbd_coeff(const uint8_t *buf, int n) {
    if constexpr (ETA == 2) {
        return kyber::cbd2_from_nibble((buf[n >> 1] >> (4 * (n & 1))) & 15u);
    } else {
        const int bit = 6 * n;
        const unsigned two = (unsigned)buf[bit >> 3] | ((unsigned)buf[(bit >> 3) + 1] << 8);
        return kyber::cbd3_from_6bits((two >> (bit & 7)) & 63u);
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

// ---- kernel 2: K-PKE.Encrypt ------------------------------------------------------------------

template <int K>
__global__ void __launch_bounds__(64) mlkem_encrypt_kernel(const uint8_t *__restrict__ ek, const uint8_t *__restrict__ m,
                                                          const uint8_t *__restrict__ r_ws, uint8_t *__restrict__ ct,
                                                          uint8_t *__restrict__ ss, uint8_t *__restrict__ status, size_t n) {
    using Gm = Geom<K>;
    using P = Params<K>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *lds_a = smem;
    uint8_t *lds_noise = smem + Gm::LDS_A;
    int16_t *xch = reinterpret_cast<int16_t *>(smem + Gm::LDS_A + Gm::LDS_NOISE);
    const int lane = threadIdx.x;
    const size_t item0 = (size_t)blockIdx.x * Gm::G;

    sample_matrix<K>(lds_a, ek, item0, n, lane);
    prf_streams<K>(lds_noise, r_ws, item0, n, lane);
    const kyber::LaneZetas z = kyber::load_lane_zetas(lane);
    __syncthreads();

#pragma unroll 1
    for (int g = 0; g < Gm::G; g++) {
        const size_t item = item0 + g;
        if (item >= n) break;  // wave-uniform
        const uint8_t *ekp = ek + item * Gm::EK;
        const uint8_t *noise = lds_noise + g * Gm::NOISE * Gm::NOISE_STRIDE;

        // t-hat (12-bit codec, poly.go:123-129) in layout L4 and UnpackMLKEM's range check
        int th[K][4];
        bool bad = false;
#pragma unroll
        for (int j = 0; j < K; j++) {
            const uint16_t *src = reinterpret_cast<const uint16_t *>(ekp + 384 * j + 6 * lane);
            const uint32_t h0 = src[0], h1 = src[1], h2 = src[2];
            th[j][0] = (int)(h0 & 0xfff);
            th[j][1] = (int)((h0 >> 12) | ((h1 & 0xff) << 4));
            th[j][2] = (int)((h1 >> 8) | ((h2 & 0xf) << 8));
            th[j][3] = (int)(h2 >> 4);
#pragma unroll
            for (int r = 0; r < 4; r++) bad |= th[j][r] >= Q;
        }
        const bool reject = __any(bad);

        // r-hat = NTT(CBD_eta1(PRF(r, j))), Barrett-reduced (cpapke.go:142-144), layout L4
        int rh[K][4];
#pragma unroll
        for (int j = 0; j < K; j++) {
#pragma unroll
            for (int r = 0; r < 4; r++) rh[j][r] = cbd_coeff<P::ETA1>(noise + j * Gm::NOISE_STRIDE, kyber::idx_l1(lane, r));
            kyber::ntt(rh[j], z, xch, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) rh[j][r] = kyber::barrett(rh[j][r]);
        }

        uint8_t *ctp = ct + item * Gm::CT;
        // u[i] = InvNTT(sum_j A^T[i][j] * r-hat[j]) + e1[i]  (cpapke.go:150-164), compressed to du bits
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            int acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < K; j++) {
                const int16_t *ap = reinterpret_cast<const int16_t *>(lds_a + ((g * K + i) * K + j) * Gm::A_STRIDE) + 4 * lane;
                const int a[4] = {ap[0], ap[1], ap[2], ap[3]};
                kyber::mulhat_acc(acc, a, rh[j], z.f6);
            }
            kyber::mulhat_finish(acc);
            kyber::invntt(acc, z, xch, lane);
            const uint8_t *e1 = noise + (K + i) * Gm::NOISE_STRIDE;
            uint16_t *cq = reinterpret_cast<uint16_t *>(xch);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nidx = kyber::idx_l1(lane, r);
                const int x = kyber::normalize(acc[r] + cbd_coeff<2>(e1, nidx));
                cq[nidx] = (uint16_t)kyber::compress_coeff<P::DU>(x);
            }
            __syncthreads();
            if (!reject) pack_bits_store<P::DU>(reinterpret_cast<uint32_t *>(ctp + 32 * P::DU * i), cq, lane);
            else {
                for (int w = lane; w < 8 * P::DU; w += 64) reinterpret_cast<uint32_t *>(ctp + 32 * P::DU * i)[w] = 0;
            }
        }
        // v = InvNTT(<t-hat, r-hat>) + e2 + Decompress_q(m, 1)  (cpapke.go:167-173), dv bits
        {
            int acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < K; j++) kyber::mulhat_acc(acc, th[j], rh[j], z.f6);
            kyber::mulhat_finish(acc);
            kyber::invntt(acc, z, xch, lane);
            const uint8_t *e2 = noise + 2 * K * Gm::NOISE_STRIDE;
            const uint8_t *mp = m + item * 32;
            uint16_t *cq = reinterpret_cast<uint16_t *>(xch);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nidx = kyber::idx_l1(lane, r);
                const int mbit = (mp[nidx >> 3] >> (nidx & 7)) & 1;
                const int x = kyber::normalize(acc[r] + cbd_coeff<2>(e2, nidx) + (-mbit & ((Q + 1) / 2)));
                cq[nidx] = (uint16_t)kyber::compress_coeff<P::DV>(x);
            }
            __syncthreads();
            uint32_t *dst = reinterpret_cast<uint32_t *>(ctp + 32 * P::DU * K);
            if (!reject) pack_bits_store<P::DV>(dst, cq, lane);
            else {
                for (int w = lane; w < 8 * P::DV; w += 64) dst[w] = 0;
            }
        }
        if (lane == 0) status[item] = reject ? 1 : 0;
        if (reject && lane < 8) reinterpret_cast<uint32_t *>(ss + item * 32)[lane] = 0;
    }
}

}  // namespace mlkem
}  // namespace circl
