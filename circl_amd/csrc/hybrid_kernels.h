// hybrid_kernels.h -- the glue of the hybrid KEMs X-Wing (kem/xwing/xwing.go) and X25519MLKEM768 (kem/hybrid/hybrid.go,
// xkem.go) around the ML-KEM-768 and X25519 batch kernels: seed expansion, the X-Wing combiner, and the strided copies that
// split pk = ek || pk_X, ct = ct_M || ct_X, sk = dk || sk_X into the contiguous arrays the two halves work on.
// Lane = item for the hashes (one or two Keccak permutations each), thread = dword for the copies.
#pragma once
#include <hip/hip_runtime.h>

#include "keccak_dev.h"

namespace circl {
namespace hybridk {

__device__ __forceinline__ void state_zero(KeccakState &s) {
#pragma unroll
    for (int i = 0; i < 25; i++) s.lo[i] = s.hi[i] = 0;
}
template <int FIRST, int COUNT> __device__ __forceinline__ void state_load(KeccakState &s, const uint32_t *p) {
#pragma unroll
    for (int i = 0; i < COUNT; i++) {
        s.lo[FIRST + i] = p[2 * i];
        s.hi[FIRST + i] = p[2 * i + 1];
    }
}
template <int FIRST, int COUNT> __device__ __forceinline__ void state_store(uint32_t *p, const KeccakState &s) {
#pragma unroll
    for (int i = 0; i < COUNT; i++) {
        p[2 * i] = s.lo[FIRST + i];
        p[2 * i + 1] = s.hi[FIRST + i];
    }
}
// SHAKE256 padding for a message of WORDS 64-bit words (WORDS < 16): 0x1f after the message, 0x80 into byte 135
template <int WORDS> __device__ __forceinline__ void shake256_pad(KeccakState &s) {
    s.lo[WORDS] ^= 0x1fu;
    s.hi[16] ^= 0x80000000u;
}

// dst[r][0 .. width) = src[r][0 .. width) for rows of dwords with independent strides
static __global__ __launch_bounds__(256) void rows_copy_kernel(uint32_t *__restrict__ dst, size_t dst_stride, const uint32_t *__restrict__ src,
                                                               size_t src_stride, unsigned width, size_t n) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t r = t / width;
    const unsigned c = (unsigned)(t - r * width);
    if (r < n) dst[r * dst_stride + c] = src[r * src_stride + c];
}
// dst[r] = table[key_idx ? key_idx[r] : 0]: rows of `width` dwords gathered from a key table (a resident hybrid key table's X25519 rows)
static __global__ __launch_bounds__(256) void rows_gather_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ table,
                                                                 const KeyIdx key_idx, unsigned width, size_t n) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t r = t / width;
    const unsigned c = (unsigned)(t - r * width);
    if (r < n) dst[r * width + c] = table[(key_idx ? (size_t)key_idx[r] : size_t(0)) * width + c];
}
// rows whose status byte is non-zero are zero-filled (the reference returns nil, nil, err: hybrid.go:283-300)
static __global__ __launch_bounds__(256) void rows_zero_failed_kernel(uint32_t *__restrict__ dst, size_t dst_stride, unsigned width,
                                                                      const uint8_t *__restrict__ status, size_t n) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t r = t / width;
    const unsigned c = (unsigned)(t - r * width);
    if (r < n && status[r]) dst[r * dst_stride + c] = 0;
}

// X-Wing key expansion (xwing.go:119-124): SHAKE256(seed[32]) -> seedm[64] || sk_X[32]
static __global__ __launch_bounds__(256) void xwing_expand_kernel(const uint32_t *__restrict__ seed, uint32_t *__restrict__ seedm,
                                                                  uint32_t *__restrict__ skx, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    if (!live) i = n - 1;
    KeccakState s;
    state_zero(s);
    state_load<0, 4>(s, seed + i * 8);
    shake256_pad<4>(s);
    keccak_f1600(s);
    if (live) {
        state_store<0, 8>(seedm + i * 16, s);
        state_store<8, 4>(skx + i * 8, s);
    }
}

// kem/hybrid seed expansion (hybrid.go:236-250, :271-300; the X25519 half is a KEM of its own: xkem.go:112-123, :160-178).
// ex = SHAKE256(seed) is cut into the first component's seed, then the second's; the X25519 component turns its 32 bytes into
// the scalar sk_X = SHAKE256(xseed)[:32].  IN = seed words (8 for DeriveKeyPair, 4 for EncapsulateDeterministically), KEMW =
// words of the lattice KEM's seed (8: d || z; 4: m), XFIRST = X25519 is the first component (Kyber768-X25519) or the second
// (X25519MLKEM768).
template <int IN, int KEMW, bool XFIRST>
static __global__ __launch_bounds__(256) void hybrid_expand_kernel(const uint32_t *__restrict__ seed, uint32_t *__restrict__ kem_seed,
                                                                   uint32_t *__restrict__ skx, size_t n) {
    constexpr int KEM_AT = XFIRST ? 4 : 0, X_AT = XFIRST ? 0 : KEMW;
    static_assert(KEMW + 4 <= 17, "one squeeze block");
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    if (!live) i = n - 1;
    KeccakState s;
    state_zero(s);
    state_load<0, IN>(s, seed + i * 2 * IN);
    shake256_pad<IN>(s);
    keccak_f1600(s);
    if (live) state_store<KEM_AT, KEMW>(kem_seed + i * 2 * KEMW, s);
    KeccakState x;
    state_zero(x);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        x.lo[j] = s.lo[X_AT + j];
        x.hi[j] = s.hi[X_AT + j];
    }
    shake256_pad<4>(x);
    keccak_f1600(x);
    if (live) state_store<0, 4>(skx + i * 8, x);
}

// X-Wing combiner (xwing.go:53-71): SHA3-256(ss_M || ss_X || ct_X || pk_X || "\.//^\") -- 134 bytes, one permutation.
// Items whose ML-KEM half failed the encapsulation-key check (status != 0) get a zero shared secret.
static __global__ __launch_bounds__(256) void xwing_combine_kernel(const uint32_t *__restrict__ ssm, const uint32_t *__restrict__ ssx,
                                                                   const uint32_t *__restrict__ ctx, const uint32_t *__restrict__ pkx,
                                                                   const uint8_t *__restrict__ status, uint32_t *__restrict__ ss, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    if (!live) i = n - 1;
    KeccakState s;
    state_zero(s);
    state_load<0, 4>(s, ssm + i * 8);
    state_load<4, 4>(s, ssx + i * 8);
    state_load<8, 4>(s, ctx + i * 8);
    state_load<12, 4>(s, pkx + i * 8);
    s.lo[16] = 0x2f2f2e5cu;                // "\.//"
    s.hi[16] = 0x80060000u | 0x00005c5eu;  // "^\" , SHA3 domain byte 0x06 at byte 134, 0x80 at byte 135
    keccak_f1600(s);
    if (live) {
        const bool bad = status && status[i];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            ss[i * 8 + 2 * j] = bad ? 0u : s.lo[j];
            ss[i * 8 + 2 * j + 1] = bad ? 0u : s.hi[j];
        }
    }
}

// X25519MLKEM768 status: encapsulation -> kem.ErrPubKey (1) if the ML-KEM key check failed or the X25519 public key is a
// low-order point (xkem.go:144-146); decapsulation -> kem.ErrPrivKey (2) for a private key failing its hash check, else
// ErrPubKey (1) for a low-order X25519 ciphertext.
static __global__ __launch_bounds__(256) void hybrid_status_kernel(uint8_t *__restrict__ status, const uint8_t *__restrict__ ok_x, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) status[i] = status[i] ? status[i] : (ok_x[i] ? 0 : 1);
}

}  // namespace hybridk
}  // namespace circl
