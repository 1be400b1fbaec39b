// prim_kernels.h -- batch kernels for the unit-level building blocks the reference tests one by
// one (KeccakF1600, Poly.NTT / InvNTT / MulHat, SHAKE/SHA3 sponges).  They exist so the parity
// tests can compare every device function with the oracle in isolation; the fused ML-KEM kernels
// use the same device functions.
#pragma once
#include "kyber_dev.h"
#include "dilithium_dev.h"
#include "lane_ops.h"

namespace circl {
namespace prim {

// n states of 25 little-endian uint64 words (row-major), one state per lane.
__global__ void __launch_bounds__(256) keccak_f1600_kernel(uint64_t *states, size_t n, int first_round) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t *p = states + i * 25;
    KeccakState s;
#pragma unroll
    for (int w = 0; w < 25; w++) { s.lo[w] = (uint32_t)p[w]; s.hi[w] = (uint32_t)(p[w] >> 32); }
    keccak_f1600(s, first_round);
#pragma unroll
    for (int w = 0; w < 25; w++) p[w] = ((uint64_t)s.hi[w] << 32) | s.lo[w];
}

// ---- issue-rate probes (circl_hip_profile_valu_probe): what the chip's VALUs sustain, measured where the kernels run ---------------
// (a) the product's own Keccak-f[1600] round (keccak_dev.h: 180 VALU instructions, V_BITOP3 / V_ALIGNBIT) `iters` times on a state that
// stays in registers -- no memory traffic, so the time is pure issue; (b) the cheapest VALU instruction there is, a two-operand
// integer op on VGPRs, 8 independent chains.  Launched as 256-thread workgroups, `waves_per_simd` of them per CU.
__global__ void __launch_bounds__(256) keccak_rate_probe_kernel(uint32_t *sink, int iters) {
    KeccakState s;
#pragma unroll
    for (int w = 0; w < 25; w++) { s.lo[w] = threadIdx.x * 2654435761u + w; s.hi[w] = blockIdx.x * 40503u ^ (w << 7); }
#pragma unroll 1
    for (int i = 0; i < iters; i++) keccak_f1600(s, 0);
    uint32_t x = 0;
#pragma unroll
    for (int w = 0; w < 25; w++) x ^= s.lo[w] ^ s.hi[w];
    if (x == 0x12345678u) sink[0] = x;  // (never true in practice: keeps the chain alive)
}
constexpr int kSimpleProbePerIter = 256;  // VALU instructions per loop iteration of simple_rate_probe_kernel
__global__ void __launch_bounds__(256) simple_rate_probe_kernel(uint32_t *sink, int iters) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const uint32_t b = threadIdx.x * 2654435761u + blockIdx.x;
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < kSimpleProbePerIter / 8; r++)
            asm volatile("v_add_u32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_xor_b32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(b));
    }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678u) sink[0] = a0;
}

// The wave-cooperative form of the permutation (keccak_f1600_coop: one state per wavefront, lanes 0..24 own a lane each),
// which the ML-DSA kernels use on a rare serial path: one state per single-wave workgroup.
__global__ void __launch_bounds__(64) keccak_f1600_coop_kernel(uint64_t *states) {
    __shared__ uint64_t ws[55];
    const int lane = threadIdx.x;
    uint64_t *p = states + (size_t)blockIdx.x * 25;
    if (lane < 25) ws[lane] = p[lane];
    keccak_f1600_coop(ws, lane);
    if (lane < 25) p[lane] = ws[lane];
}

// The two-lanes-per-state form (keccak_f1600_split): lane pair t of the grid owns state t.
__global__ void __launch_bounds__(256) keccak_f1600_split_kernel(uint64_t *states, size_t n) {
    const size_t lane = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t item = lane >> 1;
    const int p = (int)(lane & 1);
    const bool live = item < n;
    if (!live) item = n - 1;  // both lanes of a pair stay in step; the duplicate is not stored
    uint32_t *w = reinterpret_cast<uint32_t *>(states + item * 25);
    SplitState s;
#pragma unroll
    for (int k = 0; k < 25; k++) s.w[k] = w[2 * k + p];
    keccak_f1600_split(s, p != 0);
    if (live) {
#pragma unroll
        for (int k = 0; k < 25; k++) w[2 * k + p] = s.w[k];
    }
}

// One polynomial per single-wave workgroup, int16[256] in standard order, in place.
// Outputs are normalised to [0,q).  The inverse carries the reference's factor: Poly.InvNTT returns 2^16 times the
// exact inverse (ntt.go:145-193; ntt_test.go:83-109 checks InvNTT(NTT(p)) = p * 2^16).
__global__ void __launch_bounds__(64) kyber_ntt_kernel(int16_t *polys, int inverse) {
    __shared__ __attribute__((aligned(16))) uint32_t xch[256];
    const int lane = threadIdx.x;
    int16_t *p = polys + (size_t)blockIdx.x * 256;
    const kyber::LaneZetas z = kyber::load_lane_zetas(lane);
    int c[4];
    if (!inverse) {
#pragma unroll
        for (int r = 0; r < 4; r++) c[r] = kyber::normalize(p[kyber::idx_l1(lane, r)]);
        kyber::ntt(c, z, xch, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) p[kyber::idx_l4(lane, r)] = (int16_t)kyber::normalize(c[r]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) c[r] = kyber::normalize(p[kyber::idx_l4(lane, r)]);
        kyber::invntt<65536u>(c, z, xch, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) p[kyber::idx_l1(lane, r)] = (int16_t)c[r];
    }
}

__global__ void __launch_bounds__(64) kyber_mulhat_kernel(int16_t *out, const int16_t *a, const int16_t *b) {
    const int lane = threadIdx.x;
    const size_t off = (size_t)blockIdx.x * 256 + 4 * lane;
    int x[4], y[4], acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; r++) { x[r] = kyber::normalize(a[off + r]); y[r] = kyber::normalize(b[off + r]); }
    kyber::mulhat_acc_packed(acc, kyber::pack16(x[0], x[1]), kyber::pack16(x[2], x[3]), kyber::hat_prepare(y, kyber::zeta_c(64 + lane), kyber::zeta_cn(64 + lane)));
    kyber::mulhat_finish(acc);  // -2^-32 times the products; the reference's MulHat carries 2^-16 (poly.go:63-100)
    constexpr uint32_t fix = kyber::mulc_const(kyber::modq((int64_t)kyber::NEG_R32 * kyber::invq(65536u % kyber::Q)));
#pragma unroll
    for (int r = 0; r < 4; r++) out[off + r] = (int16_t)kyber::mulc((uint32_t)acc[r], fix);
}

// sign/internal/dilithium Poly.NTT / Poly.InvNTT, one polynomial per single-wave workgroup,
// uint32[256] in standard order, in place, outputs normalised to [0,q).  The reference's InvNTT
// returns 2^32/256 times the exact inverse (ntt.go:212-216); the device transform is exact, so the
// factor 2^32 mod q is applied here to expose the reference's semantics.
__global__ void __launch_bounds__(64) dilithium_ntt_kernel(uint32_t *polys, int inverse) {
    __shared__ __attribute__((aligned(16))) uint32_t xch[dilithium::kXchWords];
    const int lane = threadIdx.x;
    uint32_t *p = polys + (size_t)blockIdx.x * 256;
    const dilithium::LaneZetas z = dilithium::load_lane_zetas(lane);
    uint32_t c[4];
    if (!inverse) {
#pragma unroll
        for (int r = 0; r < 4; r++) c[r] = dilithium::normalize(p[kyber::idx_l1(lane, r)]);
        dilithium::ntt(c, z, xch, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) p[kyber::idx_l4(lane, r)] = dilithium::normalize(c[r]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) c[r] = dilithium::normalize(p[kyber::idx_l4(lane, r)]);
        dilithium::invntt(c, z, xch, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) p[kyber::idx_l1(lane, r)] = dilithium::normalize(dilithium::mont32(c[r], dilithium::R32SQ));  // c * 2^32
    }
}

// n independent sponges over equal-length byte strings; one stream per lane
// (internal/sha3 State.Write / Read).  rate_words in {9, 17, 21}.
// in_off == nullptr: equal-length messages of `inlen` bytes; otherwise message i is in[in_off[i] .. in_off[i+1]).
// first_round = 0 for Keccak-f[1600], 12 for the 12-round TurboSHAKE permutation (shake.go:60-90).
// Message i is in[in_off[i] .. in_off[i + 1]) (ragged, n + 1 offsets), or in[in_off[i] .. in_off[i] + in_len[i]) when in_len is
// given (ranges of one resident buffer: the KangarooTwelve leaves, which lie inside their messages), or in[i * inlen_eq ..)
// when there are no offsets (equal lengths).
__global__ void __launch_bounds__(256) sponge_kernel(int rate_words, uint32_t ds, int first_round, const uint8_t *in, size_t inlen_eq,
                                                     const uint64_t *in_off, uint8_t *out, size_t outlen, size_t n,
                                                     const uint64_t *in_len = nullptr, uint64_t suffix = 0) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t *p = in_off ? in + in_off[i] : in + i * inlen_eq;
    // in_len[i]: bits 0..55 the range's length, bits 56..63 how many bytes of `suffix` (little-endian) follow it in the
    // message (KangarooTwelve's length_encode(|C|) after a message that is hashed where it lies)
    const size_t body = in_len ? (size_t)(in_len[i] & 0x00ffffffffffffffull) : in_off ? (size_t)(in_off[i + 1] - in_off[i]) : inlen_eq;
    const size_t inlen = body + (in_len ? (size_t)(in_len[i] >> 56) : 0);
    uint8_t *o = out + i * outlen;
    const size_t rate = (size_t)rate_words * 8;
    KeccakState s;
    keccak_zero(s);
    size_t pos = 0;
    bool padded = false;
    // Words that lie wholly inside the message come from aligned 64-bit loads (two of them and a funnel shift when the
    // message starts at an odd address: the second one reaches at most 7 bytes past the message, inside the 16 bytes of
    // slack every staged blob carries); only the words around the message's end are assembled byte by byte.  (The first
    // version assembled every word from 8 checked byte loads: ~1.5 k instructions per block next to the 2.2 k of a
    // 12-round permutation.)
    const unsigned mis = (unsigned)(reinterpret_cast<uintptr_t>(p) & 7);
    const uint64_t *pa = reinterpret_cast<const uint64_t *>(p - mis);
    while (!padded) {
        // xor one block: message bytes, then the ds byte, then 0x80 at the end of the final block
#pragma unroll
        for (int w = 0; w < 21; w++) {
            if (w < rate_words) {
                const size_t k0 = pos + 8 * (size_t)w;
                uint64_t v = 0;
                if (k0 + 8 <= body) {
                    const uint64_t lo = pa[k0 >> 3];
                    v = lo;
                    if (mis) v = (lo >> (8 * mis)) | (pa[(k0 >> 3) + 1] << (64 - 8 * mis));
                } else {
                    for (int b = 0; b < 8; b++) {
                        const size_t k = k0 + b;
                        uint64_t byte = 0;
                        if (k < body) byte = p[k];
                        else if (k < inlen) byte = (suffix >> (8 * (k - body))) & 0xff;
                        else if (k == inlen) byte = ds;
                        v |= byte << (8 * b);
                    }
                }
                s.lo[w] ^= (uint32_t)v;
                s.hi[w] ^= (uint32_t)(v >> 32);
            }
        }
        if (pos + rate > inlen) {  // this block holds the ds byte: it is the last one
            padded = true;
#pragma unroll
            for (int w = 0; w < 21; w++)
                if (w == rate_words - 1) s.hi[w] ^= 0x80000000u;
        }
        keccak_f1600(s, first_round);
        pos += rate;
    }
    size_t done = 0;
    while (done < outlen) {
#pragma unroll
        for (int w = 0; w < 21; w++) {
            if (w < rate_words) {
                const uint64_t v = ((uint64_t)s.hi[w] << 32) | s.lo[w];
                for (int b = 0; b < 8; b++) {
                    const size_t k = done + 8 * (size_t)w + b;
                    if (k < outlen) o[k] = (uint8_t)(v >> (8 * b));
                }
            }
        }
        done += rate;
        if (done < outlen) keccak_f1600(s, first_round);
    }
}


// ---- lane-local arithmetic, one element per lane -------------------------------------------------------------------------
// The coefficient-level functions of kyber_dev.h / dilithium_dev.h applied elementwise by the DEVICE instantiation (which
// uses __mul24, __umulhi, V_BITOP3, V_ALIGNBIT, V_DOT2 where the host instantiation of the same source uses plain C), so
// that the parity tests can sweep their whole domains on the GPU: compress / decompress = exact rounding for every
// representative (pke/kyber/internal/common/poly_test.go:351-378), decompose / useHint / makeHint / power2round for every
// a < q and both gamma2 (sign/mldsa/mldsa65/internal/rounding_test.go:14-67), the Montgomery products at their bounds.
// op codes: include/circl_hip.h (CIRCL_HIP_LANE_*).
__global__ void __launch_bounds__(256) lane_op_kernel(int op, int arg, const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                                                      uint32_t *__restrict__ out0, uint32_t *__restrict__ out1, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t r0, r1;
    lane_op_eval(op, arg, a[i], b ? b[i] : 0u, r0, r1);
    out0[i] = r0;
    if (out1) out1[i] = r1;
}

}  // namespace prim
}  // namespace circl
