// x25519_kernels.h -- batch X25519 (dh/x25519 KeyGen / Shared), one scalar multiplication per lane.
// A wavefront is 64 independent Montgomery ladders (Shared) or 64 fixed-base combs (KeyGen): no LDS, no cross-lane traffic,
// 64 bytes in and 33 bytes out per item against ~3.9 x 10^5 (ladder) / ~1.2 x 10^5 (comb) integer instructions -- the
// kernels are pure VALU issue (multiplier class).
#pragma once
#include <hip/hip_runtime.h>

#include "x25519_dev.h"

namespace circl {
namespace x25519 {

// scalar, point, out: n rows of 32 bytes (4-byte aligned); ok[n] = 1 unless the point is one of the low-order
// u-coordinates (key.go:24-31); BASE: point is ignored, the base point u = 9 is used (key.go:34-36).
// register-allocated for 4 waves per SIMD (126 VGPRs, no spills; 179 and 2 waves without the hint: -4 % throughput)
template <bool BASE>
static __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void x25519_kernel(const uint32_t *__restrict__ scalar, const uint32_t *__restrict__ point,
                                                           uint32_t *__restrict__ out, uint8_t *__restrict__ ok, size_t n) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8], u[8], r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        k[j] = scalar[i * 8 + j];
        u[j] = BASE ? 0u : point[i * 8 + j];
    }
    if (ok) {  // before the ladder: the point words are not kept alive across it
        uint32_t m[8];
#pragma unroll
        for (int j = 0; j < 8; j++) m[j] = u[j];
        m[7] &= 0x7fffffffu;
        ok[i] = BASE ? (uint8_t)1 : (uint8_t)valid_public(m);
    }
    if (BASE) base_mult(r, k);  // fixed-base comb (x25519_dev.h)
    else scalar_mult<false>(r, k, u);
#pragma unroll
    for (int j = 0; j < 8; j++) out[i * 8 + j] = r[j];
}

// Both ladders of a hybrid encapsulation / X-Wing decapsulation in ONE launch: workgroups [0, nb) compute the public keys
// X25519(scalar_i, 9) -> out_base, workgroups [nb, 2 nb) the shared secrets X25519(scalar_i, point_i) -> out_shared (+ ok).
// A ladder is ~0.9 ms of dependent instructions however few items there are, so a chunk of 2^15..2^16 items wants both
// ladders resident at once rather than one after the other.
static __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void x25519_pair_kernel(const uint32_t *__restrict__ scalar, const uint32_t *__restrict__ point,
                                                                uint32_t *__restrict__ out_base, uint32_t *__restrict__ out_shared,
                                                                uint8_t *__restrict__ ok, size_t n, unsigned nb) {
    const bool shared = blockIdx.x >= nb;  // wave-uniform
    const size_t i = (size_t)(blockIdx.x - (shared ? nb : 0)) * 64 + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8], u[8], r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        k[j] = scalar[i * 8 + j];
        u[j] = shared ? point[i * 8 + j] : 0u;
    }
    if (shared) {
        if (ok) {
            uint32_t m[8];
#pragma unroll
            for (int j = 0; j < 8; j++) m[j] = u[j];
            m[7] &= 0x7fffffffu;
            ok[i] = (uint8_t)valid_public(m);
        }
        scalar_mult<false>(r, k, u);
    } else {
        base_mult(r, k);
    }
    uint32_t *out = shared ? out_shared : out_base;
#pragma unroll
    for (int j = 0; j < 8; j++) out[i * 8 + j] = r[j];
}

}  // namespace x25519
}  // namespace circl
