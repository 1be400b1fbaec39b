// x25519_kernels.h -- batch X25519 (dh/x25519 KeyGen / Shared), one scalar multiplication per lane.
// A wavefront is 64 independent Montgomery ladders: no LDS, no cross-lane traffic, 64 bytes in and 33 bytes out per item
// against ~3.7 x 10^5 integer instructions -- the kernel is pure VALU issue (multiplier class).
#pragma once
#include <hip/hip_runtime.h>

#include "x25519_dev.h"

namespace circl {
namespace x25519 {

// scalar, point, out: n rows of 32 bytes (4-byte aligned); ok[n] = 1 unless the point is one of the low-order
// u-coordinates (key.go:24-31); BASE: point is ignored, the base point u = 9 is used (key.go:34-36).
template <bool BASE>
static __global__ __launch_bounds__(64) void x25519_kernel(const uint32_t *__restrict__ scalar, const uint32_t *__restrict__ point,
                                                           uint32_t *__restrict__ out, uint8_t *__restrict__ ok, size_t n) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8], u[8], r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        k[j] = scalar[i * 8 + j];
        u[j] = BASE ? 0u : point[i * 8 + j];
    }
    scalar_mult<BASE>(r, k, u);
#pragma unroll
    for (int j = 0; j < 8; j++) out[i * 8 + j] = r[j];
    if (ok) {
        u[7] &= 0x7fffffffu;
        ok[i] = BASE ? (uint8_t)1 : (uint8_t)valid_public(u);
    }
}

}  // namespace x25519
}  // namespace circl
