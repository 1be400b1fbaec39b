// sampler_prims.h -- the samplers of the fused kernels as unit-level primitives (included by api_prims.hip only).
//
// The fused ML-KEM / ML-DSA kernels never expose their samplers: the matrix streams only ever run with the coordinates of
// a K x K / K x L matrix and the PRF streams with nonces 0 .. 2K.  The reference pins its samplers on fixed vectors with
// other arguments (pke/kyber/internal/common/sample_test.go:23-138: DeriveNoise2 / DeriveNoise3 with nonce 37,
// DeriveUniform(seed, 1, 0); sign/mldsa/mldsa65/internal/sample_test.go:12-63: PolyDeriveUniform with nonce 30000 and
// nonces 0..99).  These kernels run the SAME device functions -- the branch-free LDS-FIFO rejection samplers
// (sample_matrix_scratch, expand_a_scratch) and the PRF pass + CBD decode (prf_streams, cbd_coeff) -- with caller-chosen
// arguments, one stream per lane, so that the GPU tests can check those vectors and sweep against the oracle.
#pragma once
#include "mlkem_kernels.h"
#include "mldsa_kernels.h"

namespace circl {
namespace prim {

// Poly.DeriveUniform(seed_i, x_i, y_i) (sample.go:192-236) for 64 items per single-wave workgroup -> int16[256] rows in
// coefficient order, values in [0, q)
__global__ void __launch_bounds__(64) kyber_uniform_prim_kernel(const uint8_t *__restrict__ seeds, const uint8_t *__restrict__ xy,
                                                                int16_t *__restrict__ out, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const size_t item0 = (size_t)blockIdx.x * 64;
    mlkem::sample_matrix_scratch<3, true, 64, true>(smem, out + item0 * 256, seeds, (size_t)32, item0, n, threadIdx.x, xy);
}

// Poly.DeriveNoise(seed_i, nonce, eta) for ALL nonces 0..63 (sample.go:17-95): one seed per single-wave workgroup, lane =
// nonce in the PRF pass, then the ring-phase decode -> int16[64][256], centred values in [-eta, eta].
// K selects the parameter set whose eta1 is wanted: K = 3 -> eta 2 (packed-nibble form), K = 2 -> eta 3.
template <int K>
__global__ void __launch_bounds__(64) kyber_cbd_prim_kernel(const uint8_t *__restrict__ seeds, int16_t *__restrict__ out, size_t n) {
    using Gm = mlkem::Geom<K>;
    constexpr int ETA = mlkem::Params<K>::ETA1;
    __shared__ __attribute__((aligned(16))) uint8_t noise[64 * Gm::NOISE_STRIDE];
    const int lane = threadIdx.x;
    const size_t item = blockIdx.x;
    mlkem::prf_streams<K, 64, 64, 1>(noise, seeds, (size_t)32, item, n, lane);
    __syncthreads();
#pragma unroll 1
    for (int nonce = 0; nonce < 64; nonce++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int idx = kyber::idx_l1(lane, r);
            out[(item * 64 + nonce) * 256 + idx] = (int16_t)(mlkem::cbd_coeff<ETA>(noise + nonce * Gm::NOISE_STRIDE, idx) - kyber::Q);
        }
}

// PolyDeriveUniform(seed_i, nonce_i) (sign/mldsa/mldsa65/internal/sample.go:92-123) for 64 items per single-wave
// workgroup: the 24-bit packed row the sampler produces, then unpacked by the lane that wrote it -> uint32[256] rows
__global__ void __launch_bounds__(64) mldsa_uniform_prim_kernel(const uint8_t *__restrict__ seeds, const uint16_t *__restrict__ nonces,
                                                                uint32_t *__restrict__ packed_rows, uint32_t *__restrict__ out, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x;
    const size_t item0 = (size_t)blockIdx.x * 64;
    uint32_t *rows = packed_rows + item0 * mldsa::kPackedRowDwords;
    mldsa::expand_a_scratch<44, false, 64, true>(smem, rows, seeds, (size_t)32, item0, n, lane, nonces);
    __threadfence_block();  // the row below was written by this very lane
    if (item0 + lane >= n) return;
    const uint32_t *row = rows + lane * mldsa::kPackedRowDwords;  // written by this very lane
    uint32_t *o = out + (item0 + lane) * 256;
#pragma unroll 1
    for (int g = 0; g < 64; g++) {
        const uint32_t w0 = row[3 * g], w1 = row[3 * g + 1], w2 = row[3 * g + 2];
        o[4 * g] = w0 & 0xffffffu;
        o[4 * g + 1] = (w0 >> 24) | ((w1 & 0xffffu) << 8);
        o[4 * g + 2] = (w1 >> 16) | ((w2 & 0xffu) << 16);
        o[4 * g + 3] = w2 >> 8;
    }
}

}  // namespace prim
}  // namespace circl
