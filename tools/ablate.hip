// tools/ablate.hip -- profiling aid, not part of the product.
//   1. VALU issue-rate probe: Keccak-f[1600] back to back (no memory) at 1/2/4/8 waves per SIMD.
//   2. Phase ablation of mlkem_encrypt_kernel<3>: time with phase A / B / C removed.
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I circl_amd/csrc tools/ablate.hip -o build/ablate && build/ablate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mlkem_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

using namespace circl;

__global__ void __launch_bounds__(64) keccak_loop(uint64_t *out, int perms) {
    extern __shared__ uint8_t pad[];
    KeccakState s;
    keccak_zero(s);
    s.lo[0] = threadIdx.x + blockIdx.x * 64;
    for (int i = 0; i < perms; i++) keccak_f1600(s);
    if (s.lo[3] == 0x12345678u && pad[threadIdx.x] == 77) out[0] = s.lo[0];  // keep the work alive
}

// NTT-only probe: forward + inverse transforms in a loop, one wave per block
__global__ void __launch_bounds__(64) ntt_loop(int16_t *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t sm[];
    int16_t *xch = reinterpret_cast<int16_t *>(sm);
    const int lane = threadIdx.x;
    const kyber::LaneZetas z = kyber::load_lane_zetas(lane);
    int c[4] = {lane, lane + 1, lane + 2, lane + 3};
    for (int i = 0; i < iters; i++) {
        kyber::ntt(c, z, xch, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) c[r] = kyber::normalize(c[r]);
        kyber::invntt<1u>(c, z, xch, lane);
    }
    if (c[0] == 12345) out[0] = (int16_t)c[1];
}

static float time_ms(void (*launch)(void *), void *arg, int reps = 3) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(arg);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) launch(arg);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    uint64_t *d_out; CK(hipMalloc(&d_out, 4096));
    printf("== VALU probe: Keccak-f[1600], 4320 VALU/perm, 64 states per wave ==\n");
    for (int wps : {1, 2, 4, 8}) {
        // LDS per block caps residency: 160 KB / (4*wps) blocks per CU
        const int lds = 160 * 1024 / (4 * wps) - 512;
        const int blocks = 256 * 4 * wps * 4, perms = 200;
        CK(hipFuncSetAttribute((const void *)keccak_loop, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        struct A { uint64_t *o; int b, l, p; } a{d_out, blocks, lds, perms};
        float ms = time_ms([](void *v) { A *a = (A *)v; hipLaunchKernelGGL(keccak_loop, dim3(a->b), dim3(64), a->l, 0, a->o, a->p); }, &a);
        double perm_s = (double)blocks * 64 * perms / (ms * 1e-3);
        printf("  %d wave/SIMD: %.3f ms  %.3e perm/s  %.2f cycles/wave-instr @2.4GHz (per SIMD)\n", wps, ms, perm_s,
               2.4e9 * 1024 / (perm_s / 64 * 4320));
    }
    printf("== NTT probe (fwd+inv per iteration), one wave per block ==\n");
    for (int wps : {1, 2, 4, 8}) {
        const int lds = 160 * 1024 / (4 * wps) - 512;
        const int blocks = 256 * 4 * wps * 4, iters = 500;
        CK(hipFuncSetAttribute((const void *)ntt_loop, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        struct A { int16_t *o; int b, l, p; } a{(int16_t *)d_out, blocks, lds, iters};
        float ms = time_ms([](void *v) { A *a = (A *)v; hipLaunchKernelGGL(ntt_loop, dim3(a->b), dim3(64), a->l, 0, a->o, a->p); }, &a);
        printf("  %d wave/SIMD: %.3f ms  %.3e (ntt+invntt)/s  %.0f cycles per pair per wave-slot\n", wps, ms, (double)blocks * iters / (ms * 1e-3),
               2.4e9 * 1024 * wps / ((double)blocks * iters / (ms * 1e-3)));
    }

    printf("== ablation of mlkem_encrypt_kernel<3>, n = 2^18 ==\n");
    constexpr int K = 3;
    using Gm = mlkem::Geom<K>;
    const size_t n = 1 << 18;
    uint8_t *seed, *ek, *dk, *m, *ct, *ss, *st, *ws;
    CK(hipMalloc(&seed, 64 * n)); CK(hipMalloc(&ek, Gm::EK * n)); CK(hipMalloc(&dk, Gm::DK * n)); CK(hipMalloc(&m, 32 * n));
    CK(hipMalloc(&ct, Gm::CT * n)); CK(hipMalloc(&ss, 32 * n)); CK(hipMalloc(&st, n)); CK(hipMalloc(&ws, 128 * n));
    std::vector<uint8_t> h(64 * n);
    srand(1);
    for (auto &x : h) x = (uint8_t)rand();
    CK(hipMemcpy(seed, h.data(), 64 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(m, h.data(), 32 * n, hipMemcpyHostToDevice));
    const unsigned hb = (unsigned)((n + 255) / 256), eb = (unsigned)((n + Gm::G - 1) / Gm::G);
    hipLaunchKernelGGL(mlkem::mlkem_keygen_seed_kernel<K>, dim3(hb), dim3(256), 0, 0, seed, ws, n);
    uint8_t *scratch; CK(hipMalloc(&scratch, (size_t)256 * 32 * Gm::SCRATCH_BYTES));
    hipLaunchKernelGGL((mlkem::mlkem_keygen_kernel<K, false>), dim3(eb), dim3(64), Gm::LDS_TOTAL, 0, (const uint8_t *)ws, ek, dk, scratch, (unsigned *)nullptr, n);
    hipLaunchKernelGGL(mlkem::mlkem_hash_kernel<K>, dim3(hb), dim3(256), 0, 0, ek, m, ss, ws, n);
    CK(hipDeviceSynchronize());
    unsigned *work; CK(hipMalloc(&work, 256));
    struct E { const uint8_t *ek, *m, *r; uint8_t *ct, *ss, *st, *scratch; unsigned *work; size_t n; unsigned eb; } e{ek, m, ws, ct, ss, st, scratch, work, n, eb};
#define RUN(MASK, NAME)                                                                                                         \
    {                                                                                                                           \
        float ms = time_ms([](void *v) { E *e = (E *)v;                                                                         \
            hipLaunchKernelGGL((mlkem::mlkem_encrypt_kernel<K, mlkem::ENCAPS, MASK, false>), dim3(e->eb), dim3(64), Gm::LDS_TOTAL, 0, e->ek, \
                               (size_t)Gm::EK, e->m, e->r, e->ct, e->ss, e->st, (const uint8_t *)nullptr, (const uint8_t *)nullptr, e->scratch, (unsigned *)nullptr, e->n, (const uint32_t *)nullptr, (const int16_t *)nullptr); }, &e); \
        printf("  %-28s %.3f ms\n", NAME, ms);                                                                                  \
    }
#define RUNS(MASK, BPC, NAME)                                                                                                   \
    {                                                                                                                           \
        e.eb = 256 * BPC;                                                                                                       \
        float ms = time_ms([](void *v) { E *e = (E *)v;                                                                         \
            hipMemsetAsync(e->work, 0, 4, 0);                                                                                   \
            hipLaunchKernelGGL((mlkem::mlkem_encrypt_kernel<K, mlkem::ENCAPS, MASK, true>), dim3(e->eb), dim3(64), Gm::LDS_SCRATCH_TOTAL, 0, e->ek, \
                               (size_t)Gm::EK, e->m, e->r, e->ct, e->ss, e->st, (const uint8_t *)nullptr, (const uint8_t *)nullptr, e->scratch, e->work, e->n, (const uint32_t *)nullptr, (const int16_t *)nullptr); }, &e); \
        printf("  %-28s %.3f ms  (%d blocks/CU)\n", NAME, ms, BPC);                                                              \
        e.eb = eb;                                                                                                              \
    }
    printf("  LDS per workgroup: %d bytes\n", Gm::LDS_TOTAL);
    RUN(0, "full");
    RUN(1, "without A (matrix)");
    RUN(2, "without B (prf)");
    RUN(4, "without C (ring)");
    RUN(6, "A only");
    RUN(5, "B only");
    RUN(3, "C only");
    RUN(7, "nothing (launch + zetas)");
    printf("  -- scratch variant (persistent), LDS %d bytes --\n", Gm::LDS_SCRATCH_TOTAL);
    RUNS(0, 4, "scratch full");
    RUNS(0, 8, "scratch full");
    RUNS(0, 12, "scratch full");
    RUNS(0, 16, "scratch full");
    RUNS(0, 20, "scratch full");
    RUNS(6, 16, "scratch A only");
    RUNS(3, 16, "scratch C only");
    RUNS(5, 16, "scratch B only");
    // would a split into a sampling kernel (A + B) and a ring kernel (C) pay?  Each compiled alone needs fewer VGPRs
    RUNS(4, 16, "scratch A + B only");
    RUNS(4, 20, "scratch A + B only");
    RUNS(4, 24, "scratch A + B only");
    RUNS(3, 20, "scratch C only");
    RUNS(3, 24, "scratch C only");
    {
        float ms = time_ms([](void *v) { E *e = (E *)v;
            hipLaunchKernelGGL(mlkem::mlkem_hash_kernel<K>, dim3((unsigned)((e->n + 255) / 256)), dim3(256), 0, 0, e->ek, e->m, e->ss, (uint8_t *)e->r, e->n); }, &e);
        printf("  %-28s %.3f ms\n", "hash kernel", ms);
    }
    return 0;
}
