// tools/ablate_dsa.hip -- profiling aid: phase ablation of mldsa_verify_kernel<65> on random inputs
// (signatures are garbage, so every item fails verification, but the work done is the same).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mldsa_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
using namespace circl;

static uint8_t *g_scratch;
static unsigned *g_work;
static int g_bpc = 16;  // workgroups per CU
template <int MODE, int MASK> float run(const uint8_t *pk, const uint8_t *sig, uint8_t *muw1, uint8_t *ball, uint8_t *fail, size_t n) {
    using G = mldsa::DG<MODE>;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto launch = [&] {
        CK(hipMemsetAsync(g_work, 0, 256, 0));
        hipLaunchKernelGGL((mldsa::mldsa_verify_kernel<MODE, MASK>), dim3(256 * g_bpc), dim3(64), G::LDS_V_TOTAL, 0, pk, sig, muw1,
                           (const uint8_t *)ball, fail, g_scratch, g_work, n, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 3; i++) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 3;
}

template <int MODE> void bench(size_t n) {
    using G = mldsa::DG<MODE>;
    uint8_t *pk, *sig, *muw1, *ball, *fail;
    CK(hipMalloc(&pk, n * G::PK)); CK(hipMalloc(&sig, n * G::SIG + 64)); CK(hipMalloc(&muw1, n * G::MUW1)); CK(hipMalloc(&ball, n * 200)); CK(hipMalloc(&fail, n));
    std::vector<uint8_t> h(n * G::SIG);
    srand(2);
    for (auto &x : h) x = (uint8_t)rand();
    // make z fields small so decoding behaves like a real signature; hints all zero
    CK(hipMemcpy(sig, h.data(), n * G::SIG, hipMemcpyHostToDevice));
    CK(hipMemcpy(pk, h.data(), n * G::PK, hipMemcpyHostToDevice));
    CK(hipMemcpy(ball, h.data(), n * 200, hipMemcpyHostToDevice));
    CK(hipMemset(fail, 0, n));
    printf("== ML-DSA-%d verify kernel, n = %zu, LDS %d bytes, IT %d ==\n", MODE, n, G::LDS_V_TOTAL, G::IT);
    for (int bpc : {4, 8, 12, 16}) {
        g_bpc = bpc;
        printf("  full            %.3f ms  (%d workgroups/CU)\n", run<MODE, 0>(pk, sig, muw1, ball, fail, n), bpc);
    }
    g_bpc = 16;
    printf("  full, no row stores          %.3f ms\n", run<MODE, 8>(pk, sig, muw1, ball, fail, n));
    printf("  full, row loads hit L2       %.3f ms\n", run<MODE, 16>(pk, sig, muw1, ball, fail, n));
    printf("  full, neither                %.3f ms\n", run<MODE, 24>(pk, sig, muw1, ball, fail, n));
    printf("  phase A only    %.3f ms\n", run<MODE, 6>(pk, sig, muw1, ball, fail, n));
    printf("  phase A only, no row stores  %.3f ms\n", run<MODE, 14>(pk, sig, muw1, ball, fail, n));
    printf("  phase 1 only    %.3f ms\n", run<MODE, 5>(pk, sig, muw1, ball, fail, n));
    printf("  phases 2+3 only %.3f ms\n", run<MODE, 3>(pk, sig, muw1, ball, fail, n));
    printf("  nothing         %.3f ms\n", run<MODE, 7>(pk, sig, muw1, ball, fail, n));
}

int main() {
    CK(hipMalloc(&g_scratch, (size_t)256 * 16 * 65536));
    CK(hipMalloc(&g_work, 256));
    bench<65>(1 << 16);
    bench<87>(1 << 16);
    return 0;
}
