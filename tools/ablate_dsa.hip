// tools/ablate_dsa.hip -- profiling aid: phase ablation of mldsa_verify_kernel<65> on random inputs
// (signatures are garbage, so every item fails verification, but the work done is the same).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
                           (const uint8_t *)ball, fail, g_scratch, g_work, n, KeyIdx{nullptr, 0}, (const uint32_t *)nullptr);
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

// `ablate_dsa phases <65|87> <log2 n>`: one run per phase subset at the config's own batch size, for the counter passes of
// tools/verify_phases.sh (kernel names carry the mask: mldsa_verify_kernel<65, MASK, 0>)
template <int MODE> void phases(size_t n) {
    using G = mldsa::DG<MODE>;
    uint8_t *pk, *sig, *muw1, *ball, *fail;
    CK(hipMalloc(&pk, n * G::PK)); CK(hipMalloc(&sig, n * G::SIG + 64)); CK(hipMalloc(&muw1, n * G::MUW1)); CK(hipMalloc(&ball, n * 200)); CK(hipMalloc(&fail, n));
    std::vector<uint8_t> h(n * G::SIG);
    srand(2);
    for (auto &x : h) x = (uint8_t)rand();
    CK(hipMemcpy(sig, h.data(), n * G::SIG, hipMemcpyHostToDevice));
    CK(hipMemcpy(pk, h.data(), n * G::PK, hipMemcpyHostToDevice));
    CK(hipMemcpy(ball, h.data(), n * 200, hipMemcpyHostToDevice));
    CK(hipMemset(fail, 0, n));
    g_bpc = 16;
    printf("== ML-DSA-%d verify kernel phases, n = %zu ==\n", MODE, n);
    printf("  mask 0  full                          %.3f ms\n", run<MODE, 0>(pk, sig, muw1, ball, fail, n));
    printf("  mask 6  phase A (ExpandA) only        %.3f ms\n", run<MODE, 6>(pk, sig, muw1, ball, fail, n));
    printf("  mask 14 phase A, no row stores        %.3f ms\n", run<MODE, 14>(pk, sig, muw1, ball, fail, n));
    printf("  mask 5  phase 1 (sig decode, z-hat)   %.3f ms\n", run<MODE, 5>(pk, sig, muw1, ball, fail, n));
    printf("  mask 3  phases 2+3 (A z, t1, w1)      %.3f ms\n", run<MODE, 3>(pk, sig, muw1, ball, fail, n));
    printf("  mask 1  phases 1+2+3 (no ExpandA)     %.3f ms\n", run<MODE, 1>(pk, sig, muw1, ball, fail, n));
    printf("  mask 7  nothing (ticket loop)         %.3f ms\n", run<MODE, 7>(pk, sig, muw1, ball, fail, n));
}

// `ablate_dsa clocks <65|87> <log2 n>`: the FULL kernel with the shader clock read at every phase boundary (ABLATE bit 6): where a
// wavefront's wall time goes when four co-resident wavefronts overlap their phases
template <int MODE, int MASK> void clocks_run(const char *what, const uint8_t *pk, const uint8_t *sig, uint8_t *muw1, uint8_t *ball, uint8_t *fail, size_t n) {
    using G = mldsa::DG<MODE>;
    const int nwg = 256 * g_bpc;
    uint64_t *prof;
    CK(hipMalloc(&prof, (size_t)nwg * 5 * 8));
    CK(hipMemset(prof, 0, (size_t)nwg * 5 * 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {  // (the second launch is the one read)
        CK(hipMemsetAsync(g_work, 0, 256, 0));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((mldsa::mldsa_verify_kernel<MODE, MASK>), dim3(nwg), dim3(64), (MASK & 32) ? G::LDS_V_PAIR : G::LDS_V_TOTAL, 0, pk, sig, muw1, (const uint8_t *)ball, fail, g_scratch,
                           g_work, n, KeyIdx{nullptr, 0}, (const uint32_t *)prof);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
    }
    std::vector<uint64_t> h((size_t)nwg * 5);
    CK(hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost));
    double sum[5] = {0, 0, 0, 0, 0}, wmin = 1e30, wmax = 0;
    for (int w = 0; w < nwg; w++) {
        double tot = 0;
        for (int k = 0; k < 5; k++) { sum[k] += (double)h[(size_t)w * 5 + k]; if (k < 4) tot += (double)h[(size_t)w * 5 + k]; }
        wmin = std::min(wmin, tot); wmax = std::max(wmax, tot);
    }
    const double tot = sum[0] + sum[1] + sum[2] + sum[3], items = sum[4];
    printf("%-34s %7.3f ms | wavefront cycles per item: ticket %7.0f  phase A %7.0f  phase 1 %7.0f  phases 2+3 %7.0f  (sum %7.0f) | share A %.3f  1 %.3f  2+3 %.3f  ticket %.3f | "
           "busiest / idlest wavefront %.3f / %.3f of the mean | clock %.0f MHz\n",
           what, ms, sum[0] / items, sum[1] / items, sum[2] / items, sum[3] / items, tot / items, sum[1] / tot, sum[2] / tot, sum[3] / tot, sum[0] / tot,
           wmax / (tot / nwg), wmin / (tot / nwg), tot / nwg / (ms * 1e3));
    CK(hipFree(prof));
}
template <int MODE> void clocks(size_t n) {
    using G = mldsa::DG<MODE>;
    uint8_t *pk, *sig, *muw1, *ball, *fail;
    CK(hipMalloc(&pk, n * G::PK)); CK(hipMalloc(&sig, n * G::SIG + 64)); CK(hipMalloc(&muw1, n * G::MUW1)); CK(hipMalloc(&ball, n * 200)); CK(hipMalloc(&fail, n));
    std::vector<uint8_t> h(n * G::SIG);
    srand(2);
    for (auto &x : h) x = (uint8_t)rand();
    CK(hipMemcpy(sig, h.data(), n * G::SIG, hipMemcpyHostToDevice));
    CK(hipMemcpy(pk, h.data(), n * G::PK, hipMemcpyHostToDevice));
    CK(hipMemcpy(ball, h.data(), n * 200, hipMemcpyHostToDevice));
    CK(hipMemset(fail, 0, n));
    printf("== ML-DSA-%d verify kernel, shader clock at the phase boundaries, n = %zu, %d wavefronts per CU x 256 CUs; built with CIRCL_DSA_WAVES_PER_EU=%d CIRCL_DSA_VERIFY_PRIO=%d ==\n",
           MODE, n, g_bpc, CIRCL_DSA_WAVES_PER_EU, CIRCL_DSA_VERIFY_PRIO);
    clocks_run<MODE, 64>("full", pk, sig, muw1, ball, fail, n);
    clocks_run<MODE, 64 + 8>("full, no row stores", pk, sig, muw1, ball, fail, n);
    clocks_run<MODE, 64 + 16>("full, row loads hit one row", pk, sig, muw1, ball, fail, n);
    clocks_run<MODE, 64 + 6>("phase A alone", pk, sig, muw1, ball, fail, n);
    clocks_run<MODE, 64 + 1>("phases 1+2+3 alone", pk, sig, muw1, ball, fail, n);
    clocks_run<MODE, 64 + 32>("full, paired transforms", pk, sig, muw1, ball, fail, n);
}

int main(int argc, char **argv) {
    CK(hipMalloc(&g_scratch, (size_t)256 * 32 * 65536));
    CK(hipMalloc(&g_work, 256));
    if (argc >= 4 && !strcmp(argv[1], "clocks")) {
        const size_t n = size_t(1) << atoi(argv[3]);
        if (argc >= 5) g_bpc = atoi(argv[4]);  // workgroups (= wavefronts) per CU: 16 = 4 per SIMD
        if (atoi(argv[2]) == 87) clocks<87>(n); else clocks<65>(n);
        return 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "phases")) {
        const size_t n = size_t(1) << atoi(argv[3]);
        if (atoi(argv[2]) == 87) phases<87>(n); else phases<65>(n);
        return 0;
    }
    bench<65>(1 << 16);
    bench<87>(1 << 16);
    return 0;
}
