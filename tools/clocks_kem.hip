// tools/clocks_kem.hip -- profiling aid, not part of the product: where a wavefront of the headline kernel spends its wall time.
// mlkem_encrypt_kernel<K, ENCAPS, 8, true> is the product's kernel (the fused sampling pass, 16 persistent single-wavefront workgroups per CU,
// ticketed groups of G items) with the shader clock read at its phase boundaries (template bit 3): per workgroup, the cycles spent waiting
// for a ticket, in the sampling phase (matrix A^T on 9 / 16 streams + the PRF streams: Keccak rounds back to back) and in the ring phase
// (NTTs through LDS exchanges, the products against the scratch rows, compression) -- while three other wavefronts of the same SIMD are in
// whatever phase they are in.  Real keys (GPU key generation), real r (the hash kernel): the kernel does exactly the headline's work.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icircl_amd/csrc -Iinclude [-DCIRCL_KEM_RING_PRIO=0] tools/clocks_kem.hip -o tools/bin/clocks_kem
//   tools/bin/clocks_kem [log2 n = 20]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mlkem_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
using namespace circl;

template <int K, int MASK> void run(const char *what, size_t n) {
    using Gm = mlkem::Geom<K>;
    uint8_t *seed, *ek, *dk, *m, *ct, *ss, *st, *ws, *scratch;
    unsigned *work;
    const int bpc = 16, nwg = 256 * bpc;
    CK(hipMalloc(&seed, 64 * n)); CK(hipMalloc(&ek, Gm::EK * n)); CK(hipMalloc(&dk, Gm::DK * n)); CK(hipMalloc(&m, 32 * n));
    CK(hipMalloc(&ct, Gm::CT * n)); CK(hipMalloc(&ss, 32 * n)); CK(hipMalloc(&st, n)); CK(hipMalloc(&ws, 128 * n));
    CK(hipMalloc(&scratch, (size_t)nwg * Gm::SCRATCH_BYTES)); CK(hipMalloc(&work, 256));
    std::vector<uint8_t> h(64 * n);
    srand(1);
    for (auto &x : h) x = (uint8_t)rand();
    CK(hipMemcpy(seed, h.data(), 64 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(m, h.data(), 32 * n, hipMemcpyHostToDevice));
    const unsigned hb = (unsigned)((n + 255) / 256), eb = (unsigned)((n + Gm::G - 1) / Gm::G);
    hipLaunchKernelGGL(mlkem::mlkem_keygen_seed_kernel<K>, dim3(hb), dim3(256), 0, 0, seed, ws, n);
    hipLaunchKernelGGL((mlkem::mlkem_keygen_kernel<K, false>), dim3(eb), dim3(64), Gm::LDS_TOTAL, 0, (const uint8_t *)ws, ek, dk, scratch, (unsigned *)nullptr, n);
    hipLaunchKernelGGL(mlkem::mlkem_hash_kernel<K>, dim3(hb), dim3(256), 0, 0, ek, m, ss, ws, n);
    CK(hipDeviceSynchronize());
    uint64_t *prof;
    CK(hipMalloc(&prof, (size_t)nwg * 4 * 8));
    CK(hipMemset(prof, 0, (size_t)nwg * 4 * 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {  // (the last launch is the one read)
        CK(hipMemsetAsync(work, 0, 256, 0));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((mlkem::mlkem_encrypt_kernel<K, mlkem::ENCAPS, MASK, true>), dim3(nwg), dim3(64), Gm::LDS_SCRATCH_TOTAL, 0, ek, (size_t)Gm::EK, m, (const uint8_t *)ws, ct,
                           ss, st, (const uint8_t *)nullptr, (const uint8_t *)nullptr, scratch, work, n, KeyIdx{nullptr, 0}, (const int16_t *)prof);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
    }
    std::vector<uint64_t> hp((size_t)nwg * 4);
    CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
    double sum[4] = {0, 0, 0, 0}, wmin = 1e30, wmax = 0;
    for (int w = 0; w < nwg; w++) {
        double tot = 0;
        for (int k = 0; k < 4; k++) { sum[k] += (double)hp[(size_t)w * 4 + k]; if (k < 3) tot += (double)hp[(size_t)w * 4 + k]; }
        wmin = std::min(wmin, tot); wmax = std::max(wmax, tot);
    }
    const double tot = sum[0] + sum[1] + sum[2], items = sum[3] > 0 ? sum[3] : (double)n;  // (a variant without the ring phase counts no items)
    printf("%-30s %7.3f ms | wavefront cycles per item: ticket %6.0f  sampling %7.0f  ring %7.0f  (sum %7.0f) | share sampling %.3f  ring %.3f  ticket %.3f | "
           "busiest / idlest wavefront %.3f / %.3f of the mean | clock %.0f MHz\n",
           what, ms, sum[0] / items, sum[1] / items, sum[2] / items, tot / items, sum[1] / tot, sum[2] / tot, sum[0] / tot, wmax / (tot / nwg), wmin / (tot / nwg),
           tot / nwg / (ms * 1e3));
    for (void *p : {(void *)seed, (void *)ek, (void *)dk, (void *)m, (void *)ct, (void *)ss, (void *)st, (void *)ws, (void *)scratch, (void *)work, (void *)prof}) CK(hipFree(p));
}

int main(int argc, char **argv) {
    const size_t n = size_t(1) << (argc > 1 ? atoi(argv[1]) : 20);
    printf("== mlkem_encrypt_kernel, shader clock at the phase boundaries, n = %zu, 16 wavefronts per CU x 256 CUs; built with CIRCL_KEM_RING_PRIO=%d ==\n", n, CIRCL_KEM_RING_PRIO);
    run<3, 8>("ML-KEM-768 full (the headline)", n);
    run<3, 8 + 4>("ML-KEM-768 sampling alone", n);   // (the un-fused matrix and PRF passes: the fused pass exists only in the full kernel)
    run<3, 8 + 3>("ML-KEM-768 ring phase alone", n);
    run<4, 8>("ML-KEM-1024 full", n / 4);
    return 0;
}
