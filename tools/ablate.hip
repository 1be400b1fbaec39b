// tools/ablate.hip -- profiling aid, not part of the product.
//   1. VALU issue-rate probe: Keccak-f[1600] back to back (no memory) at 1/2/4/8 waves per SIMD.
//   2. Phase ablation of mlkem_encrypt_kernel<3>: time with phase A / B / C removed.
//   3. (round 3) The open question of VERDICT r02 item 5, with the kill criteria written first:
//      a. Keccak at <= 64 VGPRs and 8 waves per SIMD.  Built with -DCIRCL_KEM_WAVES_PER_EU=8 (build/ablate_w8) the REAL encrypt
//         kernel and the pure-Keccak loop are compiled under a 64-VGPR cap (the compiler parks what does not fit in scratch);
//         they then run at 24 / 28 / 32 workgroups per CU.  KILL: the sampling phases (A + B) at 8 waves must beat the same
//         phases at 4 waves (build/ablate, 16 workgroups per CU) by more than 5 %, else the formulation is dropped -- a
//         hand-parked variant can only win what this one loses to its spill traffic, and the pure-Keccak loop bounds that.
//      b. A different fast:slow mix for rho.  The 58 V_ALIGNBIT of a round are the slow class (4.2 cycles); the only other
//         encodings of a 64-bit rotation on this ISA are a right shift (fast class) + V_LSHL_OR (slow class) per half, and
//         V_PERM_B32 for the byte-aligned offsets (8, 56).  KILL: a variant must lower the time per permutation at 4 AND at 8
//         waves per SIMD; otherwise V_ALIGNBIT stays.
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

// Keccak-f[1600] with the rotation encoding as a policy (probe 3b); ROT = 0: V_ALIGNBIT (the product's form), 1: right shift +
// V_LSHL_OR per half, 2: V_ALIGNBIT except V_PERM_B32 for the byte-aligned offsets
template <int ROT, int N> __device__ __forceinline__ void rol64v(uint32_t lo, uint32_t hi, uint32_t &olo, uint32_t &ohi) {
    constexpr int S = N % 32;
    const uint32_t a = N < 32 ? lo : hi, b = N < 32 ? hi : lo;  // rotate (a, b) left by S < 32, a = the new low half's source
    if constexpr (S == 0) {
        olo = a; ohi = b;
    } else if constexpr (ROT == 1) {
        // inline assembly: written as C shifts the compiler recognises the funnel shift and emits V_ALIGNBIT again
        uint32_t t0, t1;
        asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(t0) : "n"(32 - S), "v"(b));
        asm volatile("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(olo) : "v"(a), "n"(S), "v"(t0));
        asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(t1) : "n"(32 - S), "v"(a));
        asm volatile("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(ohi) : "v"(b), "n"(S), "v"(t1));
    } else if constexpr (ROT == 2 && S % 8 == 0) {
        constexpr uint32_t sel = S == 8 ? 0x02010007u : S == 16 ? 0x01000706u : 0x00070605u;  // bytes of (hi operand = src0, lo operand = src1)
        olo = __builtin_amdgcn_perm(a, b, sel);
        ohi = __builtin_amdgcn_perm(b, a, sel);
    } else {
        olo = alignbit(a, b, 32 - S);
        ohi = alignbit(b, a, 32 - S);
    }
}
template <int ROT> __device__ __forceinline__ void keccak_variant(KeccakState &s) {
#pragma unroll 1
    for (int r = 0; r < 24; r++) {
        const RcPair rc = rc_pair(r);
        uint32_t cl[5], ch[5], rl[5], rh[5], bl[25], bh[25];
#pragma unroll
        for (int x = 0; x < 5; x++) {
            cl[x] = bitop3_xor(bitop3_xor(s.lo[x], s.lo[x + 5], s.lo[x + 10]), s.lo[x + 15], s.lo[x + 20]);
            ch[x] = bitop3_xor(bitop3_xor(s.hi[x], s.hi[x + 5], s.hi[x + 10]), s.hi[x + 15], s.hi[x + 20]);
        }
#pragma unroll
        for (int x = 0; x < 5; x++) rol64v<ROT, 1>(cl[x], ch[x], rl[x], rh[x]);
        detail::static_for<0, 25>([&](auto ic) {
            constexpr int i = decltype(ic)::v, x = i % 5, y = i / 5;
            const uint32_t tl = bitop3_xor(s.lo[i], cl[(x + 4) % 5], rl[(x + 1) % 5]);
            const uint32_t th = bitop3_xor(s.hi[i], ch[(x + 4) % 5], rh[(x + 1) % 5]);
            constexpr int d = y + 5 * ((2 * x + 3 * y) % 5);
            rol64v<ROT, detail::rho_of(i)>(tl, th, bl[d], bh[d]);
        });
#pragma unroll
        for (int y = 0; y < 25; y += 5)
#pragma unroll
            for (int x = 0; x < 5; x++) {
                s.lo[x + y] = bitop3_chi(bl[x + y], bl[(x + 1) % 5 + y], bl[(x + 2) % 5 + y]);
                s.hi[x + y] = bitop3_chi(bh[x + y], bh[(x + 1) % 5 + y], bh[(x + 2) % 5 + y]);
            }
        s.lo[0] ^= rc.lo;
        s.hi[0] ^= rc.hi;
    }
}
template <int ROT> __global__ void __launch_bounds__(64) keccak_variant_loop(uint64_t *out, int perms) {
    extern __shared__ uint8_t pad[];
    KeccakState s;
    keccak_zero(s);
    s.lo[0] = threadIdx.x + blockIdx.x * 64;
    for (int i = 0; i < perms; i++) keccak_variant<ROT>(s);
    if (s.lo[3] == 0x12345678u && pad[threadIdx.x] == 77) out[0] = s.lo[0] ^ s.hi[7];
}
// the product's permutation under a hard register cap (probe 3a): -DCIRCL_KEM_WAVES_PER_EU=8 -> 64 VGPRs, the rest in scratch
__global__ void __launch_bounds__(64, CIRCL_KEM_WAVES_PER_EU) keccak_loop_capped(uint64_t *out, int perms) {
    extern __shared__ uint8_t pad[];
    KeccakState s;
    keccak_zero(s);
    s.lo[0] = threadIdx.x + blockIdx.x * 64;
    for (int i = 0; i < perms; i++) keccak_f1600(s);
    if (s.lo[3] == 0x12345678u && pad[threadIdx.x] == 77) out[0] = s.lo[0];
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
    printf("== probe 3b: rotation encodings of Keccak-f[1600] (0 = V_ALIGNBIT, 1 = shift + V_LSHL_OR, 2 = V_PERM_B32 for offsets 8 / 56) ==\n");
    for (int wps : {4, 8}) {
        const int lds = 160 * 1024 / (4 * wps) - 512;
        const int blocks = 256 * 4 * wps * 4, perms = 200;
        struct A { uint64_t *o; int b, l, p; } a{d_out, blocks, lds, perms};
        CK(hipFuncSetAttribute((const void *)keccak_variant_loop<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute((const void *)keccak_variant_loop<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute((const void *)keccak_variant_loop<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        float m0 = time_ms([](void *v) { A *a = (A *)v; hipLaunchKernelGGL(keccak_variant_loop<0>, dim3(a->b), dim3(64), a->l, 0, a->o, a->p); }, &a);
        float m1 = time_ms([](void *v) { A *a = (A *)v; hipLaunchKernelGGL(keccak_variant_loop<1>, dim3(a->b), dim3(64), a->l, 0, a->o, a->p); }, &a);
        float m2 = time_ms([](void *v) { A *a = (A *)v; hipLaunchKernelGGL(keccak_variant_loop<2>, dim3(a->b), dim3(64), a->l, 0, a->o, a->p); }, &a);
        const double perms_total = (double)blocks * 64 * perms;
        printf("  %d wave/SIMD: alignbit %.3e perm/s | shift+lshl_or %.3e perm/s (%+.1f %%) | perm_b32 for 8/56 %.3e perm/s (%+.1f %%)\n", wps,
               perms_total / (m0 * 1e-3), perms_total / (m1 * 1e-3), 100.0 * (m0 / m1 - 1.0), perms_total / (m2 * 1e-3), 100.0 * (m0 / m2 - 1.0));
    }
    printf("== probe 3a: Keccak-f[1600] compiled with __launch_bounds__(64, %d) ==\n", CIRCL_KEM_WAVES_PER_EU);
    for (int wps : {4, 6, 8}) {
        if (wps > CIRCL_KEM_WAVES_PER_EU && CIRCL_KEM_WAVES_PER_EU >= 4) continue;
        const int lds = 160 * 1024 / (4 * wps) - 512;
        const int blocks = 256 * 4 * wps * 4, perms = 200;
        CK(hipFuncSetAttribute((const void *)keccak_loop_capped, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        struct A { uint64_t *o; int b, l, p; } a{d_out, blocks, lds, perms};
        float ms = time_ms([](void *v) { A *a = (A *)v; hipLaunchKernelGGL(keccak_loop_capped, dim3(a->b), dim3(64), a->l, 0, a->o, a->p); }, &a);
        printf("  %d wave/SIMD: %.3e perm/s\n", wps, (double)blocks * 64 * perms / (ms * 1e-3));
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
#if CIRCL_KEM_WAVES_PER_EU > 4
    // probe 3a on the real kernel: the 64-VGPR build at the residency it was compiled for
    RUNS(4, 24, "scratch A + B only");
    RUNS(4, 28, "scratch A + B only");
    RUNS(4, 32, "scratch A + B only");
    RUNS(0, 24, "scratch full");
    RUNS(0, 32, "scratch full");
#endif
    {
        float ms = time_ms([](void *v) { E *e = (E *)v;
            hipLaunchKernelGGL(mlkem::mlkem_hash_kernel<K>, dim3((unsigned)((e->n + 255) / 256)), dim3(256), 0, 0, e->ek, e->m, e->ss, (uint8_t *)e->r, e->n); }, &e);
        printf("  %-28s %.3f ms\n", "hash kernel", ms);
    }
    return 0;
}
