// keccak_dev.h -- Keccak-f[1600] for gfx950, one sponge state per lane (64 independent
// permutations per wavefront).
//
// Replaces the reference's scalar internal/sha3/keccakf.go:12-391 and the 4-way AVX2
// simd/keccakf1600 (f1600x.go:77-91).  CDNA4 has no 64-bit rotate or 3-input xor, but gfx950
// has V_BITOP3_B32 (arbitrary 3-input boolean) and V_ALIGNBIT_B32 (funnel shift), so a state
// is held as 25 (lo,hi) 32-bit register pairs and one round is
//     theta : 20 bitop3 (5-way column xor) + 10 alignbit (rol 1) + 50 bitop3 (s ^ C ^ rol(C))
//     rho/pi: 48 alignbit (register renaming is free)
//     chi   : 50 bitop3 (a ^ (~b & c))
//     iota  : 2 xor
// = 180 VALU instructions per round, 4320 per permutation, no memory traffic.
//
// Lane-local code in this header is __host__ __device__ so tests/hostsim can run the very same
// source on the CPU; the host branch of the three wrappers below exists only for that harness
// (the product never calls it: every exported entry point launches kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CIRCL_HD __host__ __device__ __forceinline__

namespace circl {

CIRCL_HD uint32_t bitop3_xor(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}
CIRCL_HD uint32_t bitop3_chi(uint32_t a, uint32_t b, uint32_t c) {  // a ^ (~b & c)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xD2);
#else
    return a ^ (~b & c);
#endif
}
CIRCL_HD uint32_t bitop3_xor_and(uint32_t a, uint32_t b, uint32_t c) {  // a ^ (b & c)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x78);
#else
    return a ^ (b & c);
#endif
}
// ({hi,lo} >> s)[31:0], 0 < s < 32
CIRCL_HD uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
#endif
}

struct KeccakState {
    uint32_t lo[25], hi[25];
};

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// The wave-level ordering point of every "no-wait" exchange (kyber_dev.h wave_sync<true>, dilithium_dev.h xch_sync<true>,
// keccak_f1600_coop / coop2<true>, the `handoff` points of the chain kernels): LDS stores of this wavefront in front of it, LDS loads
// of this wavefront behind it.  The hardware needs nothing (a wavefront's LDS instructions execute in order); the COMPILER must
// not move a load over a store it cannot prove disjoint, and wave_barrier alone (IntrNoMem) does not formally say so.  The two
// wavefront-scope fences do: they emit no instruction (no s_barrier, no s_waitcnt -- tests/test_abi.py checks the disassembly) and
// order every memory access of the wavefront around the barrier, as HIP's tiled-group sync does.
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#endif

// The index vector of a key-table batch as the kernels see it: item i uses table entry key_idx[i].  The host entry points check the
// vector before a launch (an index >= nkeys is CIRCL_HIP_EPARAM); a device-resident vector cannot be checked without a round
// trip, so every read goes through operator[], which bounds it to the table: an index past the end reads the LAST entry, never
// memory behind the table (a private-key table sits next to other tenants' keys).  p == nullptr: no vector (entry 0 / item i,
// as the kernel documents).  `last` = nkeys - 1.
struct KeyIdx {
    const uint32_t *p;
    uint32_t last;
    CIRCL_HD explicit operator bool() const { return p != nullptr; }
    CIRCL_HD uint32_t operator[](size_t i) const { const uint32_t k = p[i]; return k < last ? k : last; }
};

// A launch that tells the HOST it is over: the one-launch routes of small resident-key batches take an optional TailFlag; every
// workgroup makes its results visible system-wide and counts itself, the last one re-arms the counter and raises `flag` (a dword in
// page-locked host memory) to `value`.  The coalescer polls that flag instead of enqueueing a second kernel behind the batch
// (host_coalesce.hip, completion mode 2: the flag kernel cost a one-item call ~3 us of its 24).  flag == nullptr: nothing happens.
struct TailFlag {
    uint32_t *flag;
    unsigned *count;
    uint32_t value;
};
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ void tail_signal(const TailFlag &t) {
    if (t.flag == nullptr) return;  // (uniform)
    __threadfence_system();         // this thread's result rows ...
    __syncthreads();                // ... and those of the whole workgroup are out
    if (threadIdx.x == 0) {
        if (atomicAdd(t.count, 1u) == gridDim.x - 1) {  // the last workgroup of the launch
            *t.count = 0;
            __threadfence_system();
            __hip_atomic_store(t.flag, t.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
#endif

// Round constants of FIPS 202 (the reference tabulates the same 24 values in
// internal/sha3/rc.go:4-29), split into (hi,lo) halves.
#define CIRCL_RC_LIST                                                                              \
    RC(0x00000000, 0x00000001) RC(0x00000000, 0x00008082) RC(0x80000000, 0x0000808a)               \
    RC(0x80000000, 0x80008000) RC(0x00000000, 0x0000808b) RC(0x00000000, 0x80000001)               \
    RC(0x80000000, 0x80008081) RC(0x80000000, 0x00008009) RC(0x00000000, 0x0000008a)               \
    RC(0x00000000, 0x00000088) RC(0x00000000, 0x80008009) RC(0x00000000, 0x8000000a)               \
    RC(0x00000000, 0x8000808b) RC(0x80000000, 0x0000008b) RC(0x80000000, 0x00008089)               \
    RC(0x80000000, 0x00008003) RC(0x80000000, 0x00008002) RC(0x80000000, 0x00000080)               \
    RC(0x00000000, 0x0000800a) RC(0x80000000, 0x8000000a) RC(0x80000000, 0x80008081)               \
    RC(0x80000000, 0x00008080) RC(0x00000000, 0x80000001) RC(0x80000000, 0x80008008)

// (lo, hi) pairs, plus a dummy 25th entry for the prefetch of keccak_f1600's last round.  A function-local constexpr
// array becomes a private constant addressed pc-relative: a `__device__ __constant__` table is reached through the
// GOT, and that pointer load was re-done every round (two dependent scalar loads in front of each round constant).
struct RcPair {
    uint32_t lo, hi;
};
#define RC(h, l) {l, h},
CIRCL_HD RcPair rc_pair(int r) {
    constexpr uint32_t t[25][2] = {CIRCL_RC_LIST{0, 0}};
    return RcPair{t[r][0], t[r][1]};
}
#undef RC

namespace detail {
template <int I> struct IC { static constexpr int v = I; };
template <int I, int N, class F> CIRCL_HD void static_for(F &&f) {
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<I + 1, N>(f);
    }
}
// rho offsets for lane x+5y (FIPS 202 table 2; keccakf.go's rotation literals)
CIRCL_HD constexpr int rho_of(int i) {
    constexpr int t[25] = {0,  1,  62, 28, 27, 36, 44, 6,  55, 20, 3,  10, 43,
                           25, 39, 41, 45, 15, 21, 8,  18, 2,  61, 56, 14};
    return t[i];
}
template <int N> CIRCL_HD void rol64(uint32_t lo, uint32_t hi, uint32_t &olo, uint32_t &ohi) {
    if constexpr (N == 0) {
        olo = lo; ohi = hi;
    } else if constexpr (N < 32) {
        olo = alignbit(lo, hi, 32 - N); ohi = alignbit(hi, lo, 32 - N);
    } else if constexpr (N == 32) {
        olo = hi; ohi = lo;
    } else {
        olo = alignbit(hi, lo, 64 - N); ohi = alignbit(lo, hi, 64 - N);
    }
}
}  // namespace detail

// Keccak-f[1600]; first_round = 0 for 24 rounds, 12 for the 12-round "turbo" variant
// (keccakf.go:20-24).  The round loop is NOT unrolled: one round is ~1.5 KB of code and every
// inlined call site stays I-cache resident.
CIRCL_HD void keccak_f1600(KeccakState &s, int first_round = 0) {
    RcPair rc = rc_pair(first_round);
#pragma unroll 1
    for (int r = first_round; r < 24; r++) {
        const RcPair rc_next = rc_pair(r + 1);  // scalar load issued a whole round ahead of its use
        uint32_t cl[5], ch[5], rl[5], rh[5], bl[25], bh[25];
#pragma unroll
        for (int x = 0; x < 5; x++) {
            cl[x] = bitop3_xor(bitop3_xor(s.lo[x], s.lo[x + 5], s.lo[x + 10]), s.lo[x + 15], s.lo[x + 20]);
            ch[x] = bitop3_xor(bitop3_xor(s.hi[x], s.hi[x + 5], s.hi[x + 10]), s.hi[x + 15], s.hi[x + 20]);
        }
#pragma unroll
        for (int x = 0; x < 5; x++) {  // rol(C[x], 1)
            rl[x] = alignbit(cl[x], ch[x], 31);
            rh[x] = alignbit(ch[x], cl[x], 31);
        }
        detail::static_for<0, 25>([&](auto ic) {
            constexpr int i = decltype(ic)::v, x = i % 5, y = i / 5;
            const uint32_t tl = bitop3_xor(s.lo[i], cl[(x + 4) % 5], rl[(x + 1) % 5]);
            const uint32_t th = bitop3_xor(s.hi[i], ch[(x + 4) % 5], rh[(x + 1) % 5]);
            constexpr int d = y + 5 * ((2 * x + 3 * y) % 5);  // pi
            detail::rol64<detail::rho_of(i)>(tl, th, bl[d], bh[d]);
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
        rc = rc_next;
    }
}

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// Keccak-f[1600] on ONE state by one wavefront, for rare serial paths inside register-tight kernels (the second SHAKE256
// block of SampleInBall): lanes 0..24 own one 64-bit lane each, theta / pi / chi exchange through LDS.  It needs a dozen
// registers instead of the ~120 of the lane-per-state form, at ~60 instructions and four barriers per round.
// `ws` = 55 x 8 bytes of LDS: a[25] (the state, in and out), c[5], b[25].  Single-wave workgroups only.
// NW (a workgroup of several wavefronts, `ws` private to this one): the wave-level ordering points are not workgroup barriers
template <bool NW = false> __device__ __forceinline__ void keccak_f1600_coop(uint64_t *ws, int lane) {
    auto sync = [] {
        if constexpr (NW) { __builtin_amdgcn_s_waitcnt(0); wave_lds_order(); }
        else __syncthreads();
    };
    uint64_t *a = ws, *c = ws + 25, *b = ws + 30;
    const bool on = lane < 25;
    const int i = on ? lane : 0, x = i % 5, y = i / 5;
    constexpr int rho_t[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    const int rho = rho_t[i], dst = y + 5 * ((2 * x + 3 * y) % 5);
    sync();
    uint64_t v = a[i];
#pragma unroll 1
    for (int r = 0; r < 24; r++) {
        if (lane < 5) c[lane] = a[lane] ^ a[lane + 5] ^ a[lane + 10] ^ a[lane + 15] ^ a[lane + 20];
        sync();
        if (on) {
            const uint64_t c1 = c[(x + 1) % 5];
            v ^= c[(x + 4) % 5] ^ ((c1 << 1) | (c1 >> 63));
            if (rho) v = (v << rho) | (v >> (64 - rho));
            b[dst] = v;
        }
        sync();
        if (on) {
            v = b[i] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
            if (lane == 0) {
                const RcPair rc = rc_pair(r);
                v ^= ((uint64_t)rc.hi << 32) | rc.lo;
            }
            a[i] = v;
        }
        sync();
    }
}

// Keccak-f[1600] with a state on TWO ADJACENT LANES, for latency-bound chains on a mostly idle chip (hashing of a small or
// medium batch): the even lane holds the low 32-bit halves of the 25 words, the odd lane the high halves -- the natural
// little-endian dword order, so absorbing and squeezing are plain 32-bit loads and stores at dword 2 k + (lane & 1).  All of
// theta, chi and the column sums are lane-local; only a 64-bit rotation needs the other half, fetched by a DPP move from the
// partner lane (quad_perm [1,0,3,2], full rate, no LDS): rol64 by n < 32 is alignbit(own, other, 32 - n) IN BOTH LANES, by
// n > 32 alignbit(other, own, 64 - n) in both.  One round is
//     theta : 10 bitop3 (column sums) + 5 dpp + 5 alignbit (rol 1) + 25 bitop3
//     rho/pi: 24 dpp + 24 alignbit
//     chi   : 25 bitop3,   iota: 2
// = 120 instructions per lane against 180 of the lane-per-state form: a lone wavefront, which issues an instruction every ~5.4
// cycles whatever it is, finishes a permutation in 2/3 of the time, at 4/3 of the total issue slots (twice the wavefronts).
// Worth it while the lane-per-state form leaves the SIMDs at or below one wavefront each.
struct SplitState {
    uint32_t w[25];
};
__device__ __forceinline__ uint32_t split_partner(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); }
template <int N> __device__ __forceinline__ uint32_t split_rol64(uint32_t own) {
    if constexpr (N == 0) {
        return own;
    } else {
        static_assert(N != 32, "no rho offset is 32");
        const uint32_t oth = split_partner(own);
        if constexpr (N < 32) return alignbit(own, oth, 32 - N);
        else return alignbit(oth, own, 64 - N);
    }
}
__device__ __forceinline__ void keccak_f1600_split(SplitState &s, bool hi_lane) {
#pragma unroll 1
    for (int r = 0; r < 24; r++) {
        const RcPair rc = rc_pair(r);
        uint32_t c[5], rl[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; x++) c[x] = bitop3_xor(bitop3_xor(s.w[x], s.w[x + 5], s.w[x + 10]), s.w[x + 15], s.w[x + 20]);
#pragma unroll
        for (int x = 0; x < 5; x++) rl[x] = split_rol64<1>(c[x]);
        detail::static_for<0, 25>([&](auto ic) {
            constexpr int i = decltype(ic)::v, x = i % 5, y = i / 5;
            const uint32_t t = bitop3_xor(s.w[i], c[(x + 4) % 5], rl[(x + 1) % 5]);
            constexpr int d = y + 5 * ((2 * x + 3 * y) % 5);  // pi
            b[d] = split_rol64<detail::rho_of(i)>(t);
        });
#pragma unroll
        for (int y = 0; y < 25; y += 5)
#pragma unroll
            for (int x = 0; x < 5; x++) s.w[x + y] = bitop3_chi(b[x + y], b[(x + 1) % 5 + y], b[(x + 2) % 5 + y]);
        s.w[0] ^= hi_lane ? rc.hi : rc.lo;
    }
}

// Keccak-f[1600] on TWO states by one wavefront, for latency-bound chains (H(ek) || G of a small ML-KEM batch: ten dependent
// permutations).  Lanes 0..24 hold the 25 lanes of state A, lanes 32..56 those of state B, each as a (lo, hi) register pair
// that STAYS in its lane across rounds and across absorbed blocks.  A lone wavefront issues one instruction per ~5.4 cycles
// whatever their dependences, so the length of a dependent chain IS its instruction count: 180 per round in the lane-per-state
// form, ~70 in round 3's version of this one (3.7 us per permutation), 27 now (1.5 us).  A round:
//   theta  the five lanes of a column XOR their word into ONE LDS word (DS_XOR_B64, the LDS serialises the five; the buffer was
//          cleared a round earlier, two buffers alternate), then every lane reads C[x-1] and C[x+1]: 1 + 1 + 2 LDS instructions
//          and 4 VALU -- it was a store, ten loads and eight three-input XORs;
//   rho    a funnel shift by 32 - rho of the (swapped, for rho >= 32) halves; rho = 0 -- the lane of a[0] -- is a shift by 0 of
//          the SWAPPED halves, so it needs no select of its own;
//   pi     one store to the word's new place;  chi  three loads, one V_BITOP3 per half;
//   iota   the round constant comes out of a register (lane r of c.rcl / c.rch, v_readlane) and goes in with one V_BITOP3 per half,
//          a ^ (rc & mask-of-the-lane-that-owns-a[0]) -- it was a pc-relative scalar load, waited for, every round.
// No lane is ever masked off (the seven idle lanes of a half store into words nobody reads: every `if (on)` was an exec save /
// restore), and the ordering points are compiler fences in BOTH forms (the exchange area belongs to one wavefront, whose LDS
// instructions execute in order; the workgroup-barrier form waited for every store to land before the loads were issued).
// (keccak_f1600_coop above is the older single-state form with three exchanges per round.)  `ws`: 2 x 50 x 8 bytes of LDS.
struct CoopLane {
    bool on;             // this lane owns a state lane
    int i;               // which one (x + 5 y)
    uint64_t *c_self, *c_m, *c_p, *b_dst, *b0, *b1, *b2;  // column-parity words (two buffers, 8 words apart), pi / chi words
    uint32_t rsh;        // rho as a right funnel shift: 32 - (rho mod 32) (V_ALIGNBIT reads 5 bits: 32 is 0)
    bool swap;           // the halves trade places first: rho >= 32 -- and rho == 0, where the shift by 0 returns the OTHER half
    uint32_t iota;       // all ones in the lane that owns a[0], 0 elsewhere
    uint32_t rcl, rch;   // wavefront lane r < 24: round constant r
};
// Per half of the wavefront (50 words): b[25] | C0[8] | C1[8] | 7 words the idle lanes store into | 2 unused.
__device__ __forceinline__ CoopLane coop_lane(uint64_t *ws, int lane) {
    constexpr int rho_t[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    CoopLane c;
    const int half = lane >> 5, j = lane & 31;
    c.on = j < 25;
    c.i = c.on ? j : 0;
    const int x = c.i % 5, y = c.i / 5;
    uint64_t *b = ws + 50 * half, *col = b + 25, *dump = b + 41;
    c.c_self = col + (c.on ? x : 5 + (j - 25) % 3);  // (idle lanes: the three column words nobody reads)
    c.c_m = col + (x + 4) % 5;
    c.c_p = col + (x + 1) % 5;
    c.b_dst = c.on ? b + y + 5 * ((2 * x + 3 * y) % 5) : dump + (j - 25);
    c.b0 = b + c.i;
    c.b1 = b + (x + 1) % 5 + 5 * y;
    c.b2 = b + (x + 2) % 5 + 5 * y;
    const int rho = rho_t[c.i];
    c.rsh = 32u - (uint32_t)(rho & 31);
    c.swap = rho >= 32 || rho == 0;
    c.iota = (c.on && c.i == 0) ? 0xffffffffu : 0u;
    const RcPair rc = rc_pair(lane < 24 ? lane : 24);  // (entry 24 is zero)
    c.rcl = rc.lo;
    c.rch = rc.hi;
    return c;
}
// One round; BUF: which of the two column-parity buffers this round accumulates into (the other one is cleared for the next).
template <int BUF> __device__ __forceinline__ void coop2_round(uint32_t &vlo, uint32_t &vhi, const CoopLane &c, int r) {
    auto ld = [](const uint64_t *p, uint32_t &lo, uint32_t &hi) { const uint64_t w = *p; lo = (uint32_t)w; hi = (uint32_t)(w >> 32); };
    const uint32_t rcl = (uint32_t)__builtin_amdgcn_readlane((int)c.rcl, r), rch = (uint32_t)__builtin_amdgcn_readlane((int)c.rch, r);
    wave_lds_order();  // chi's reads of the previous round are done
    // theta's column parities: the five lanes of a column XOR their word into ONE LDS word (DS_XOR_B64: the LDS serialises the five)
    __hip_atomic_fetch_xor(reinterpret_cast<unsigned long long *>(c.c_self + 8 * BUF), ((unsigned long long)vhi << 32) | vlo, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WAVEFRONT);
    c.c_self[8 * (BUF ^ 1)] = 0;
    wave_lds_order();
    uint32_t cml, cmh, cpl, cph;
    ld(c.c_m + 8 * BUF, cml, cmh);
    ld(c.c_p + 8 * BUF, cpl, cph);
    // theta: v ^= C[x-1] ^ rol(C[x+1], 1)
    const uint32_t tl = bitop3_xor(vlo, cml, alignbit(cpl, cph, 31)), th = bitop3_xor(vhi, cmh, alignbit(cph, cpl, 31));
    // rho: rotate left by the lane's offset (a per-lane amount: swap the halves for offsets >= 32, funnel-shift by the rest)
    const uint32_t sl = c.swap ? th : tl, sh = c.swap ? tl : th;
    const uint32_t rl = alignbit(sl, sh, c.rsh), rh = alignbit(sh, sl, c.rsh);
    *c.b_dst = ((uint64_t)rh << 32) | rl;  // pi
    wave_lds_order();
    uint32_t b0l, b0h, b1l, b1h, b2l, b2h;
    ld(c.b0, b0l, b0h); ld(c.b1, b1l, b1h); ld(c.b2, b2l, b2h);
    vlo = bitop3_xor_and(bitop3_chi(b0l, b1l, b2l), rcl, c.iota);  // chi, iota (idle lanes carry garbage nobody reads)
    vhi = bitop3_xor_and(bitop3_chi(b0h, b1h, b2h), rch, c.iota);
}
// NW: called by ONE wavefront of a workgroup of several (no workgroup barrier may be executed); otherwise the whole (single-
// wavefront) workgroup calls it, and a barrier on entry and on exit orders it against whatever else the caller keeps in LDS.
template <bool NW = false> __device__ __forceinline__ void keccak_f1600_coop2(uint32_t &vlo, uint32_t &vhi, const CoopLane &c) {
    if constexpr (!NW) __syncthreads();
    wave_lds_order();
    c.c_self[0] = 0;
#pragma unroll 1
    for (int r = 0; r < 24; r += 2) {
        coop2_round<0>(vlo, vhi, c, r);
        coop2_round<1>(vlo, vhi, c, r + 1);
    }
    if constexpr (!NW) __syncthreads();
}
#endif

CIRCL_HD void keccak_zero(KeccakState &s) {
#pragma unroll
    for (int i = 0; i < 25; i++) s.lo[i] = s.hi[i] = 0;
}

// Sponge parameters (internal/sha3/shake.go:42-46, hashes.go:21-37): rate in 64-bit words.
constexpr int kShake128Words = 21;  // 168 B
constexpr int kShake256Words = 17;  // 136 B, also SHA3-256
constexpr int kSha3_512Words = 9;   //  72 B
constexpr uint32_t kDsShake = 0x1f, kDsSha3 = 0x06;

}  // namespace circl
