// dilithium_dev.h -- the Dilithium ring Z_8380417[x]/(x^256+1) on gfx950, one polynomial per
// wavefront, 4 coefficients per lane.
//
// Replaces sign/internal/dilithium {field,ntt,poly}.go and its AVX2 assembler.  Montgomery arithmetic
// with R = 2^32 as in the reference (field.go:20-24), on 32-bit halves:
//     mont32(a, b) = a b 2^-32 mod q  =  hi(a b) - hi(lo(a b) q^-1 * q) + q     (4 multiplies, 2 adds)
// (the low words of a b and m q are equal, so the high words subtract without a borrow).  On gfx950
// V_MUL_LO_U32 / V_MUL_HI_U32 issue at the same rate as the 24-bit multiplies (tools/gen_valu_rate.py,
// profiles/r01_valu_issue_rates.txt), so this is cheaper than a 24-bit Montgomery product, and operands
// need no pre-reduction: any a < 2^32 with a b < 2^32 q gives a result in (0, 2q).
// Twiddles are stored pre-multiplied by 2^32 (as ntt.go:19-57 does), so a butterfly computes the plain
// product zeta*b; coefficients are plain residues mod q (no stray Montgomery factor), which is all that
// the packed outputs of ML-DSA depend on.  The lazy schedule is the reference's: the forward transform
// grows by 2q per layer (ntt.go:111-184), the inverse doubles per layer and stays below 512 q < 2^32
// (ntt.go:191-217).
//
// Register layouts L1..L4 are those of kyber_dev.h; the 8 NTT layers (strides 128..1) are done
// two at a time on register-local pairs, with three re-distributions through 1 KB of LDS.
#pragma once
#include "kyber_dev.h"

namespace circl {
namespace dilithium {

constexpr uint32_t Q = 8380417;  // 2^23 - 2^13 + 1
constexpr int D = 13;

constexpr uint32_t cpow(uint64_t b, uint32_t e) {
    uint64_t r = 1;
    b %= Q;
    while (e) {
        if (e & 1) r = r * b % Q;
        b = b * b % Q;
        e >>= 1;
    }
    return (uint32_t)r;
}
constexpr uint32_t R32 = (uint32_t)((1ull << 32) % Q);           // 2^32 mod q = 4193792 (params.go ROver256 * 256)
constexpr uint32_t R32SQ = (uint32_t)((uint64_t)R32 * R32 % Q);  // 2^64 mod q = 2365951 (field.go R2)
// q^-1 mod 2^32 (Newton iteration; checked by static_assert below)
constexpr uint32_t qinv32() {
    uint32_t x = 1;
    for (int i = 0; i < 6; i++) x = x * (2 - Q * x);
    return x;
}
constexpr uint32_t QINV32 = qinv32();
static_assert((uint32_t)(Q * QINV32) == 1u, "q * qinv == 1 mod 2^32");
static_assert(R32 == 4193792u && R32SQ == 2365951u, "the reference's Montgomery constants");

CIRCL_HD uint32_t umul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (uint32_t)((uint64_t)(a & 0xffffff) * (b & 0xffffff));
#endif
}
CIRCL_HD uint32_t umulhi24(uint32_t a, uint32_t b) {  // bits 32..47 of the 48-bit product
    // the AMDGPU backend selects V_MUL_HI_U32_U24 for this shape (operands provably 24-bit)
    return (uint32_t)(((uint64_t)(a & 0xffffff) * (uint64_t)(b & 0xffffff)) >> 32);
}

CIRCL_HD uint32_t umulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
// a b 2^-32 mod q for a b < 2^32 q; result in (0, 2q)  (field.go:20-24 montReduceLe2Q on a 64-bit product)
CIRCL_HD uint32_t mont32(uint32_t a, uint32_t b) {
    uint32_t lo = a * b;
    const uint32_t hi = umulhi32(a, b);
#if defined(__HIP_DEVICE_COMPILE__)
    // keep lo opaque: otherwise LLVM rewrites (a b) qinv as a (b qinv) and hoists b qinv for every
    // loop-invariant twiddle, doubling the registers the twiddles occupy (spills in the big kernels)
    asm("" : "+v"(lo));
#endif
    const uint32_t m = lo * QINV32;
    return hi - umulhi32(m, Q) + Q;
}
// The same reduction of a 64-bit value t < 2^32 q (a lazily accumulated sum of products): t 2^-32 mod q in (0, 2q).
// A dot product of L terms is then L V_MAD_U64_U32 and ONE reduction instead of L mont32.
CIRCL_HD uint32_t mont64(uint64_t t) {
    uint32_t lo = (uint32_t)t;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(lo));  // as in mont32
#endif
    return (uint32_t)(t >> 32) - umulhi32(lo * QINV32, Q) + Q;
}
// field.go:5-13 ReduceLe2Q generalised: x < 2^32 -> < 2^24 (and < 2q when x < 2^28)
CIRCL_HD uint32_t fold(uint32_t x) { return (x & 0x7fffff) + umul24(x >> 23, 8191u); }
// field.go:27-31 le2qModQ: x < 2q -> [0,q)
CIRCL_HD uint32_t csubq(uint32_t x) {
    x -= Q;
    return x + ((uint32_t)((int32_t)x >> 31) & Q);
}
CIRCL_HD uint32_t normalize(uint32_t x) { return csubq(fold(fold(x))); }  // any x < 2^32

// ntt.go:19-57: zeta^brv8(k) times 2^32
struct ZetaTable {
    uint32_t v[256];
};
constexpr ZetaTable make_zetas() {
    ZetaTable t{};
    for (int i = 0; i < 256; i++) {
        int brv = 0;
        for (int b = 0; b < 8; b++) brv |= ((i >> b) & 1) << (7 - b);
        t.v[i] = (uint32_t)((uint64_t)cpow(1753, (uint32_t)brv) * R32 % Q);
    }
    return t;
}
static __device__ __constant__ const ZetaTable kZetasDev = make_zetas();
static const ZetaTable kZetasHost = make_zetas();
CIRCL_HD uint32_t zeta(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return kZetasDev.v[i];
#else
    return kZetasHost.v[i];
#endif
}

struct LaneZetas {
    uint32_t f2, f3a, f3b, f4, f5a, f5b, f6, f7a, f7b;
    uint32_t i2, i3a, i3b, i4, i5a, i5b, i6, i7a, i7b;
};
// forward layer t (stride 128>>t): k = 2^t + block; inverse: index 2^(t+1) - 1 - block, with the
// butterfly written as b' = zeta * (b - a)  (ntt.go:191-217: InvZetas[k] = -Zetas[255-k]).
CIRCL_HD LaneZetas load_lane_zetas(int lane) {
    LaneZetas z;
    const int h = lane >> 4, m = lane >> 2;
    z.f2 = zeta(4 + h);
    z.f3a = zeta(8 + 2 * h);
    z.f3b = zeta(9 + 2 * h);
    z.f4 = zeta(16 + m);
    z.f5a = zeta(32 + 2 * m);
    z.f5b = zeta(33 + 2 * m);
    z.f6 = zeta(64 + lane);
    z.f7a = zeta(128 + 2 * lane);
    z.f7b = zeta(129 + 2 * lane);
    z.i2 = zeta(7 - h);
    z.i3a = zeta(15 - 2 * h);
    z.i3b = zeta(14 - 2 * h);
    z.i4 = zeta(31 - m);
    z.i5a = zeta(63 - 2 * m);
    z.i5b = zeta(62 - 2 * m);
    z.i6 = zeta(127 - lane);
    z.i7a = zeta(255 - 2 * lane);
    z.i7b = zeta(254 - 2 * lane);
    return z;
}

// Cooley-Tukey: (a, b) -> (a + zb, a - zb); b any value < 2^32, a grows by at most 2q
CIRCL_HD void ct(uint32_t &a, uint32_t &b, uint32_t z) {
    const uint32_t t = mont32(b, z);  // < 2q
    b = a + (2 * Q - t);
    a = a + t;
}
// Gentleman-Sande, layer with inputs < BOUND q: (a, b) -> (a + b, z (a - b)), outputs < 2 BOUND q and < 2q
template <uint32_t BOUND> CIRCL_HD void gs(uint32_t &a, uint32_t &b, uint32_t z) {
    static_assert((uint64_t)2 * BOUND * Q < (1ull << 32), "a - b + BOUND q fits 32 bits");
    const uint32_t t = b + (BOUND * Q - a);
    a = a + b;
    b = mont32(t, z);
}

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// The exchange buffer is padded by four words after every 32 (word i lives at i + 4 (i >> 5), kXchWords in all): with the
// plain layout the 32 lanes of a half-wave hit 8 distinct banks in layout L3 (4-way conflict on every access) and the
// single-word accesses of L4 likewise; padded, L1 and L3 are conflict-free, L4 is one 16-byte access per lane, and only L2
// keeps its 2-way conflict (no additive padding serves L2 and L3 at once).  The padding is additive in r for every layout,
// so the four accesses of a lane still differ by immediate offsets from one address register.
// LDS time per transform: 64 instead of 112-136 cycles.  Measured neutral on the kernels' run time (the exchanges are not on
// their critical path: sign_w is HBM-bound, verify VALU-bound) -- kept because it frees LDS issue slots for nothing.
constexpr int kXchWords = 256 + 4 * 8;
__device__ __forceinline__ int xch_pad(int i) { return i + 4 * (i >> 5); }
// NW ("no wait"): the buffer belongs to ONE wavefront, whose LDS instructions execute in order -- the reads behind the writes need no
// s_waitcnt in between (what __syncthreads() costs a single-wave workgroup), only the compiler must keep their order.
constexpr bool kNWDefault = false;  // (measured neutral, +0.2 %, on the verification kernel: left off there)
template <bool NW> __device__ __forceinline__ void xch_sync() {
    if constexpr (NW) wave_lds_order();
    else __syncthreads();
}
template <int FROM, int TO, bool NW = kNWDefault> __device__ __forceinline__ void relayout(uint32_t (&c)[4], uint32_t *xch, int lane) {
    // padded position of coefficient r of this lane: base(lane) + an immediate per r
    auto pos = [&](int which, int r) {
        switch (which) {
        case 1: return lane + 4 * (lane >> 5) + 72 * r;                                              // l + 64 r:  (l + 64 r) >> 5 = (l >> 5) + 2 r
        case 2: return ((lane >> 4) * 72) + (lane & 15) + 16 * r + 4 * (r >> 1);                    // 64 a + 16 r + j:  >> 5 = 2 a + (r >> 1)
        case 3: return ((lane >> 2) << 4) + 4 * (lane >> 3) + (lane & 3) + 4 * r;                   // 16 g + 4 r + j:  >> 5 = g >> 1
        default: return 4 * lane + 4 * (lane >> 3) + r;                                             // 4 l + r:  >> 5 = l >> 3
        }
    };
    xch_sync<NW>();
#pragma unroll
    for (int r = 0; r < 4; r++) xch[pos(FROM, r)] = c[r];
    xch_sync<NW>();
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = xch[pos(TO, r)];
}

// Poly.NTT (ntt.go:166-183).  In: layout L1, c < 2^32 - 16 q.  Out: layout L4, c < in + 16 q, plain residues.
template <bool NW = kNWDefault> __device__ __forceinline__ void ntt(uint32_t (&c)[4], const LaneZetas &z, uint32_t *xch, int lane) {
    const uint32_t z1 = zeta(1), z2 = zeta(2), z3 = zeta(3);
    ct(c[0], c[2], z1); ct(c[1], c[3], z1);
    ct(c[0], c[1], z2); ct(c[2], c[3], z3);
    relayout<1, 2, NW>(c, xch, lane);
    ct(c[0], c[2], z.f2); ct(c[1], c[3], z.f2);
    ct(c[0], c[1], z.f3a); ct(c[2], c[3], z.f3b);
    relayout<2, 3, NW>(c, xch, lane);
    ct(c[0], c[2], z.f4); ct(c[1], c[3], z.f4);
    ct(c[0], c[1], z.f5a); ct(c[2], c[3], z.f5b);
    relayout<3, 4, NW>(c, xch, lane);
    ct(c[0], c[2], z.f6); ct(c[1], c[3], z.f6);
    ct(c[0], c[1], z.f7a); ct(c[2], c[3], z.f7b);
}

// Exact inverse transform INCLUDING the factor 1/256 (the reference's InvNTT returns R/256 times
// this, ntt.go:212-216, compensated by its R^-1-carrying MulHat).  In: layout L4, c < 2q.
// Out: layout L1, c < 2q.  No reduction between the layers: values double and stay below 512 q.
// FINAL is the last step's multiplier in mont32 form: INV256_R (default) for the exact inverse, INV256_RR when the input
// carries a factor 2^-32 (it came out of mont64 / mont32 on unscaled operands) that should disappear on the way.
constexpr uint32_t INV256_R = (uint32_t)((uint64_t)cpow(256, Q - 2) * R32 % Q);
constexpr uint32_t INV256_RR = (uint32_t)((uint64_t)INV256_R * R32 % Q);
template <uint32_t FINAL = INV256_R, bool NW = kNWDefault>
__device__ __forceinline__ void invntt(uint32_t (&c)[4], const LaneZetas &z, uint32_t *xch, int lane) {
    gs<2>(c[0], c[1], z.i7a); gs<2>(c[2], c[3], z.i7b);
    gs<4>(c[0], c[2], z.i6); gs<4>(c[1], c[3], z.i6);
    relayout<4, 3, NW>(c, xch, lane);
    gs<8>(c[0], c[1], z.i5a); gs<8>(c[2], c[3], z.i5b);
    gs<16>(c[0], c[2], z.i4); gs<16>(c[1], c[3], z.i4);
    relayout<3, 2, NW>(c, xch, lane);
    gs<32>(c[0], c[1], z.i3a); gs<32>(c[2], c[3], z.i3b);
    gs<64>(c[0], c[2], z.i2); gs<64>(c[1], c[3], z.i2);
    relayout<2, 1, NW>(c, xch, lane);
    const uint32_t z1 = zeta(1), z2 = zeta(2), z3 = zeta(3);
    gs<128>(c[0], c[1], z3); gs<128>(c[2], c[3], z2);
    gs<256>(c[0], c[2], z1); gs<256>(c[1], c[3], z1);
    // times 256^-1:  mont32(x, 2^32 / 256) = x / 256   (x < 512 q)
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = mont32(c[r], FINAL);
}
// ---- two polynomials at a time -------------------------------------------------------------------------------------------
// A transform is a chain of 4 register-local stages separated by 3 exchanges through LDS, and a lone wavefront waits out the
// full latency of every exchange (write, read back, ~100+ cycles) with nothing else to issue: the batch-signing kernels, one
// wavefront per attempt at 4 waves per SIMD, ran at 6.7 cycles per VALU instruction.  Two INDEPENDENT polynomials with an
// exchange buffer each go through the stages in lock step -- both buffers are written, then both are read -- so every
// wait covers two transforms.  Same arithmetic, same layouts and bounds as ntt / invntt above.
template <int FROM, int TO, bool NW = kNWDefault> __device__ __forceinline__ void relayout2(uint32_t (&c0)[4], uint32_t (&c1)[4], uint32_t *xch0, uint32_t *xch1, int lane) {
    auto pos = [&](int which, int r) {
        switch (which) {
        case 1: return lane + 4 * (lane >> 5) + 72 * r;
        case 2: return ((lane >> 4) * 72) + (lane & 15) + 16 * r + 4 * (r >> 1);
        case 3: return ((lane >> 2) << 4) + 4 * (lane >> 3) + (lane & 3) + 4 * r;
        default: return 4 * lane + 4 * (lane >> 3) + r;
        }
    };
    xch_sync<NW>();
#pragma unroll
    for (int r = 0; r < 4; r++) { xch0[pos(FROM, r)] = c0[r]; xch1[pos(FROM, r)] = c1[r]; }
    xch_sync<NW>();
#pragma unroll
    for (int r = 0; r < 4; r++) { c0[r] = xch0[pos(TO, r)]; c1[r] = xch1[pos(TO, r)]; }
}
template <bool NW = kNWDefault>
__device__ __forceinline__ void ntt2(uint32_t (&a)[4], uint32_t (&b)[4], const LaneZetas &z, uint32_t *xch0, uint32_t *xch1, int lane) {
    const uint32_t z1 = zeta(1), z2 = zeta(2), z3 = zeta(3);
    ct(a[0], a[2], z1); ct(a[1], a[3], z1); ct(b[0], b[2], z1); ct(b[1], b[3], z1);
    ct(a[0], a[1], z2); ct(a[2], a[3], z3); ct(b[0], b[1], z2); ct(b[2], b[3], z3);
    relayout2<1, 2, NW>(a, b, xch0, xch1, lane);
    ct(a[0], a[2], z.f2); ct(a[1], a[3], z.f2); ct(b[0], b[2], z.f2); ct(b[1], b[3], z.f2);
    ct(a[0], a[1], z.f3a); ct(a[2], a[3], z.f3b); ct(b[0], b[1], z.f3a); ct(b[2], b[3], z.f3b);
    relayout2<2, 3, NW>(a, b, xch0, xch1, lane);
    ct(a[0], a[2], z.f4); ct(a[1], a[3], z.f4); ct(b[0], b[2], z.f4); ct(b[1], b[3], z.f4);
    ct(a[0], a[1], z.f5a); ct(a[2], a[3], z.f5b); ct(b[0], b[1], z.f5a); ct(b[2], b[3], z.f5b);
    relayout2<3, 4, NW>(a, b, xch0, xch1, lane);
    ct(a[0], a[2], z.f6); ct(a[1], a[3], z.f6); ct(b[0], b[2], z.f6); ct(b[1], b[3], z.f6);
    ct(a[0], a[1], z.f7a); ct(a[2], a[3], z.f7b); ct(b[0], b[1], z.f7a); ct(b[2], b[3], z.f7b);
}
template <uint32_t FINAL = INV256_R, bool NW = kNWDefault>
__device__ __forceinline__ void invntt2(uint32_t (&a)[4], uint32_t (&b)[4], const LaneZetas &z, uint32_t *xch0, uint32_t *xch1, int lane) {
    gs<2>(a[0], a[1], z.i7a); gs<2>(a[2], a[3], z.i7b); gs<2>(b[0], b[1], z.i7a); gs<2>(b[2], b[3], z.i7b);
    gs<4>(a[0], a[2], z.i6); gs<4>(a[1], a[3], z.i6); gs<4>(b[0], b[2], z.i6); gs<4>(b[1], b[3], z.i6);
    relayout2<4, 3, NW>(a, b, xch0, xch1, lane);
    gs<8>(a[0], a[1], z.i5a); gs<8>(a[2], a[3], z.i5b); gs<8>(b[0], b[1], z.i5a); gs<8>(b[2], b[3], z.i5b);
    gs<16>(a[0], a[2], z.i4); gs<16>(a[1], a[3], z.i4); gs<16>(b[0], b[2], z.i4); gs<16>(b[1], b[3], z.i4);
    relayout2<3, 2, NW>(a, b, xch0, xch1, lane);
    gs<32>(a[0], a[1], z.i3a); gs<32>(a[2], a[3], z.i3b); gs<32>(b[0], b[1], z.i3a); gs<32>(b[2], b[3], z.i3b);
    gs<64>(a[0], a[2], z.i2); gs<64>(a[1], a[3], z.i2); gs<64>(b[0], b[2], z.i2); gs<64>(b[1], b[3], z.i2);
    relayout2<2, 1, NW>(a, b, xch0, xch1, lane);
    const uint32_t z1 = zeta(1), z2 = zeta(2), z3 = zeta(3);
    gs<128>(a[0], a[1], z3); gs<128>(a[2], a[3], z2); gs<128>(b[0], b[1], z3); gs<128>(b[2], b[3], z2);
    gs<256>(a[0], a[2], z1); gs<256>(a[1], a[3], z1); gs<256>(b[0], b[2], z1); gs<256>(b[1], b[3], z1);
#pragma unroll
    for (int r = 0; r < 4; r++) { a[r] = mont32(a[r], FINAL); b[r] = mont32(b[r], FINAL); }
}
#endif

// rounding.go:13-43 decompose for a in [0,q): returns a1, and a0+q through the reference's formula
template <uint32_t GAMMA2> CIRCL_HD void decompose(uint32_t a, uint32_t &a0plusq, uint32_t &a1) {
    constexpr uint32_t ALPHA = 2 * GAMMA2;
    a1 = (a + 127) >> 7;
    if constexpr (ALPHA == 523776) {
        a1 = (a1 * 1025 + (1u << 21)) >> 22;
        a1 &= 15;
    } else {
        a1 = (a1 * 11275 + (1u << 23)) >> 24;
        a1 ^= (uint32_t)((int32_t)(43 - a1) >> 31) & a1;
    }
    a0plusq = a - a1 * ALPHA;
    a0plusq += (uint32_t)((int32_t)(a0plusq - (Q - 1) / 2) >> 31) & Q;
}
// rounding.go:98-135 PolyUseHint for one coefficient
template <uint32_t GAMMA2> CIRCL_HD uint32_t use_hint(uint32_t a, uint32_t hint) {
    uint32_t a0, a1;
    decompose<GAMMA2>(a, a0, a1);
    if (hint == 0) return a1;
    if constexpr (GAMMA2 == 261888) {
        return a0 > Q ? (a1 + 1) & 15 : (a1 - 1) & 15;
    } else {
        if (a0 > Q) return a1 == 43 ? 0 : a1 + 1;
        return a1 == 0 ? 43 : a1 - 1;
    }
}
// rounding.go:56-70 makeHint for z0 = r0 - f in [0,q) and the unmodified high bits r1
template <uint32_t GAMMA2> CIRCL_HD bool make_hint(uint32_t z0, uint32_t r1) {
    return !(z0 <= GAMMA2 || z0 > Q - GAMMA2 || (z0 == Q - GAMMA2 && r1 == 0));
}
// field.go:35-52 power2round for a in [0,q): a = a1 2^D + a0 with -2^(D-1) < a0 <= 2^(D-1); returns a0 + q like the reference
CIRCL_HD void power2round(uint32_t a, uint32_t &a0plusq, uint32_t &a1) {
    uint32_t a0 = a & ((1u << D) - 1);
    a0 -= (1u << (D - 1)) + 1;
    a0 += (uint32_t)((int32_t)a0 >> 31) & (1u << D);
    a0 -= (1u << (D - 1)) - 1;   // now the signed a0 (two's complement)
    a1 = (a - a0) >> D;
    a0plusq = Q + a0;
}
// poly.go:51-71 exceeds for one coefficient x in [0,q)
CIRCL_HD bool exceeds(uint32_t x, uint32_t bound) {
    int32_t t = (int32_t)((Q - 1) / 2) - (int32_t)x;
    t ^= (t >> 31);
    t = (int32_t)((Q - 1) / 2) - t;
    return (uint32_t)t >= bound;
}

}  // namespace dilithium
}  // namespace circl
