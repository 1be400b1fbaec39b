// tests/hostsim/hostsim.hip -- TEST INFRASTRUCTURE: runs the lane-local __host__ __device__ functions
// of circl_amd/csrc/*_dev.h on the CPU (their host instantiation), so that the CPU-only test tier
// can check the very source the kernels are built from against the oracle.  Nothing here is linked
// into libcirclhip.so, and the product never calls a host instantiation.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "dilithium_dev.h"
#include "kyber_dev.h"
#include "lane_ops.h"
#include "x25519_dev.h"

using namespace circl;

extern "C" {

void hs_keccak_f1600(uint64_t *a, int rounds) {
    KeccakState s;
    for (int i = 0; i < 25; i++) { s.lo[i] = (uint32_t)a[i]; s.hi[i] = (uint32_t)(a[i] >> 32); }
    keccak_f1600(s, 24 - rounds);
    for (int i = 0; i < 25; i++) a[i] = ((uint64_t)s.hi[i] << 32) | s.lo[i];
}

int hs_kyber_barrett(int x) { return kyber::barrett(x); }
int hs_kyber_normalize(int x) { return kyber::normalize(x); }
unsigned hs_kyber_zeta_plain(int i) { return kyber::zeta_plain(i); }
// w b mod q through the two-instruction product by a constant; reduce32(t) = -t 2^-32 mod q
unsigned hs_kyber_mulc(unsigned b, unsigned w) { return kyber::mulc(b, kyber::mulc_const(w)); }
unsigned hs_kyber_reduce32(unsigned t) { return kyber::reduce32(t); }
unsigned hs_kyber_mulc_limit(void) { return kyber::MULC_LIMIT; }
unsigned hs_kyber_compress(int x, int d) {
    switch (d) {
    case 4: return kyber::compress_coeff<4>(x);
    case 5: return kyber::compress_coeff<5>(x);
    case 10: return kyber::compress_coeff<10>(x);
    default: return kyber::compress_coeff<11>(x);
    }
}
int hs_kyber_decompress(unsigned t, int d) {
    switch (d) {
    case 4: return kyber::decompress_coeff<4>(t);
    case 5: return kyber::decompress_coeff<5>(t);
    case 10: return kyber::decompress_coeff<10>(t);
    default: return kyber::decompress_coeff<11>(t);
    }
}
unsigned hs_kyber_msg_bit(int x) { return kyber::msg_bit(x); }
int hs_kyber_cbd2(unsigned t) { return kyber::cbd2_from_nibble(t); }
int hs_kyber_cbd3(unsigned t) { return kyber::cbd3_from_6bits(t); }
unsigned hs_kyber_cbd2_bias8_word(unsigned w) { return kyber::cbd2_bias8_word(w); }
// one lane's share of MulHat: a[4] in [0,q), b[4] < 2^15 are coefficients 4l..4l+3; returns reduce32 of the four lazy sums
void hs_kyber_mulhat4_packed(int *out, const int *a, const int *b, int lane) {
    int acc[4] = {0, 0, 0, 0}, y[4];
    for (int i = 0; i < 4; i++) y[i] = b[i];
    const kyber::HatOperand op = kyber::hat_prepare(y, kyber::zeta_c(64 + lane), kyber::zeta_cn(64 + lane));
    kyber::mulhat_acc_packed(acc, kyber::pack16(a[0], a[1]), kyber::pack16(a[2], a[3]), op);
    kyber::mulhat_finish(acc);
    for (int i = 0; i < 4; i++) out[i] = acc[i];
}
// the 4-register butterfly network of the wave-level NTT, executed lane by lane on the host: exactly the
// layer / zeta schedule of kyber::ntt / invntt<scale> with the LDS exchanges replaced by array indexing
// (including their element widths).  p holds values in [0, bound); *maxval receives the largest intermediate.
void hs_kyber_ntt(uint32_t *p, int inverse, uint32_t scale, uint32_t *maxval) {
    int c[64][4];
    uint32_t mx = 0;
    auto idx = [](int which, int l, int r) { return which == 1 ? kyber::idx_l1(l, r) : which == 2 ? kyber::idx_l2(l, r) : which == 3 ? kyber::idx_l3(l, r) : kyber::idx_l4(l, r); };
    auto track = [&]() { for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) { if (c[l][r] < 0) mx = 0xffffffffu; else if ((uint32_t)c[l][r] > mx) mx = (uint32_t)c[l][r]; } };
    auto relayout = [&](int from, int to, bool wide) {
        uint32_t x[256];
        track();
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) x[idx(from, l, r)] = wide ? (uint32_t)c[l][r] : (uint32_t)(uint16_t)c[l][r];
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) c[l][r] = (int)x[idx(to, l, r)];
    };
    const uint32_t z1 = kyber::zeta_c(1), z2 = kyber::zeta_c(2), z3 = kyber::zeta_c(3);
    if (!inverse) {
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) c[l][r] = (int)p[idx(1, l, r)];
        for (int l = 0; l < 64; l++) { int *v = c[l]; kyber::ct(v[0], v[2], z1); kyber::ct(v[1], v[3], z1); kyber::ct(v[0], v[1], z2); kyber::ct(v[2], v[3], z3); }
        relayout(1, 2, false);
        for (int l = 0; l < 64; l++) { int *v = c[l]; auto z = kyber::load_lane_zetas(l); kyber::ct(v[0], v[2], z.f2); kyber::ct(v[1], v[3], z.f2); kyber::ct(v[0], v[1], z.f3a); kyber::ct(v[2], v[3], z.f3b); }
        relayout(2, 3, false);
        for (int l = 0; l < 64; l++) { int *v = c[l]; auto z = kyber::load_lane_zetas(l); kyber::ct(v[0], v[2], z.f4); kyber::ct(v[1], v[3], z.f4); kyber::ct(v[0], v[1], z.f5a); kyber::ct(v[2], v[3], z.f5b); }
        relayout(3, 4, false);
        for (int l = 0; l < 64; l++) { int *v = c[l]; auto z = kyber::load_lane_zetas(l); kyber::ct(v[0], v[2], z.f6); kyber::ct(v[1], v[3], z.f6); }
        track();
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) p[idx(4, l, r)] = (uint32_t)c[l][r];
    } else {
        constexpr int Q = kyber::Q;
        const uint32_t fin = kyber::mulc_const((uint32_t)((uint64_t)(scale % Q) * kyber::invq(128) % Q));
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) c[l][r] = (int)p[idx(4, l, r)];
        for (int l = 0; l < 64; l++) { int *v = c[l]; auto z = kyber::load_lane_zetas(l); kyber::gs<Q>(v[0], v[2], z.i6); kyber::gs<Q>(v[1], v[3], z.i6); }
        relayout(4, 3, false);
        for (int l = 0; l < 64; l++) { int *v = c[l]; auto z = kyber::load_lane_zetas(l); kyber::gs<2 * Q>(v[0], v[1], z.i5a); kyber::gs<2 * Q>(v[2], v[3], z.i5b); kyber::gs<4 * Q>(v[0], v[2], z.i4); kyber::gs<4 * Q>(v[1], v[3], z.i4); }
        relayout(3, 2, false);
        for (int l = 0; l < 64; l++) { int *v = c[l]; auto z = kyber::load_lane_zetas(l); kyber::gs<8 * Q>(v[0], v[1], z.i3a); kyber::gs<8 * Q>(v[2], v[3], z.i3b); kyber::gs<16 * Q>(v[0], v[2], z.i2); kyber::gs<16 * Q>(v[1], v[3], z.i2); }
        relayout(2, 1, true);
        for (int l = 0; l < 64; l++) { int *v = c[l]; kyber::gs<32 * Q>(v[0], v[1], z3); kyber::gs<32 * Q>(v[2], v[3], z2); kyber::gs<64 * Q>(v[0], v[2], z1); kyber::gs<64 * Q>(v[1], v[3], z1); }
        track();
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) p[idx(1, l, r)] = kyber::mulc((uint32_t)c[l][r], fin);
    }
    *maxval = mx;
}

uint32_t hs_dil_mont32(uint32_t a, uint32_t b) { return dilithium::mont32(a, b); }
uint32_t hs_dil_mont64(uint64_t t) { return dilithium::mont64(t); }
uint32_t hs_dil_fold(uint32_t x) { return dilithium::fold(x); }
uint32_t hs_dil_normalize(uint32_t x) { return dilithium::normalize(x); }
uint32_t hs_dil_zeta(int i) { return dilithium::zeta(i); }
uint32_t hs_dil_r32(void) { return dilithium::R32; }
uint32_t hs_dil_use_hint(uint32_t a, uint32_t h, int gamma2_is_88) {
    return gamma2_is_88 ? dilithium::use_hint<95232>(a, h) : dilithium::use_hint<261888>(a, h);
}
void hs_dil_decompose(uint32_t a, int gamma2_is_88, uint32_t *a0, uint32_t *a1) {
    if (gamma2_is_88) dilithium::decompose<95232>(a, *a0, *a1);
    else dilithium::decompose<261888>(a, *a0, *a1);
}
int hs_dil_exceeds(uint32_t x, uint32_t bound) { return dilithium::exceeds(x, bound) ? 1 : 0; }


// ---- x25519_dev.h: limb arithmetic on raw (possibly un-carried) limbs, and the whole scalar multiplication ----
void hs_fe_mul(uint32_t *r, const uint32_t *f, const uint32_t *g) {
    x25519::Fe a, b;
    for (int i = 0; i < 10; i++) { a.v[i] = f[i]; b.v[i] = g[i]; }
    const x25519::Fe c = x25519::fe_mul(a, b);
    for (int i = 0; i < 10; i++) r[i] = c.v[i];
}
void hs_fe_sqr(uint32_t *r, const uint32_t *f) {
    x25519::Fe a;
    for (int i = 0; i < 10; i++) a.v[i] = f[i];
    const x25519::Fe c = x25519::fe_sqr(a);
    for (int i = 0; i < 10; i++) r[i] = c.v[i];
}
void hs_fe_mul_small(uint32_t *r, const uint32_t *f, uint32_t k) {
    x25519::Fe a;
    for (int i = 0; i < 10; i++) a.v[i] = f[i];
    const x25519::Fe c = x25519::fe_mul_small(a, k);
    for (int i = 0; i < 10; i++) r[i] = c.v[i];
}
void hs_fe_sub(uint32_t *r, const uint32_t *f, const uint32_t *g) {
    x25519::Fe a, b;
    for (int i = 0; i < 10; i++) { a.v[i] = f[i]; b.v[i] = g[i]; }
    const x25519::Fe c = x25519::fe_sub(a, b);
    for (int i = 0; i < 10; i++) r[i] = c.v[i];
}
void hs_fe_inv(uint32_t *r, const uint32_t *f) {
    x25519::Fe a;
    for (int i = 0; i < 10; i++) a.v[i] = f[i];
    const x25519::Fe c = x25519::fe_inv(a);
    for (int i = 0; i < 10; i++) r[i] = c.v[i];
}
void hs_fe_from_words(uint32_t *r, const uint32_t *w) {
    const x25519::Fe c = x25519::fe_from_words(w);
    for (int i = 0; i < 10; i++) r[i] = c.v[i];
}
void hs_fe_to_words(uint32_t *w, const uint32_t *f) {
    x25519::Fe a;
    for (int i = 0; i < 10; i++) a.v[i] = f[i];
    x25519::fe_to_words(w, a);
}
uint32_t hs_x25519_valid_public(const uint32_t *u) {
    uint32_t w[8];
    for (int i = 0; i < 8; i++) w[i] = u[i];
    w[7] &= 0x7fffffffu;
    return x25519::valid_public(w);
}
void hs_x25519(uint32_t *out, const uint32_t *k, const uint32_t *u, int base, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (base == 2) x25519::base_mult(out + 8 * i, k + 8 * i);  // the fixed-base comb
        else if (base) x25519::scalar_mult<true>(out + 8 * i, k + 8 * i, k + 8 * i);
        else x25519::scalar_mult<false>(out + 8 * i, k + 8 * i, u + 8 * i);
    }
}

// the switch behind circl_hip_lane_op, host instantiation, elementwise
void hs_lane_op_array(int op, int arg, const uint32_t *a, const uint32_t *b, uint32_t *out0, uint32_t *out1, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint32_t r0, r1;
        prim::lane_op_eval(op, arg, a[i], b ? b[i] : 0u, r0, r1);
        out0[i] = r0;
        if (out1) out1[i] = r1;
    }
}

}  // extern "C"
