// api_hybrid.hip -- the hybrid KEMs that carry ML-KEM-768 today, composed on the device behind the C ABI
// (include/circl_hip.h):
//   CIRCL_HIP_HYBRID_XWING            kem/xwing/xwing.go      (X25519 + ML-KEM-768, SHA3-256 combiner)
//   CIRCL_HIP_HYBRID_X25519MLKEM768   kem/hybrid/hybrid.go    (ct_M || ct_X, ss_M || ss_X; X25519 as a KEM: xkem.go)
//   CIRCL_HIP_HYBRID_KYBER768_X25519, _KYBER512_X25519   the same scheme type with X25519 first and round-3 Kyber second
// A call is a handful of launches per chunk over HBM-resident arrays: strided splits of the packed keys / ciphertexts,
// seed expansion (SHAKE256, lane = item), the ML-KEM-768 batch kernels, two X25519 ladders per item (lane = item), the
// combiner, strided joins.  Nothing is computed on the host.
#include "host_common.h"
#include "hybrid_kernels.h"
#include "keytable.h"

using namespace circl::host;
using circl::KeyIdx;
namespace hk = circl::hybridk;

namespace {
// one hybrid = one lattice KEM (ML-KEM or round-3 Kyber) + X25519, either of them first in every packed array
struct Desc {
    bool xwing;     // X-Wing: seed expansion, combiner and private-key format of kem/xwing; otherwise kem/hybrid's concatenation scheme
    int param;      // 768 | 512
    bool r3;        // round-3 Kyber (kem/kyber) instead of ML-KEM
    bool x_first;   // X25519 is the `first` component of hybrid.go's scheme{name, first, second}
    size_t EK, DK, CTM;
    size_t seed, eseed, pk, sk, ct, ss;
    // byte offsets of the two halves inside a packed row
    size_t kem_off(size_t x_bytes) const { return x_first ? x_bytes : 0; }
    size_t x_off(size_t kem_bytes) const { return x_first ? 0 : kem_bytes; }
};
bool desc_of(int scheme, Desc &d) {
    auto lattice = [&](int param) {
        d.param = param;
        d.EK = circl_hip_mlkem_ek_size(param);
        d.DK = circl_hip_mlkem_dk_size(param);
        d.CTM = circl_hip_mlkem_ct_size(param);
        d.pk = d.EK + 32;
        d.ct = d.CTM + 32;
    };
    switch (scheme) {
    case CIRCL_HIP_HYBRID_XWING:
        d.xwing = true; d.r3 = false; d.x_first = false;
        lattice(768);
        d.seed = 32; d.eseed = 64; d.sk = 32; d.ss = 32;
        return true;
    case CIRCL_HIP_HYBRID_X25519MLKEM768:  // scheme{"X25519MLKEM768", mlkem768, x25519Kem}: hybrid.go:95-99
        d.xwing = false; d.r3 = false; d.x_first = false;
        lattice(768);
        break;
    case CIRCL_HIP_HYBRID_KYBER768_X25519:  // scheme{"Kyber768-X25519", x25519Kem, kyber768}: hybrid.go:77-81
        d.xwing = false; d.r3 = true; d.x_first = true;
        lattice(768);
        break;
    case CIRCL_HIP_HYBRID_KYBER512_X25519:  // scheme{"Kyber512-X25519", x25519Kem, kyber512}: hybrid.go:71-75
        d.xwing = false; d.r3 = true; d.x_first = true;
        lattice(512);
        break;
    default:
        return false;
    }
    d.seed = 64;   // max of the components' seed sizes (hybrid.go:128-138): 64 for the lattice KEM, 32 for X25519
    d.eseed = 32;  // hybrid.go:148-157
    d.sk = d.DK + 32;
    d.ss = 64;
    return true;
}

// temporaries of one call, carved from the caller's workspace in front of the lattice KEM's workspace
struct Carve {
    uint8_t *p;
    uint8_t *take(size_t bytes) {
        uint8_t *r = p;
        p += (bytes + 255) & ~size_t(255);
        return r;
    }
};
size_t tmp_bytes(const Desc &d, size_t n) {  // upper bound over the three operations: ek, dk, ctm + a dozen 32/64-byte rows per item
    auto r = [](size_t b) { return (b + 255) & ~size_t(255); };
    return r(n * d.EK) + r(n * d.DK) + r(n * d.CTM) + 10 * r(n * 64) + r(n);
}

int copy_rows(hipStream_t st, void *dst, size_t dst_row, const void *src, size_t src_row, size_t width, size_t n) {
    const size_t total = n * (width / 4);
    if (total == 0) return CIRCL_HIP_OK;
    hipLaunchKernelGGL(hk::rows_copy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, static_cast<uint32_t *>(dst), dst_row / 4,
                       static_cast<const uint32_t *>(src), src_row / 4, (unsigned)(width / 4), n);
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}
int zero_failed(hipStream_t st, void *dst, size_t row, const uint8_t *status, size_t n) {
    const size_t total = n * (row / 4);
    hipLaunchKernelGGL(hk::rows_zero_failed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, static_cast<uint32_t *>(dst), row / 4,
                       (unsigned)(row / 4), status, n);
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}
#define TRY(expr)                          \
    do {                                   \
        const int rc_ = (expr);            \
        if (rc_ != CIRCL_HIP_OK) return rc_; \
    } while (0)
inline const uint32_t *w(const uint8_t *p) { return reinterpret_cast<const uint32_t *>(p); }
inline uint32_t *w(uint8_t *p) { return reinterpret_cast<uint32_t *>(p); }
inline dim3 g256(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

// The secret temporaries of a call, zeroed on the call's stream on EVERY path: finish() on the way out of a successful call
// (its failures are reported), the destructor behind any early error return.
struct SecretWipe {
    hipStream_t st;
    struct R { uint8_t *p; size_t bytes; };
    std::vector<R> regions;
    bool done = false;
    void add(uint8_t *p, size_t bytes) { regions.push_back({p, bytes}); }
    int finish() {
        done = true;
        for (auto &r : regions) HIP_TRY(hipMemsetAsync(r.p, 0, r.bytes, st));
        return CIRCL_HIP_OK;
    }
    ~SecretWipe() {
        if (done) return;
        for (auto &r : regions) (void)hipMemsetAsync(r.p, 0, r.bytes, st);
        (void)hipGetLastError();
    }
};

bool args_ok(const void *a, const void *b, const void *c, const void *d, const void *ws) {
    return !((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(d) |
              reinterpret_cast<uintptr_t>(ws)) & 15);
}

int misaligned() {  // (the header's code for a misaligned device pointer; the message tells it from a short workspace)
    g_err = "hybrid _dev entry point: device pointers must be 16-byte aligned";
    return CIRCL_HIP_EWORKSPACE;
}

// the lattice half: ML-KEM (per-item status) or round-3 Kyber (no per-item failure: status = 0)
int kem_keygen(const Desc &d, const uint8_t *seed64, uint8_t *ek, uint8_t *dk, size_t n, void *ws, size_t wsb, hipStream_t st) {
    return d.r3 ? circl_hip_kyber_keygen_dev(d.param, seed64, ek, dk, n, ws, wsb, st) : circl_hip_mlkem_keygen_dev(d.param, seed64, ek, dk, n, ws, wsb, st);
}
int kem_encaps(const Desc &d, const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, void *ws, size_t wsb,
               hipStream_t st) {
    if (!d.r3) return circl_hip_mlkem_encaps_dev(d.param, ek, m, ct, ss, status, n, ws, wsb, st);
    HIP_TRY(hipMemsetAsync(status, 0, n, st));
    return circl_hip_kyber_encaps_dev(d.param, ek, m, ct, ss, n, ws, wsb, st);
}
int kem_decaps(const Desc &d, const uint8_t *dk, const uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, void *ws, size_t wsb, hipStream_t st) {
    if (!d.r3) return circl_hip_mlkem_decaps_dev(d.param, dk, ct, ss, status, n, ws, wsb, st);
    HIP_TRY(hipMemsetAsync(status, 0, n, st));
    return circl_hip_kyber_decaps_dev(d.param, dk, ct, ss, n, ws, wsb, st);
}
}  // namespace

extern "C" {

size_t circl_hip_hybrid_seed_size(int scheme) { Desc d; return desc_of(scheme, d) ? d.seed : 0; }
size_t circl_hip_hybrid_eseed_size(int scheme) { Desc d; return desc_of(scheme, d) ? d.eseed : 0; }
size_t circl_hip_hybrid_pk_size(int scheme) { Desc d; return desc_of(scheme, d) ? d.pk : 0; }
size_t circl_hip_hybrid_sk_size(int scheme) { Desc d; return desc_of(scheme, d) ? d.sk : 0; }
size_t circl_hip_hybrid_ct_size(int scheme) { Desc d; return desc_of(scheme, d) ? d.ct : 0; }
size_t circl_hip_hybrid_ss_size(int scheme) { Desc d; return desc_of(scheme, d) ? d.ss : 0; }

size_t circl_hip_hybrid_workspace_size(int scheme, size_t n) {
    Desc d;
    if (!desc_of(scheme, d)) return 0;
    return tmp_bytes(d, n) + circl_hip_mlkem_workspace_size(d.param, n);
}
// what a call cannot do without: the ML-KEM half then takes its scratch routes (no row cache: 256 MB less for a chunk of 2^16 items).
// The _dev entry points accept this much; circl_hip_hybrid_workspace_size stays what callers are told to bring.
static size_t hybrid_ws_min(const Desc &d, size_t n) { return tmp_bytes(d, n) + mlkem_ws_min_bytes(n); }

int circl_hip_hybrid_keygen_dev(int scheme, const uint8_t *d_seed, uint8_t *d_pk, uint8_t *d_sk, size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    Desc d;
    if (!desc_of(scheme, d)) return CIRCL_HIP_EPARAM;
    if (ndev() <= 0) return CIRCL_HIP_ENODEV;
    if (n == 0) return CIRCL_HIP_OK;
    if (!d_seed || !d_pk || !d_sk || !d_ws) return CIRCL_HIP_EPARAM;
    if (ws_bytes < hybrid_ws_min(d, n)) return CIRCL_HIP_EWORKSPACE;
    if (!args_ok(d_seed, d_pk, d_sk, nullptr, d_ws)) return misaligned();
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carve c{static_cast<uint8_t *>(d_ws)};
    uint8_t *seedm = c.take(n * 64), *skx = c.take(n * 32), *pkx = c.take(n * 32), *ek = c.take(n * d.EK), *dk = c.take(n * d.DK);
    uint8_t *kws = static_cast<uint8_t *>(d_ws) + tmp_bytes(d, n);
    const size_t kws_bytes = ws_bytes - tmp_bytes(d, n);
    SecretWipe wipe{st};  // nothing key-equivalent stays behind in the caller's workspace
    wipe.add(seedm, n * 64); wipe.add(skx, n * 32); wipe.add(dk, n * d.DK);
    if (d.xwing)
        hipLaunchKernelGGL(hk::xwing_expand_kernel, g256(n), dim3(256), 0, st, w(d_seed), w(seedm), w(skx), n);
    else if (d.x_first)
        hipLaunchKernelGGL((hk::hybrid_expand_kernel<8, 8, true>), g256(n), dim3(256), 0, st, w(d_seed), w(seedm), w(skx), n);
    else
        hipLaunchKernelGGL((hk::hybrid_expand_kernel<8, 8, false>), g256(n), dim3(256), 0, st, w(d_seed), w(seedm), w(skx), n);
    HIP_TRY(hipGetLastError());
    TRY(kem_keygen(d, seedm, ek, dk, n, kws, kws_bytes, st));
    TRY(circl_hip_x25519_dev(skx, nullptr, pkx, nullptr, n, st));
    TRY(copy_rows(st, d_pk + d.kem_off(32), d.pk, ek, d.EK, d.EK, n));
    TRY(copy_rows(st, d_pk + d.x_off(d.EK), d.pk, pkx, 32, 32, n));
    if (d.xwing) {
        TRY(copy_rows(st, d_sk, 32, d_seed, 32, 32, n));  // the packed private key is the seed (xwing.go:156-163)
    } else {
        TRY(copy_rows(st, d_sk + d.kem_off(32), d.sk, dk, d.DK, d.DK, n));
        TRY(copy_rows(st, d_sk + d.x_off(d.DK), d.sk, skx, 32, 32, n));
    }
    return wipe.finish();
}

int circl_hip_hybrid_encaps_dev(int scheme, const uint8_t *d_pk, const uint8_t *d_eseed, uint8_t *d_ct, uint8_t *d_ss, uint8_t *d_status, size_t n,
                                void *d_ws, size_t ws_bytes, void *stream) {
    Desc d;
    if (!desc_of(scheme, d)) return CIRCL_HIP_EPARAM;
    if (ndev() <= 0) return CIRCL_HIP_ENODEV;
    if (n == 0) return CIRCL_HIP_OK;
    if (!d_status || !d_pk || !d_eseed || !d_ct || !d_ss || !d_ws) return CIRCL_HIP_EPARAM;
    if (ws_bytes < hybrid_ws_min(d, n)) return CIRCL_HIP_EWORKSPACE;
    if (!args_ok(d_pk, d_eseed, d_ct, d_ss, d_ws)) return misaligned();
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carve c{static_cast<uint8_t *>(d_ws)};
    uint8_t *ek = c.take(n * d.EK), *pkx = c.take(n * 32), *m = c.take(n * 32), *ekx = c.take(n * 32), *ctm = c.take(n * d.CTM), *ssm = c.take(n * 32),
            *ctx = c.take(n * 32), *ssx = c.take(n * 32), *okx = c.take(n);
    uint8_t *kws = static_cast<uint8_t *>(d_ws) + tmp_bytes(d, n);
    const size_t kws_bytes = ws_bytes - tmp_bytes(d, n);
    SecretWipe wipe{st};  // the ephemeral secrets and the two half shared secrets
    wipe.add(m, n * 32); wipe.add(ekx, n * 32); wipe.add(ssm, n * 32); wipe.add(ssx, n * 32);
    TRY(copy_rows(st, ek, d.EK, d_pk + d.kem_off(32), d.pk, d.EK, n));
    TRY(copy_rows(st, pkx, 32, d_pk + d.x_off(d.EK), d.pk, 32, n));
    if (d.xwing) {  // xwing.go:247-248: seedm = seed[:32], ekx = seed[32:]
        TRY(copy_rows(st, m, 32, d_eseed, 64, 32, n));
        TRY(copy_rows(st, ekx, 32, d_eseed + 32, 64, 32, n));
    } else {
        if (d.x_first)
            hipLaunchKernelGGL((hk::hybrid_expand_kernel<4, 4, true>), g256(n), dim3(256), 0, st, w(d_eseed), w(m), w(ekx), n);
        else
            hipLaunchKernelGGL((hk::hybrid_expand_kernel<4, 4, false>), g256(n), dim3(256), 0, st, w(d_eseed), w(m), w(ekx), n);
        HIP_TRY(hipGetLastError());
    }
    TRY(kem_encaps(d, ek, m, ctm, ssm, d_status, n, kws, kws_bytes, st));
    TRY(x25519_pair_dev(ekx, pkx, ctx, ssx, okx, n, st));  // ct_X = X25519(ekx, 9), ss_X = X25519(ekx, pk_X)
    TRY(copy_rows(st, d_ct + d.kem_off(32), d.ct, ctm, d.CTM, d.CTM, n));
    TRY(copy_rows(st, d_ct + d.x_off(d.CTM), d.ct, ctx, 32, 32, n));
    if (d.xwing) {  // a low-order pk_X is not an error in X-Wing (xwing.go:251-254)
        hipLaunchKernelGGL(hk::xwing_combine_kernel, g256(n), dim3(256), 0, st, w(ssm), w(ssx), w(ctx), w(pkx), d_status, w(d_ss), n);
        HIP_TRY(hipGetLastError());
    } else {
        hipLaunchKernelGGL(hk::hybrid_status_kernel, g256(n), dim3(256), 0, st, d_status, okx, n);
        HIP_TRY(hipGetLastError());
        TRY(copy_rows(st, d_ss + d.kem_off(32), 64, ssm, 32, 32, n));
        TRY(copy_rows(st, d_ss + d.x_off(32), 64, ssx, 32, 32, n));
        TRY(zero_failed(st, d_ss, 64, d_status, n));
    }
    TRY(zero_failed(st, d_ct, d.ct, d_status, n));
    return wipe.finish();
}

int circl_hip_hybrid_decaps_dev(int scheme, const uint8_t *d_sk, const uint8_t *d_ct, uint8_t *d_ss, uint8_t *d_status, size_t n, void *d_ws,
                                size_t ws_bytes, void *stream) {
    Desc d;
    if (!desc_of(scheme, d)) return CIRCL_HIP_EPARAM;
    if (ndev() <= 0) return CIRCL_HIP_ENODEV;
    if (n == 0) return CIRCL_HIP_OK;
    if (!d_status || !d_sk || !d_ct || !d_ss || !d_ws) return CIRCL_HIP_EPARAM;
    if (ws_bytes < hybrid_ws_min(d, n)) return CIRCL_HIP_EWORKSPACE;
    if (!args_ok(d_sk, d_ct, d_ss, nullptr, d_ws)) return misaligned();
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carve c{static_cast<uint8_t *>(d_ws)};
    uint8_t *dk = c.take(n * d.DK), *ek = c.take(n * d.EK), *skx = c.take(n * 32), *ctm = c.take(n * d.CTM), *ctx = c.take(n * 32), *ssm = c.take(n * 32),
            *ssx = c.take(n * 32), *pkx = c.take(n * 32), *seedm = c.take(n * 64), *okx = c.take(n);
    uint8_t *kws = static_cast<uint8_t *>(d_ws) + tmp_bytes(d, n);
    const size_t kws_bytes = ws_bytes - tmp_bytes(d, n);
    SecretWipe wipe{st};
    wipe.add(dk, n * d.DK); wipe.add(skx, n * 32); wipe.add(ssm, n * 32); wipe.add(ssx, n * 32); wipe.add(seedm, n * 64);
    TRY(copy_rows(st, ctm, d.CTM, d_ct + d.kem_off(32), d.ct, d.CTM, n));
    TRY(copy_rows(st, ctx, 32, d_ct + d.x_off(d.CTM), d.ct, 32, n));
    if (d.xwing) {  // the private key is the seed: re-derive (xwing.go:165-185 Unpack = deriveKeyPair)
        hipLaunchKernelGGL(hk::xwing_expand_kernel, g256(n), dim3(256), 0, st, w(d_sk), w(seedm), w(skx), n);
        HIP_TRY(hipGetLastError());
        TRY(kem_keygen(d, seedm, ek, dk, n, kws, kws_bytes, st));
        TRY(x25519_pair_dev(skx, ctx, pkx, ssx, okx, n, st));  // sk.xpk = X25519(sk_X, 9), ss_X = X25519(sk_X, ct_X)
    } else {
        TRY(copy_rows(st, dk, d.DK, d_sk + d.kem_off(32), d.sk, d.DK, n));
        TRY(copy_rows(st, skx, 32, d_sk + d.x_off(d.DK), d.sk, 32, n));
        TRY(circl_hip_x25519_dev(skx, ctx, ssx, okx, n, st));
    }
    TRY(kem_decaps(d, dk, ctm, ssm, d_status, n, kws, kws_bytes, st));
    if (d.xwing) {
        hipLaunchKernelGGL(hk::xwing_combine_kernel, g256(n), dim3(256), 0, st, w(ssm), w(ssx), w(ctx), w(pkx), static_cast<const uint8_t *>(nullptr),
                           w(d_ss), n);
        HIP_TRY(hipGetLastError());
    } else {
        hipLaunchKernelGGL(hk::hybrid_status_kernel, g256(n), dim3(256), 0, st, d_status, okx, n);
        HIP_TRY(hipGetLastError());
        TRY(copy_rows(st, d_ss + d.kem_off(32), 64, ssm, 32, 32, n));
        TRY(copy_rows(st, d_ss + d.x_off(32), 64, ssx, 32, 32, n));
        TRY(zero_failed(st, d_ss, 64, d_status, n));
    }
    return wipe.finish();
}

// ---- hybrid key tables that live across calls (keytable.h) ------------------------------------------------------------------
// kem/xwing's PrivateKey keeps the expanded ML-KEM-768 key, the X25519 scalar and its public point next to the 32-byte seed
// (xwing.go:20-25), its PublicKey the parsed ML-KEM key (:28-31); kem/hybrid's keys hold the component schemes' parsed keys
// (hybrid.go:101-114).  A hybrid table is that: an ML-KEM key table (A^T, H(ek), the private key's hash verdict) of the lattice
// halves plus the X25519 rows, built once; a call then moves only seeds / ciphertexts and runs the shared-key ML-KEM work.
static int gather_rows(hipStream_t st, uint8_t *dst, const uint8_t *table, size_t nkeys, const uint32_t *key_idx, size_t n) {
    hipLaunchKernelGGL(hk::rows_gather_kernel, g256(n * 8), dim3(256), 0, st, w(dst), w(table), KeyIdx{key_idx, (uint32_t)(nkeys - 1)}, 8u, n);
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}
static bool hybrid_table_ok(const circl_hip_keytable *t, int want_private) {
    return t && t->magic == kKeytableMagic && t->family == 3 && t->private_keys == want_private && t->inner && t->d_x;
}
static int hybrid_keytable_new_one(int scheme, const Desc &d, int private_keys, const uint8_t *keys, size_t nkeys, int device, uint8_t *key_status,
                                   circl_hip_keytable **out) {
    HIP_TRY(hipSetDevice(physical_device(device)));
    hipStream_t h2d = nullptr, d2h = nullptr, st = nullptr;
    TRY(pipeline_streams(device, &h2d, &d2h, &st));
    circl_hip_keytable *t = new (std::nothrow) circl_hip_keytable();
    if (!t) return CIRCL_HIP_ENOMEM;
    t->magic = kKeytableMagic; t->family = 3; t->param = d.param; t->device = device; t->private_keys = private_keys ? 1 : 0; t->nkeys = nkeys;
    t->scheme = scheme;
    t->row = private_keys ? d.sk : d.pk;
    t->x_bytes = up256(nkeys * 64);
    struct Tmp {  // scratch of the build, freed (private material wiped) on every path
        uint8_t *p = nullptr; size_t bytes = 0;
        ~Tmp() { if (p) { (void)hipMemset(p, 0, bytes); (void)hipFree(p); } }
    } tmp;
    // the lattice halves, contiguous on the host, for the inner ML-KEM table
    const size_t KROW = private_keys ? d.DK : d.EK;
    std::vector<uint8_t> krows, xrows(nkeys * 32);
    // X-Wing private keys: the expanded decapsulation keys come back from the device into PAGE-LOCKED memory of the library's own (a
    // copy into pageable memory would bounce through the runtime's staging, which nobody wipes) and go from there into the inner table
    struct Pinned {
        uint8_t *p = nullptr; size_t bytes = 0;
        ~Pinned() {
            if (!p) return;
            volatile uint8_t *z = p;
            for (size_t i = 0; i < bytes; i++) z[i] = 0;
            (void)hipHostFree(p);
        }
    } kpin;
    int rc = CIRCL_HIP_OK;
    if (hipMalloc(reinterpret_cast<void **>(&t->d_x), t->x_bytes) != hipSuccess) { (void)hipGetLastError(); rc = CIRCL_HIP_ENOMEM; }
    if (rc == CIRCL_HIP_OK && private_keys && d.xwing) {
        // the packed private key is the 32-byte seed: expand it on the device as every X-Wing decapsulation would (xwing.go:98-144)
        const size_t ws_bytes = circl_hip_mlkem_workspace_size(d.param, nkeys);
        auto r = [](size_t b) { return (b + 255) & ~size_t(255); };
        tmp.bytes = r(nkeys * 32) + r(nkeys * 64) + r(nkeys * d.EK) + r(nkeys * d.DK) + ws_bytes;
        if (hipMalloc(reinterpret_cast<void **>(&tmp.p), tmp.bytes) != hipSuccess) { (void)hipGetLastError(); rc = CIRCL_HIP_ENOMEM; }
        if (rc == CIRCL_HIP_OK) {
            Carve c{tmp.p};
            uint8_t *seed = c.take(nkeys * 32), *seedm = c.take(nkeys * 64), *ek = c.take(nkeys * d.EK), *dk = c.take(nkeys * d.DK), *kws = c.p;
            uint8_t *skx = t->d_x, *pkx = t->d_x + nkeys * 32;
            kpin.bytes = nkeys * d.DK;
            if (hipHostMalloc(reinterpret_cast<void **>(&kpin.p), kpin.bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); kpin.p = nullptr; rc = CIRCL_HIP_ENOMEM; }
            auto build = [&]() -> int {
                TRY(upload_secret(seed, keys, nkeys * 32, st));  // (the seeds ARE the private keys)
                hipLaunchKernelGGL(hk::xwing_expand_kernel, g256(nkeys), dim3(256), 0, st, w(seed), w(seedm), w(skx), nkeys);
                HIP_TRY(hipGetLastError());
                TRY(kem_keygen(d, seedm, ek, dk, nkeys, kws, ws_bytes, st));
                TRY(circl_hip_x25519_dev(skx, nullptr, pkx, nullptr, nkeys, st));
                HIP_TRY(hipMemcpyAsync(kpin.p, dk, nkeys * d.DK, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                return CIRCL_HIP_OK;
            };
            if (rc == CIRCL_HIP_OK) rc = build();
        }
    } else if (rc == CIRCL_HIP_OK) {
        krows.resize(nkeys * KROW);
        for (size_t i = 0; i < nkeys; i++) {
            const uint8_t *row = keys + i * t->row;
            memcpy(krows.data() + i * KROW, row + d.kem_off(32), KROW);
            memcpy(xrows.data() + i * 32, row + d.x_off(KROW), 32);
        }
        // public: pk_X rows; private (kem/hybrid): sk_X rows (the decapsulation never needs pk_X there)
        if (private_keys) rc = upload_secret(t->d_x, xrows.data(), nkeys * 32, st);
        else if (hipMemcpyAsync(t->d_x, xrows.data(), nkeys * 32, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            (void)hipGetLastError();
            rc = CIRCL_HIP_EHIP;
        }
    }
    if (rc == CIRCL_HIP_OK) rc = circl_hip_mlkem_keytable_new(d.param, private_keys ? 1 : 0, kpin.p ? kpin.p : krows.data(), nkeys, device, key_status, &t->inner);
    if (private_keys) {  // the host copies of the private rows do not outlive the build
        volatile uint8_t *z = krows.data();
        for (size_t i = 0; i < krows.size(); i++) z[i] = 0;
        volatile uint8_t *x = xrows.data();
        for (size_t i = 0; i < xrows.size(); i++) x[i] = 0;
    }
    if (rc != CIRCL_HIP_OK) { circl_hip_keytable_free(t); return rc; }
    *out = t;
    return CIRCL_HIP_OK;
}

int circl_hip_hybrid_keytable_new(int scheme, int private_keys, const uint8_t *keys, size_t nkeys, int device, uint8_t *key_status,
                                  circl_hip_keytable **out) {
    if (out) *out = nullptr;
    Desc d;
    if (!desc_of(scheme, d) || d.r3 || !keys || !out || nkeys == 0 || nkeys > 0xffffffffull) return CIRCL_HIP_EPARAM;  // (round-3 Kyber has no key tables)
    return keytable_replicate(device, [&](int dev, circl_hip_keytable **one) {
        return hybrid_keytable_new_one(scheme, d, private_keys, keys, nkeys, dev, dev == 0 || device >= 0 ? key_status : nullptr, one);
    }, out);
}

// workspace of the table forms: the per-call temporaries + the ML-KEM workspace (as circl_hip_hybrid_workspace_size)
int circl_hip_hybrid_encaps_table_dev(const circl_hip_keytable *t, const uint32_t *d_key_idx, const uint8_t *d_eseed, uint8_t *d_ct, uint8_t *d_ss,
                                      uint8_t *d_status, size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    t = keytable_here(t);
    if (!hybrid_table_ok(t, 0)) return CIRCL_HIP_EPARAM;
    Desc d;
    if (!desc_of(t->scheme, d)) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    if (!d_status || !d_eseed || !d_ct || !d_ss || !d_ws) return CIRCL_HIP_EPARAM;
    if (ws_bytes < hybrid_ws_min(d, n)) return CIRCL_HIP_EWORKSPACE;
    if (!args_ok(d_eseed, d_ct, d_ss, nullptr, d_ws) || (reinterpret_cast<uintptr_t>(d_key_idx) & 3)) return misaligned();
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carve c{static_cast<uint8_t *>(d_ws)};
    uint8_t *pkx = c.take(n * 32), *m = c.take(n * 32), *ekx = c.take(n * 32), *ctm = c.take(n * d.CTM), *ssm = c.take(n * 32), *ctx = c.take(n * 32),
            *ssx = c.take(n * 32), *okx = c.take(n);
    uint8_t *kws = static_cast<uint8_t *>(d_ws) + tmp_bytes(d, n);
    const size_t kws_bytes = ws_bytes - tmp_bytes(d, n);
    SecretWipe wipe{st};
    wipe.add(m, n * 32); wipe.add(ekx, n * 32); wipe.add(ssm, n * 32); wipe.add(ssx, n * 32);
    TRY(gather_rows(st, pkx, t->d_x, t->nkeys, d_key_idx, n));
    if (d.xwing) {
        TRY(copy_rows(st, m, 32, d_eseed, 64, 32, n));
        TRY(copy_rows(st, ekx, 32, d_eseed + 32, 64, 32, n));
    } else {
        hipLaunchKernelGGL((hk::hybrid_expand_kernel<4, 4, false>), g256(n), dim3(256), 0, st, w(d_eseed), w(m), w(ekx), n);
        HIP_TRY(hipGetLastError());
    }
    TRY(circl_hip_mlkem_encaps_table_dev(t->inner, d_key_idx, m, ctm, ssm, d_status, n, kws, kws_bytes, st));
    TRY(x25519_pair_dev(ekx, pkx, ctx, ssx, okx, n, st));
    TRY(copy_rows(st, d_ct + d.kem_off(32), d.ct, ctm, d.CTM, d.CTM, n));
    TRY(copy_rows(st, d_ct + d.x_off(d.CTM), d.ct, ctx, 32, 32, n));
    if (d.xwing) {
        hipLaunchKernelGGL(hk::xwing_combine_kernel, g256(n), dim3(256), 0, st, w(ssm), w(ssx), w(ctx), w(pkx), d_status, w(d_ss), n);
        HIP_TRY(hipGetLastError());
    } else {
        hipLaunchKernelGGL(hk::hybrid_status_kernel, g256(n), dim3(256), 0, st, d_status, okx, n);
        HIP_TRY(hipGetLastError());
        TRY(copy_rows(st, d_ss + d.kem_off(32), 64, ssm, 32, 32, n));
        TRY(copy_rows(st, d_ss + d.x_off(32), 64, ssx, 32, 32, n));
        TRY(zero_failed(st, d_ss, 64, d_status, n));
    }
    TRY(zero_failed(st, d_ct, d.ct, d_status, n));
    return wipe.finish();
}

int circl_hip_hybrid_decaps_table_dev(const circl_hip_keytable *t, const uint32_t *d_key_idx, const uint8_t *d_ct, uint8_t *d_ss, uint8_t *d_status,
                                      size_t n, void *d_ws, size_t ws_bytes, void *stream) {
    t = keytable_here(t);
    if (!hybrid_table_ok(t, 1)) return CIRCL_HIP_EPARAM;
    Desc d;
    if (!desc_of(t->scheme, d)) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    if (!d_status || !d_ct || !d_ss || !d_ws) return CIRCL_HIP_EPARAM;
    if (ws_bytes < hybrid_ws_min(d, n)) return CIRCL_HIP_EWORKSPACE;
    if (!args_ok(d_ct, d_ss, nullptr, nullptr, d_ws) || (reinterpret_cast<uintptr_t>(d_key_idx) & 3)) return misaligned();
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carve c{static_cast<uint8_t *>(d_ws)};
    uint8_t *skx = c.take(n * 32), *pkx = c.take(n * 32), *ctm = c.take(n * d.CTM), *ctx = c.take(n * 32), *ssm = c.take(n * 32), *ssx = c.take(n * 32),
            *okx = c.take(n);
    uint8_t *kws = static_cast<uint8_t *>(d_ws) + tmp_bytes(d, n);
    const size_t kws_bytes = ws_bytes - tmp_bytes(d, n);
    SecretWipe wipe{st};
    wipe.add(skx, n * 32); wipe.add(ssm, n * 32); wipe.add(ssx, n * 32);
    TRY(copy_rows(st, ctm, d.CTM, d_ct + d.kem_off(32), d.ct, d.CTM, n));
    TRY(copy_rows(st, ctx, 32, d_ct + d.x_off(d.CTM), d.ct, 32, n));
    TRY(gather_rows(st, skx, t->d_x, t->nkeys, d_key_idx, n));
    if (d.xwing) TRY(gather_rows(st, pkx, t->d_x + t->nkeys * 32, t->nkeys, d_key_idx, n));  // sk.xpk, computed when the table was built
    TRY(circl_hip_x25519_dev(skx, ctx, ssx, okx, n, st));
    TRY(circl_hip_mlkem_decaps_table_dev(t->inner, d_key_idx, ctm, ssm, d_status, n, kws, kws_bytes, st));
    if (d.xwing) {
        hipLaunchKernelGGL(hk::xwing_combine_kernel, g256(n), dim3(256), 0, st, w(ssm), w(ssx), w(ctx), w(pkx), static_cast<const uint8_t *>(nullptr),
                           w(d_ss), n);
        HIP_TRY(hipGetLastError());
    } else {
        hipLaunchKernelGGL(hk::hybrid_status_kernel, g256(n), dim3(256), 0, st, d_status, okx, n);
        HIP_TRY(hipGetLastError());
        TRY(copy_rows(st, d_ss + d.kem_off(32), 64, ssm, 32, 32, n));
        TRY(copy_rows(st, d_ss + d.x_off(32), 64, ssx, 32, 32, n));
        TRY(zero_failed(st, d_ss, 64, d_status, n));
    }
    return wipe.finish();
}

// ---- host-buffer forms on the staging pipeline ----
// Every chunk wipes what is secret in its device staging: the secret inputs / outputs (flagged per array), this file's own rows (seeds,
// dk, scalars, halves of shared secrets: all of tmp_bytes) and the per-item slots at the head of the ML-KEM workspace behind them -- not
// the matrix scratch and the row cache further back, which are public (a whole-slot memset was several hundred MB per chunk).
static PipeOpts hybrid_opts(int scheme) {
    PipeOpts o;
    o.chunk_items = host_chunk_items(size_t(1) << 16);  // 2 x 1024 ladder waves per launch: a ladder is 0.9 ms however few items
    o.depth = 3;  // measured (tools/hybrid_bench.py 20 host): 2.8e7 X-Wing encapsulations/s at 3, 2.4e7 at 4 or 6, 2.1e7 at 2
    o.wipe_device = true;
    o.ws_secret_bytes = [scheme](size_t cnt) {
        Desc d;
        return desc_of(scheme, d) ? tmp_bytes(d, cnt) + mlkem_ws_secret_bytes(cnt) : size_t(0);
    };
    return o;
}
// (a chunk beyond 2^13 items is PCIe-bound whichever ML-KEM route it takes: it gets the workspace of the scratch routes)
static std::function<size_t(size_t)> hybrid_ws_fn(int scheme) {
    return [scheme](size_t k) {
        Desc d;
        if (!desc_of(scheme, d)) return size_t(0);
        return k <= (size_t(1) << 13) ? circl_hip_hybrid_workspace_size(scheme, k) : hybrid_ws_min(d, k);
    };
}

int circl_hip_hybrid_keygen(int scheme, const uint8_t *seed, uint8_t *pk, uint8_t *sk, size_t n, int device) {
    Desc s;
    if (!desc_of(scheme, s)) return CIRCL_HIP_EPARAM;
    if (n && (!seed || !pk || !sk)) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{seed + lo * s.seed, s.seed, true}}, {}, {{pk + lo * s.pk, s.pk}, {sk + lo * s.sk, s.sk, true}},
                            hybrid_ws_fn(scheme), hybrid_opts(scheme),
                            [&](Chunk &c) { return circl_hip_hybrid_keygen_dev(scheme, c.in[0], c.out[0], c.out[1], c.cnt, c.ws, c.ws_bytes, c.st); });
    }, kHeavyOneDeviceMax);
}

int circl_hip_hybrid_encaps(int scheme, const uint8_t *pk, const uint8_t *eseed, uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, int device) {
    Desc s;
    if (!desc_of(scheme, s)) return CIRCL_HIP_EPARAM;
    if (n && (!pk || !eseed || !ct || !ss)) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        const std::vector<HIn> ins = {{pk + lo * s.pk, s.pk}, {eseed + lo * s.eseed, s.eseed, true}};
        const std::vector<HOut> outs = {{ct + lo * s.ct, s.ct}, {ss + lo * s.ss, s.ss, true}, {status ? status + lo : nullptr, 1}};
        auto launch = [&](Chunk &c) { return circl_hip_hybrid_encaps_dev(scheme, c.in[0], c.in[1], c.out[0], c.out[1], c.out[2], c.cnt, c.ws, c.ws_bytes, c.st); };
        // circl_hip_set_coalesce: what a TLS 1.3 server does per X25519MLKEM768 handshake -- one encapsulation to the client's ephemeral
        // share -- from many connections at once (kem/hybrid/hybrid.go:271-300); a ladder costs ~0.8 ms however few items share it
        if (Coalescer *co = call_coalescer(kCoHybEncaps, scheme - 1, dev)) {
            const int rc = coalesce_run(co, cnt, ins, {}, outs, hybrid_ws_fn(scheme), hybrid_opts(scheme), launch);
            if (rc != kNotCoalesced) return rc;
        }
        return run_pipeline(dev, cnt, ins, {}, outs, hybrid_ws_fn(scheme), hybrid_opts(scheme), launch);
    }, kHeavyOneDeviceMax);
}

int circl_hip_hybrid_decaps(int scheme, const uint8_t *sk, const uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, int device) {
    Desc s;
    if (!desc_of(scheme, s)) return CIRCL_HIP_EPARAM;
    if (n && (!sk || !ct || !ss)) return CIRCL_HIP_EPARAM;
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        const std::vector<HIn> ins = {{sk + lo * s.sk, s.sk, true}, {ct + lo * s.ct, s.ct}};
        const std::vector<HOut> outs = {{ss + lo * s.ss, s.ss, true}, {status ? status + lo : nullptr, 1}};
        auto launch = [&](Chunk &c) { return circl_hip_hybrid_decaps_dev(scheme, c.in[0], c.in[1], c.out[0], c.out[1], c.cnt, c.ws, c.ws_bytes, c.st); };
        if (Coalescer *co = call_coalescer(kCoHybDecaps, scheme - 1, dev)) {
            const int rc = coalesce_run(co, cnt, ins, {}, outs, hybrid_ws_fn(scheme), hybrid_opts(scheme), launch);
            if (rc != kNotCoalesced) return rc;
        }
        return run_pipeline(dev, cnt, ins, {}, outs, hybrid_ws_fn(scheme), hybrid_opts(scheme), launch);
    }, kHeavyOneDeviceMax);
}

static int check_idx(const uint32_t *key_idx, size_t n, size_t nkeys) {
    if (key_idx)
        for (size_t i = 0; i < n; i++)
            if (key_idx[i] >= nkeys) return CIRCL_HIP_EPARAM;
    return CIRCL_HIP_OK;
}
// host buffers through a resident hybrid table.  key_idx is the one OPTIONAL array (absent: entry 0 for every item).  A small call joins the
// table's cross-caller batch (circl_hip_keytable_set_coalesce) or, submitted, its asynchronous queue (circl_hip_keytable_async_start) -- the
// X25519 ladder makes ONE hybrid launch ~0.8 ms whatever it holds, so these are the calls that gain most from sharing it.
static std::vector<HIn> hyb_enc_ins(const uint32_t *ki, const uint8_t *eseed, const Desc &s) {
    return {{reinterpret_cast<const uint8_t *>(ki), size_t(4), false, false, true}, {eseed, s.eseed, true}};
}
static std::vector<HIn> hyb_dec_ins(const uint32_t *ki, const uint8_t *ct, const Desc &s) {
    return {{reinterpret_cast<const uint8_t *>(ki), size_t(4), false, false, true}, {ct, s.ct}};
}
int circl_hip_hybrid_encaps_table(const circl_hip_keytable *t, const uint32_t *key_idx, const uint8_t *eseed, uint8_t *ct, uint8_t *ss, uint8_t *status,
                                  size_t n) {
    if (!t || t->magic != kKeytableMagic || t->family != 3 || t->private_keys) return CIRCL_HIP_EPARAM;
    Desc s;
    if (!desc_of(t->scheme, s)) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    if (!eseed || !ct || !ss) return CIRCL_HIP_EPARAM;
    TRY(check_idx(key_idx, n, t->nkeys));
    const int scheme = t->scheme;
    return table_shard(t, n, [&](const circl_hip_keytable *r, size_t lo, size_t cnt) {
        const uint32_t *ki = key_idx ? key_idx + lo : nullptr;
        Coalescer *co = usable_coalescer(r);
        if (co && cnt <= coalescer_call_max(co)) {
            const int rc = coalesce_run(co, cnt, hyb_enc_ins(ki, eseed + lo * s.eseed, s), {},
                                        {{ct + lo * s.ct, s.ct}, {ss + lo * s.ss, s.ss, true}, {status ? status + lo : nullptr, 1}}, hybrid_ws_fn(scheme), hybrid_opts(scheme),
                                        [&](Chunk &c) {
                                            return circl_hip_hybrid_encaps_table_dev(r, reinterpret_cast<const uint32_t *>(c.in[0]), c.in[1], c.out[0], c.out[1], c.out[2],
                                                                                     c.cnt, c.ws, c.ws_bytes, c.st);
                                        });
            if (rc != kNotCoalesced) return rc;
        }
        return run_pipeline(r->device, cnt, {{reinterpret_cast<const uint8_t *>(ki), ki ? size_t(4) : size_t(0)}, {eseed + lo * s.eseed, s.eseed, true}}, {},
                            {{ct + lo * s.ct, s.ct}, {ss + lo * s.ss, s.ss, true}, {status ? status + lo : nullptr, 1}},
                            hybrid_ws_fn(scheme), hybrid_opts(scheme), [&](Chunk &c) {
                                return circl_hip_hybrid_encaps_table_dev(r, ki ? reinterpret_cast<const uint32_t *>(c.in[0]) : nullptr, c.in[1], c.out[0], c.out[1],
                                                                         c.out[2], c.cnt, c.ws, c.ws_bytes, c.st);
                            });
    }, kHeavyOneDeviceMax);
}
int circl_hip_hybrid_decaps_table(const circl_hip_keytable *t, const uint32_t *key_idx, const uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n) {
    if (!t || t->magic != kKeytableMagic || t->family != 3 || !t->private_keys) return CIRCL_HIP_EPARAM;
    Desc s;
    if (!desc_of(t->scheme, s)) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    if (!ct || !ss) return CIRCL_HIP_EPARAM;
    TRY(check_idx(key_idx, n, t->nkeys));
    const int scheme = t->scheme;
    return table_shard(t, n, [&](const circl_hip_keytable *r, size_t lo, size_t cnt) {
        const uint32_t *ki = key_idx ? key_idx + lo : nullptr;
        Coalescer *co = usable_coalescer(r);
        if (co && cnt <= coalescer_call_max(co)) {
            const int rc = coalesce_run(co, cnt, hyb_dec_ins(ki, ct + lo * s.ct, s), {}, {{ss + lo * s.ss, s.ss, true}, {status ? status + lo : nullptr, 1}},
                                        hybrid_ws_fn(scheme), hybrid_opts(scheme), [&](Chunk &c) {
                                            return circl_hip_hybrid_decaps_table_dev(r, reinterpret_cast<const uint32_t *>(c.in[0]), c.in[1], c.out[0], c.out[1], c.cnt,
                                                                                     c.ws, c.ws_bytes, c.st);
                                        });
            if (rc != kNotCoalesced) return rc;
        }
        return run_pipeline(r->device, cnt, {{reinterpret_cast<const uint8_t *>(ki), ki ? size_t(4) : size_t(0)}, {ct + lo * s.ct, s.ct}}, {},
                            {{ss + lo * s.ss, s.ss, true}, {status ? status + lo : nullptr, 1}},
                            hybrid_ws_fn(scheme), hybrid_opts(scheme), [&](Chunk &c) {
                                return circl_hip_hybrid_decaps_table_dev(r, ki ? reinterpret_cast<const uint32_t *>(c.in[0]) : nullptr, c.in[1], c.out[0], c.out[1],
                                                                         c.cnt, c.ws, c.ws_bytes, c.st);
                            });
    }, kHeavyOneDeviceMax);
}
// ---- the asynchronous form (include/circl_hip.h: circl_hip_keytable_async_start) ----
int circl_hip_hybrid_encaps_table_submit(const circl_hip_keytable *t, const uint32_t *key_idx, const uint8_t *eseed, uint8_t *ct, uint8_t *ss, uint8_t *status,
                                         size_t n, uint64_t *ticket) {
    if (ticket) *ticket = 0;
    if (!t || t->magic != kKeytableMagic || t->family != 3 || t->private_keys || !ticket) return CIRCL_HIP_EPARAM;
    Desc s;
    if (!desc_of(t->scheme, s)) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    if (!eseed || !ct || !ss) return CIRCL_HIP_EPARAM;
    TRY(check_idx(key_idx, n, t->nkeys));
    return table_submit(t, ticket, [&](const circl_hip_keytable *, Coalescer *co, uint64_t *seq) {
        return coalesce_submit(co, n, hyb_enc_ins(key_idx, eseed, s), {}, {{ct, s.ct}, {ss, s.ss, true}, {status, 1}}, seq, false);
    });
}
int circl_hip_hybrid_decaps_table_submit(const circl_hip_keytable *t, const uint32_t *key_idx, const uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n,
                                         uint64_t *ticket) {
    if (ticket) *ticket = 0;
    if (!t || t->magic != kKeytableMagic || t->family != 3 || !t->private_keys || !ticket) return CIRCL_HIP_EPARAM;
    Desc s;
    if (!desc_of(t->scheme, s)) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    if (!ct || !ss) return CIRCL_HIP_EPARAM;
    TRY(check_idx(key_idx, n, t->nkeys));
    return table_submit(t, ticket, [&](const circl_hip_keytable *, Coalescer *co, uint64_t *seq) {
        return coalesce_submit(co, n, hyb_dec_ins(key_idx, ct, s), {}, {{ss, s.ss, true}, {status, 1}}, seq, false);
    });
}

}  // extern "C"
namespace circl {
namespace host {
// circl_hip_queue for the hybrids with the key in the call -- what a TLS 1.3 server does per X25519MLKEM768 handshake (kem/hybrid/hybrid.go:271-300):
// the arrays and the launch of circl_hip_hybrid_encaps / _decaps
int hyb_call_queue_start(bool decaps, int scheme, Coalescer *co, bool want_eventfd, QueueShape *sh) {
    Desc s;
    if (!desc_of(scheme, s)) return CIRCL_HIP_EPARAM;
    if (!decaps) {
        *sh = QueueShape{s.pk, s.eseed, s.ct, s.ss, false, true};
        return coalescer_async_start(co, {{nullptr, s.pk}, {nullptr, s.eseed, true}}, {}, {{nullptr, s.ct}, {nullptr, s.ss, true}, {nullptr, 1}}, hybrid_ws_fn(scheme),
                                     hybrid_opts(scheme), [scheme](Chunk &c) {
                                         return circl_hip_hybrid_encaps_dev(scheme, c.in[0], c.in[1], c.out[0], c.out[1], c.out[2], c.cnt, c.ws, c.ws_bytes, c.st);
                                     }, want_eventfd);
    }
    *sh = QueueShape{s.sk, s.ct, 0, s.ss, true, false};
    return coalescer_async_start(co, {{nullptr, s.sk, true}, {nullptr, s.ct}}, {}, {{nullptr, s.ss, true}, {nullptr, 1}}, hybrid_ws_fn(scheme), hybrid_opts(scheme),
                                 [scheme](Chunk &c) { return circl_hip_hybrid_decaps_dev(scheme, c.in[0], c.in[1], c.out[0], c.out[1], c.cnt, c.ws, c.ws_bytes, c.st); },
                                 want_eventfd);
}
// the queue of one hybrid table (part): its arrays and its launch, fixed for the queue's life (`r` outlives the queue: the table owns it)
int hyb_table_async_start(const circl_hip_keytable *r, Coalescer *co, bool want_eventfd) {
    Desc s;
    if (!desc_of(r->scheme, s)) return CIRCL_HIP_EPARAM;
    const int scheme = r->scheme;
    if (!r->private_keys)
        return coalescer_async_start(co, hyb_enc_ins(nullptr, nullptr, s), {}, {{nullptr, s.ct}, {nullptr, s.ss, true}, {nullptr, 1}}, hybrid_ws_fn(scheme), hybrid_opts(scheme),
                                     [r](Chunk &c) {
                                         return circl_hip_hybrid_encaps_table_dev(r, reinterpret_cast<const uint32_t *>(c.in[0]), c.in[1], c.out[0], c.out[1], c.out[2], c.cnt,
                                                                                  c.ws, c.ws_bytes, c.st);
                                     }, want_eventfd);
    return coalescer_async_start(co, hyb_dec_ins(nullptr, nullptr, s), {}, {{nullptr, s.ss, true}, {nullptr, 1}}, hybrid_ws_fn(scheme), hybrid_opts(scheme), [r](Chunk &c) {
        return circl_hip_hybrid_decaps_table_dev(r, reinterpret_cast<const uint32_t *>(c.in[0]), c.in[1], c.out[0], c.out[1], c.cnt, c.ws, c.ws_bytes, c.st);
    }, want_eventfd);
}
}  // namespace host
}  // namespace circl
