// circl/xwing.hpp -- X-Wing (X25519 + ML-KEM-768) on top of the HIP batch engine: SURVEY.md 8(f) row f2,
// the hybrid caller of ML-KEM-768.  Mirrors kem/xwing/xwing.go:
//
//   DeriveKeyPairPacked(seed[32])     -> (sk[32], pk[1216])          xwing.go:98-144
//   Encapsulate(pk, eseed[64])        -> (ss[32], ct[1120])          xwing.go:187-201, :223-265
//   Decapsulate(ct, sk)               -> ss[32]                      xwing.go:203-210, :270-299
//   combiner = SHA3-256(ss_M || ss_X || ct_X || pk_X || "\.//^\")    xwing.go:53-71
// plus batch forms.  Division of labour as SURVEY.md prescribes: the GPU does the ML-KEM-768 half and every
// Keccak (seed expansion with SHAKE256, the SHA3-256 combiner, both as batched sponges); the CPU does
// X25519 (OpenSSL's EVP_PKEY_X25519 on all host cores -- elliptic-curve arithmetic is out of scope for the GPU path).
// Link with -lcirclhip -lcrypto.
#pragma once
#include <openssl/evp.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <thread>
#include <utility>
#include <vector>

#include "../circl_hip.h"

namespace circl {
namespace xwing {

using Bytes = std::vector<uint8_t>;
constexpr int SeedSize = 32, PublicKeySize = 1216, PrivateKeySize = 32, EncapsulationSeedSize = 64, SharedKeySize = 32, CiphertextSize = 1120;
constexpr int MlkemEk = 1184, MlkemDk = 2400, MlkemCt = 1088;

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

namespace detail {
inline void check(int rc, const char *what) {
    if (rc != CIRCL_HIP_OK) throw Error(std::string("circl-hip ") + what + ": error " + std::to_string(rc) + " " + circl_hip_last_error());
}
// X25519(scalar, u) and X25519(scalar, 9) through OpenSSL (dh/x25519 in the reference)
inline void x25519_public(uint8_t pub[32], const uint8_t priv[32]) {
    EVP_PKEY *k = EVP_PKEY_new_raw_private_key(EVP_PKEY_X25519, nullptr, priv, 32);
    size_t len = 32;
    if (!k || EVP_PKEY_get_raw_public_key(k, pub, &len) != 1 || len != 32) { EVP_PKEY_free(k); throw Error("x25519 keygen failed"); }
    EVP_PKEY_free(k);
}
inline void x25519_shared(uint8_t out[32], const uint8_t priv[32], const uint8_t peer[32]) {
    EVP_PKEY *k = EVP_PKEY_new_raw_private_key(EVP_PKEY_X25519, nullptr, priv, 32);
    EVP_PKEY *p = EVP_PKEY_new_raw_public_key(EVP_PKEY_X25519, nullptr, peer, 32);
    EVP_PKEY_CTX *c = k ? EVP_PKEY_CTX_new(k, nullptr) : nullptr;
    size_t len = 32;
    // a low-order peer point makes OpenSSL fail where the reference returns zeros (xwing.go:254-257): keep zeros
    std::memset(out, 0, 32);
    if (c && p && EVP_PKEY_derive_init(c) == 1 && EVP_PKEY_derive_set_peer(c, p) == 1) (void)EVP_PKEY_derive(c, out, &len);
    EVP_PKEY_CTX_free(c);
    EVP_PKEY_free(p);
    EVP_PKEY_free(k);
}
// The per-item X25519 work of a batch, spread over the host's cores (OpenSSL's EVP calls are thread-safe on distinct
// objects).  fn(i) for i in [0, n); the first exception thrown by any worker is rethrown on the caller's thread.
template <class F> inline void parallel_for(size_t n, F fn) {
    const size_t hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nthreads = std::min(hw, (n + 63) / 64);  // at least 64 items (a few milliseconds of X25519) per thread
    if (nthreads <= 1) {
        for (size_t i = 0; i < n; i++) fn(i);
        return;
    }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> err(nthreads);
    for (size_t t = 0; t < nthreads; t++) {
        pool.emplace_back([&, t] {
            try {
                for (size_t i = n * t / nthreads; i < n * (t + 1) / nthreads; i++) fn(i);
            } catch (...) {
                err[t] = std::current_exception();
            }
        });
    }
    for (auto &th : pool) th.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
}
}  // namespace detail

// expands n 32-byte seeds: SHAKE256(seed) -> seedm[64] || skx[32]  (xwing.go:119-124), on the GPU
inline void expand_seeds(const uint8_t *seeds, uint8_t *out96, size_t n, int device = 0) {
    detail::check(circl_hip_shake(136, 0x1f, seeds, 32, out96, 96, n, device), "shake256");
}

inline void DeriveKeyPairBatch(const uint8_t *seeds, uint8_t *sks, uint8_t *pks, size_t n, int device = 0) {
    std::vector<uint8_t> ex(96 * n), seedm(64 * n), ek(MlkemEk * n), dk(MlkemDk * n);
    expand_seeds(seeds, ex.data(), n, device);
    for (size_t i = 0; i < n; i++) std::memcpy(&seedm[64 * i], &ex[96 * i], 64);
    detail::check(circl_hip_mlkem_keygen(768, seedm.data(), ek.data(), dk.data(), n, device), "mlkem keygen");
    detail::parallel_for(n, [&](size_t i) {
        std::memcpy(sks + 32 * i, seeds + 32 * i, 32);                    // the packed private key is the seed
        std::memcpy(pks + PublicKeySize * i, &ek[MlkemEk * i], MlkemEk);
        detail::x25519_public(pks + PublicKeySize * i + MlkemEk, &ex[96 * i + 64]);
    });
}

// status[i] != 0 -> kem.ErrPubKey (the ML-KEM half failed the encapsulation-key check, xwing.go:301-311)
inline void EncapsulateBatch(const uint8_t *pks, const uint8_t *eseeds, uint8_t *sss, uint8_t *cts, uint8_t *status, size_t n, int device = 0) {
    std::vector<uint8_t> ek(MlkemEk * n), seedm(32 * n), ctm(MlkemCt * n), ssm(32 * n), comb(134 * n), st(n);
    for (size_t i = 0; i < n; i++) {
        std::memcpy(&ek[MlkemEk * i], pks + PublicKeySize * i, MlkemEk);
        std::memcpy(&seedm[32 * i], eseeds + 64 * i, 32);
    }
    detail::check(circl_hip_mlkem_encaps(768, ek.data(), seedm.data(), ctm.data(), ssm.data(), st.data(), n, device), "mlkem encaps");
    detail::parallel_for(n, [&](size_t i) {
        const uint8_t *ekx = eseeds + 64 * i + 32, *pkx = pks + PublicKeySize * i + MlkemEk;
        uint8_t *c = &comb[134 * i], *ct = cts + CiphertextSize * i;
        std::memcpy(ct, &ctm[MlkemCt * i], MlkemCt);
        detail::x25519_public(ct + MlkemCt, ekx);                          // ct_X
        std::memcpy(c, &ssm[32 * i], 32);
        detail::x25519_shared(c + 32, ekx, pkx);                           // ss_X
        std::memcpy(c + 64, ct + MlkemCt, 32);
        std::memcpy(c + 96, pkx, 32);
        std::memcpy(c + 128, "\\.//^\\", 6);
        if (status) status[i] = st[i];
    });
    detail::check(circl_hip_shake(136, 0x06, comb.data(), 134, sss, 32, n, device), "sha3-256 combiner");
}

inline void DecapsulateBatch(const uint8_t *cts, const uint8_t *sks, uint8_t *sss, size_t n, int device = 0) {
    std::vector<uint8_t> ex(96 * n), seedm(64 * n), ek(MlkemEk * n), dk(MlkemDk * n), ctm(MlkemCt * n), ssm(32 * n), comb(134 * n), st(n);
    expand_seeds(sks, ex.data(), n, device);
    for (size_t i = 0; i < n; i++) {
        std::memcpy(&seedm[64 * i], &ex[96 * i], 64);
        std::memcpy(&ctm[MlkemCt * i], cts + CiphertextSize * i, MlkemCt);
    }
    detail::check(circl_hip_mlkem_keygen(768, seedm.data(), ek.data(), dk.data(), n, device), "mlkem keygen");
    detail::check(circl_hip_mlkem_decaps(768, dk.data(), ctm.data(), ssm.data(), st.data(), n, device), "mlkem decaps");
    detail::parallel_for(n, [&](size_t i) {
        const uint8_t *skx = &ex[96 * i + 64], *ctx = cts + CiphertextSize * i + MlkemCt;
        uint8_t *c = &comb[134 * i];
        std::memcpy(c, &ssm[32 * i], 32);
        detail::x25519_shared(c + 32, skx, ctx);
        std::memcpy(c + 64, ctx, 32);
        detail::x25519_public(c + 96, skx);                                // sk.xpk
        std::memcpy(c + 128, "\\.//^\\", 6);
    });
    detail::check(circl_hip_shake(136, 0x06, comb.data(), 134, sss, 32, n, device), "sha3-256 combiner");
}

// single-shot forms with the reference's signatures
inline std::pair<Bytes, Bytes> DeriveKeyPairPacked(const Bytes &seed) {
    if ((int)seed.size() != SeedSize) throw std::invalid_argument("kem: wrong seed size");  // the reference panics (ErrSeedSize)
    Bytes sk(PrivateKeySize), pk(PublicKeySize);
    DeriveKeyPairBatch(seed.data(), sk.data(), pk.data(), 1);
    return {sk, pk};
}
inline std::pair<Bytes, Bytes> Encapsulate(const Bytes &pk, const Bytes &eseed) {
    if ((int)pk.size() != PublicKeySize) throw Error("kem: wrong size for public key");
    if ((int)eseed.size() != EncapsulationSeedSize) throw Error("kem: wrong seed size");
    Bytes ss(SharedKeySize), ct(CiphertextSize);
    uint8_t st = 0;
    EncapsulateBatch(pk.data(), eseed.data(), ss.data(), ct.data(), &st, 1);
    if (st) throw Error("kem: invalid public key");
    return {ss, ct};
}
inline Bytes Decapsulate(const Bytes &ct, const Bytes &sk) {
    if ((int)ct.size() != CiphertextSize) throw Error("kem: wrong size for ciphertext");
    if ((int)sk.size() != PrivateKeySize) throw Error("kem: wrong size for private key");
    Bytes ss(SharedKeySize);
    DecapsulateBatch(ct.data(), sk.data(), ss.data(), 1);
    return ss;
}

}  // namespace xwing
}  // namespace circl
