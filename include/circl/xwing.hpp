// circl/xwing.hpp -- X-Wing (X25519 + ML-KEM-768) on top of the HIP batch engine: SURVEY.md 8(f) row f2,
// the hybrid caller of ML-KEM-768.  Mirrors kem/xwing/xwing.go:
//
//   DeriveKeyPairPacked(seed[32])     -> (sk[32], pk[1216])          xwing.go:98-144
//   Encapsulate(pk, eseed[64])        -> (ss[32], ct[1120])          xwing.go:187-201, :223-265
//   Decapsulate(ct, sk)               -> ss[32]                      xwing.go:203-210, :270-299
//   combiner = SHA3-256(ss_M || ss_X || ct_X || pk_X || "\.//^\")    xwing.go:53-71
// plus batch forms.  Everything runs on the GPU behind circl_hip_hybrid_* (scheme CIRCL_HIP_HYBRID_XWING): seed
// expansion, ML-KEM-768, both X25519 ladders per item (one lane each), the combiner.  Link with -lcirclhip.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
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
}  // namespace detail

inline void DeriveKeyPairBatch(const uint8_t *seeds, uint8_t *sks, uint8_t *pks, size_t n, int device = 0) {
    detail::check(circl_hip_hybrid_keygen(CIRCL_HIP_HYBRID_XWING, seeds, pks, sks, n, device), "xwing keygen");
}

// status[i] != 0 -> kem.ErrPubKey (the ML-KEM half failed the encapsulation-key check, xwing.go:301-311)
inline void EncapsulateBatch(const uint8_t *pks, const uint8_t *eseeds, uint8_t *sss, uint8_t *cts, uint8_t *status, size_t n, int device = 0) {
    detail::check(circl_hip_hybrid_encaps(CIRCL_HIP_HYBRID_XWING, pks, eseeds, cts, sss, status, n, device), "xwing encaps");
}

inline void DecapsulateBatch(const uint8_t *cts, const uint8_t *sks, uint8_t *sss, size_t n, int device = 0) {
    detail::check(circl_hip_hybrid_decaps(CIRCL_HIP_HYBRID_XWING, sks, cts, sss, nullptr, n, device), "xwing decaps");
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
