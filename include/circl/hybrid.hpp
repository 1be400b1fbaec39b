// circl/hybrid.hpp -- X25519MLKEM768 on top of the HIP batch engine: SURVEY.md 8(f) row f2, the TLS hybrid
// that carries ML-KEM-768 today.  Mirrors kem/hybrid/hybrid.go (scheme{"X25519MLKEM768", mlkem768.Scheme(),
// x25519Kem}, :95-99) and kem/hybrid/xkem.go:
//
//   sizes: pk 1184 + 32, sk 2400 + 32, ct 1088 + 32, ss 32 + 32, seed max(64,32) = 64, eseed max(32,32) = 32
//          (hybrid.go:122-155: concatenation, ML-KEM half first)
//   DeriveKeyPair(seed[64])              SHAKE256(seed) -> 64 B for ML-KEM-768 || 32 B for X25519   hybrid.go:236-250
//                                        X25519 half: sk = SHAKE256(seed32)[:32], pk = X25519(sk, 9) xkem.go:112-123, :68-86
//   EncapsulateDeterministically(pk, s)  SHAKE256(s) -> m[32] || xseed[32];  ct = ct_M || X25519 pk of xseed,
//                                        ss = ss_M || X25519(sk(xseed), pk_X)                         hybrid.go:271-300, xkem.go:160-178
//   Decapsulate(sk, ct)                  ss_M || X25519(sk_X, ct_X)                                  hybrid.go:302-323, xkem.go:180-196
//   a low-order X25519 point is kem.ErrPubKey (xkem.go:144-146; xkem_test.go:35-69)
//
// Everything runs on the GPU behind circl_hip_hybrid_* (scheme CIRCL_HIP_HYBRID_X25519MLKEM768): both SHAKE256 seed
// expansions, ML-KEM-768, the X25519 ladders (one lane per item).  Link with -lcirclhip.
#pragma once
#include "xwing.hpp"

namespace circl {
namespace hybrid {
namespace x25519mlkem768 {

using Bytes = std::vector<uint8_t>;
constexpr int MlkemEk = 1184, MlkemDk = 2400, MlkemCt = 1088;
constexpr int PublicKeySize = MlkemEk + 32, PrivateKeySize = MlkemDk + 32, CiphertextSize = MlkemCt + 32, SharedKeySize = 64;
constexpr int SeedSize = 64, EncapsulationSeedSize = 32;
inline const char *Name() { return "X25519MLKEM768"; }

using Error = xwing::Error;
// per-item status of the batch calls
enum Status : uint8_t { Ok = 0, ErrPubKey = 1, ErrPrivKey = 2 };

namespace detail {
using xwing::detail::check;
}  // namespace detail

inline void DeriveKeyPairBatch(const uint8_t *seeds, uint8_t *pks, uint8_t *sks, size_t n, int device = 0) {
    detail::check(circl_hip_hybrid_keygen(CIRCL_HIP_HYBRID_X25519MLKEM768, seeds, pks, sks, n, device), "x25519mlkem768 keygen");
}

// status[i]: ErrPubKey if the ML-KEM half fails the encapsulation-key check or the X25519 half is a low-order point;
// the item's ct and ss are then zero (the reference returns nil, nil, err)
inline void EncapsulateBatch(const uint8_t *pks, const uint8_t *eseeds, uint8_t *cts, uint8_t *sss, uint8_t *status, size_t n, int device = 0) {
    detail::check(circl_hip_hybrid_encaps(CIRCL_HIP_HYBRID_X25519MLKEM768, pks, eseeds, cts, sss, status, n, device), "x25519mlkem768 encaps");
}

// status[i]: ErrPrivKey if the ML-KEM private key fails its hash check, ErrPubKey for a low-order X25519 ciphertext
inline void DecapsulateBatch(const uint8_t *sks, const uint8_t *cts, uint8_t *sss, uint8_t *status, size_t n, int device = 0) {
    detail::check(circl_hip_hybrid_decaps(CIRCL_HIP_HYBRID_X25519MLKEM768, sks, cts, sss, status, n, device), "x25519mlkem768 decaps");
}

// single-shot forms with the reference's signatures (kem.Scheme)
inline std::pair<Bytes, Bytes> DeriveKeyPair(const Bytes &seed) {
    if ((int)seed.size() != SeedSize) throw std::invalid_argument("kem: wrong seed size");  // the reference panics (ErrSeedSize)
    Bytes pk(PublicKeySize), sk(PrivateKeySize);
    DeriveKeyPairBatch(seed.data(), pk.data(), sk.data(), 1);
    return {pk, sk};
}
inline std::pair<Bytes, Bytes> EncapsulateDeterministically(const Bytes &pk, const Bytes &seed) {
    if ((int)seed.size() != EncapsulationSeedSize) throw Error("kem: wrong seed size");
    if ((int)pk.size() != PublicKeySize) throw Error("kem: wrong size for public key");
    Bytes ct(CiphertextSize), ss(SharedKeySize);
    uint8_t st = 0;
    EncapsulateBatch(pk.data(), seed.data(), ct.data(), ss.data(), &st, 1);
    if (st) throw Error("kem: invalid public key");
    return {ct, ss};
}
inline Bytes Decapsulate(const Bytes &sk, const Bytes &ct) {
    if ((int)ct.size() != CiphertextSize) throw Error("kem: wrong size for ciphertext");
    if ((int)sk.size() != PrivateKeySize) throw Error("kem: wrong size for private key");
    Bytes ss(SharedKeySize);
    uint8_t st = 0;
    DecapsulateBatch(sk.data(), ct.data(), ss.data(), &st, 1);
    if (st == ErrPrivKey) throw Error("kem: invalid private key");
    if (st) throw Error("kem: invalid public key");
    return ss;
}

}  // namespace x25519mlkem768
}  // namespace hybrid
}  // namespace circl
