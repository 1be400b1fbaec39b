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
// Division of labour as for X-Wing: ML-KEM-768 and every SHAKE256 run as batches on the GPU, X25519 on the
// CPU through OpenSSL.  Link with -lcirclhip -lcrypto.
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
using xwing::detail::x25519_public;
// X25519 with the reference's contract: false (-> kem.ErrPubKey) when the shared point is zero (dh/x25519 Shared)
inline bool x25519_shared_checked(uint8_t out[32], const uint8_t priv[32], const uint8_t peer[32]) {
    xwing::detail::x25519_shared(out, priv, peer);  // zeros on failure
    uint8_t acc = 0;
    for (int i = 0; i < 32; i++) acc |= out[i];
    return acc != 0;
}
// the X25519 "KEM" of xkem.go: DeriveKeyPair(seed) = (X25519(SHAKE256(seed)[:32], 9), SHAKE256(seed)[:32]); batched on the GPU
inline void x_private_keys(const uint8_t *seeds32, size_t stride, uint8_t *sk32, size_t n, int device) {
    std::vector<uint8_t> in(32 * n);
    for (size_t i = 0; i < n; i++) std::memcpy(&in[32 * i], seeds32 + stride * i, 32);
    check(circl_hip_shake(136, 0x1f, in.data(), 32, sk32, 32, n, device), "shake256");
}
}  // namespace detail

inline void DeriveKeyPairBatch(const uint8_t *seeds, uint8_t *pks, uint8_t *sks, size_t n, int device = 0) {
    std::vector<uint8_t> ex(96 * n), seedm(64 * n), ek(MlkemEk * n), dk(MlkemDk * n), skx(32 * n);
    detail::check(circl_hip_shake(136, 0x1f, seeds, 64, ex.data(), 96, n, device), "shake256");
    for (size_t i = 0; i < n; i++) std::memcpy(&seedm[64 * i], &ex[96 * i], 64);
    detail::check(circl_hip_mlkem_keygen(768, seedm.data(), ek.data(), dk.data(), n, device), "mlkem keygen");
    detail::x_private_keys(ex.data() + 64, 96, skx.data(), n, device);
    xwing::detail::parallel_for(n, [&](size_t i) {
        std::memcpy(pks + PublicKeySize * i, &ek[MlkemEk * i], MlkemEk);
        detail::x25519_public(pks + PublicKeySize * i + MlkemEk, &skx[32 * i]);
        std::memcpy(sks + PrivateKeySize * i, &dk[MlkemDk * i], MlkemDk);
        std::memcpy(sks + PrivateKeySize * i + MlkemDk, &skx[32 * i], 32);
    });
}

// status[i]: ErrPubKey if the ML-KEM half fails the encapsulation-key check or the X25519 half is a low-order point
inline void EncapsulateBatch(const uint8_t *pks, const uint8_t *eseeds, uint8_t *cts, uint8_t *sss, uint8_t *status, size_t n, int device = 0) {
    std::vector<uint8_t> ex(64 * n), ek(MlkemEk * n), m(32 * n), ctm(MlkemCt * n), ssm(32 * n), st(n), skx(32 * n);
    detail::check(circl_hip_shake(136, 0x1f, eseeds, 32, ex.data(), 64, n, device), "shake256");
    for (size_t i = 0; i < n; i++) {
        std::memcpy(&ek[MlkemEk * i], pks + PublicKeySize * i, MlkemEk);
        std::memcpy(&m[32 * i], &ex[64 * i], 32);
    }
    detail::check(circl_hip_mlkem_encaps(768, ek.data(), m.data(), ctm.data(), ssm.data(), st.data(), n, device), "mlkem encaps");
    detail::x_private_keys(ex.data() + 32, 64, skx.data(), n, device);
    xwing::detail::parallel_for(n, [&](size_t i) {
        uint8_t *ct = cts + CiphertextSize * i, *ss = sss + SharedKeySize * i;
        std::memcpy(ct, &ctm[MlkemCt * i], MlkemCt);
        detail::x25519_public(ct + MlkemCt, &skx[32 * i]);
        std::memcpy(ss, &ssm[32 * i], 32);
        const bool ok = detail::x25519_shared_checked(ss + 32, &skx[32 * i], pks + PublicKeySize * i + MlkemEk);
        const uint8_t s = (st[i] || !ok) ? ErrPubKey : Ok;
        if (s) { std::memset(ct, 0, CiphertextSize); std::memset(ss, 0, SharedKeySize); }  // the reference returns nil, nil, err
        if (status) status[i] = s;
    });
}

// status[i]: ErrPrivKey if the ML-KEM private key fails its hash check, ErrPubKey for a low-order X25519 ciphertext
inline void DecapsulateBatch(const uint8_t *sks, const uint8_t *cts, uint8_t *sss, uint8_t *status, size_t n, int device = 0) {
    std::vector<uint8_t> dk(MlkemDk * n), ctm(MlkemCt * n), ssm(32 * n), st(n);
    for (size_t i = 0; i < n; i++) {
        std::memcpy(&dk[MlkemDk * i], sks + PrivateKeySize * i, MlkemDk);
        std::memcpy(&ctm[MlkemCt * i], cts + CiphertextSize * i, MlkemCt);
    }
    detail::check(circl_hip_mlkem_decaps(768, dk.data(), ctm.data(), ssm.data(), st.data(), n, device), "mlkem decaps");
    xwing::detail::parallel_for(n, [&](size_t i) {
        uint8_t *ss = sss + SharedKeySize * i;
        std::memcpy(ss, &ssm[32 * i], 32);
        const bool ok = detail::x25519_shared_checked(ss + 32, sks + PrivateKeySize * i + MlkemDk, cts + CiphertextSize * i + MlkemCt);
        const uint8_t s = st[i] ? ErrPrivKey : (!ok ? ErrPubKey : Ok);
        if (s) std::memset(ss, 0, SharedKeySize);
        if (status) status[i] = s;
    });
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
