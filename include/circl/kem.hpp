// circl/kem.hpp -- host-side mirror of cloudflare/circl's kem.Scheme for the HIP batch engine.
//
// The reference's host language is Go; this image has no Go toolchain, so the host layer above the
// C ABI (include/circl_hip.h) is C++ and keeps the reference's names, argument meaning and error
// behaviour so that tests read like kem/schemes/schemes_test.go:
//
//   kem.Scheme (kem/kem.go:33-82)            circl::kem::Scheme
//     Name, PublicKeySize, PrivateKeySize,     same names
//     SeedSize, SharedKeySize, CiphertextSize,
//     EncapsulationSeedSize
//     DeriveKeyPair(seed)                      DeriveKeyPair(seed)   throws std::invalid_argument on a bad
//                                              seed length (the reference panics, kyber.go:341-343)
//     UnmarshalBinaryPublicKey(buf)            -> PublicKey; throws ErrPubKeySize / ErrPubKey
//     UnmarshalBinaryPrivateKey(buf)           -> PrivateKey; throws ErrPrivKeySize / ErrPrivKey
//     EncapsulateDeterministically(pk, seed)   -> {ct, ss}; throws ErrSeedSize
//     Encapsulate(pk)                          seed from std::random_device (crypto/rand in Go)
//     Decapsulate(sk, ct)                      -> ss; throws ErrCiphertextSize; an invalid
//                                              ciphertext is NOT an error (implicit rejection)
//   kem/schemes.ByName (schemes.go:70-75)    circl::kem::ByName
// plus the batch calls the reference lacks (EncapsulateBatch, DecapsulateBatch, DeriveKeyPairBatch).
//
// Keys are kept in MarshalBinary form; the GPU re-derives A^T and H(ek) per item (the reference
// caches them per key object, kem/mlkem/mlkem768/kyber.go:39-43), which is the right trade for
// batches of distinct keys.  Every operation runs on the GPU: there is no CPU path.
#pragma once
#include <cstdint>
#include <algorithm>
#include <random>
#include <stdexcept>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../circl_hip.h"

namespace circl {
namespace kem {

using Bytes = std::vector<uint8_t>;

// kem/kem.go:85-121 error values
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
struct ErrTypeMismatch : Error { ErrTypeMismatch() : Error("kem: type mismatch") {} };
struct ErrSeedSize : Error { ErrSeedSize() : Error("kem: wrong seed size") {} };
struct ErrPubKeySize : Error { ErrPubKeySize() : Error("kem: wrong size for public key") {} };
struct ErrCiphertextSize : Error { ErrCiphertextSize() : Error("kem: wrong size for ciphertext") {} };
struct ErrPrivKeySize : Error { ErrPrivKeySize() : Error("kem: wrong size for private key") {} };
struct ErrPubKey : Error { ErrPubKey() : Error("kem: invalid public key") {} };
struct ErrPrivKey : Error { ErrPrivKey() : Error("kem: invalid private key") {} };
struct ErrDevice : Error { explicit ErrDevice(const std::string &m) : Error("circl-hip: " + m) {} };

class Scheme;

// A key object made by UnmarshalBinary* carries what the reference's parsed objects carry (A^T, H(ek), the hash-check verdict:
// kem/mlkem/mlkem768/kyber.go:39-43, :219-228, :247-263) as a resident one-key table on the scheme's device
// (circl_hip_mlkem_keytable_new); copies of the object share it, the last one frees it.  Objects made by DeriveKeyPair carry
// none and go through the packed bytes, as before.
using ResidentKey = std::shared_ptr<circl_hip_keytable>;
class PublicKey {
  public:
    const Scheme *scheme = nullptr;
    Bytes packed;  // MarshalBinary form
    ResidentKey resident;
    Bytes MarshalBinary() const { return packed; }
    bool Equal(const PublicKey &o) const { return scheme == o.scheme && packed == o.packed; }
};
class PrivateKey {
  public:
    const Scheme *scheme = nullptr;
    Bytes packed;
    ResidentKey resident;
    Bytes MarshalBinary() const { return packed; }
    bool Equal(const PrivateKey &o) const { return scheme == o.scheme && packed == o.packed; }
    PublicKey Public() const;
};

class Scheme {
  public:
    // round3 = the pre-standard Kyber of kem/kyber/kyber{512,768,1024} (no key validation, different hashing)
    Scheme(int param, const char *name, bool round3 = false) : param_(param), name_(name), r3_(round3) {}
    std::string Name() const { return name_; }
    int PublicKeySize() const { return (int)circl_hip_mlkem_ek_size(param_); }
    int PrivateKeySize() const { return (int)circl_hip_mlkem_dk_size(param_); }
    int CiphertextSize() const { return (int)circl_hip_mlkem_ct_size(param_); }
    int SeedSize() const { return 64; }
    int SharedKeySize() const { return 32; }
    int EncapsulationSeedSize() const { return 32; }
    int device = 0;  // CIRCL_HIP_ALL_DEVICES splits batches over every GPU

    std::pair<PublicKey, PrivateKey> DeriveKeyPair(const Bytes &seed) const {
        if ((int)seed.size() != SeedSize()) throw std::invalid_argument("seed must be of length KeySeedSize");
        PublicKey pk{this, Bytes(PublicKeySize())};
        PrivateKey sk{this, Bytes(PrivateKeySize())};
        check(r3_ ? circl_hip_kyber_keygen(param_, seed.data(), pk.packed.data(), sk.packed.data(), 1, dev1())
                  : circl_hip_mlkem_keygen(param_, seed.data(), pk.packed.data(), sk.packed.data(), 1, dev1()));
        return {pk, sk};
    }
    std::pair<PublicKey, PrivateKey> GenerateKeyPair() const { return DeriveKeyPair(random_bytes(SeedSize())); }

    PublicKey UnmarshalBinaryPublicKey(const Bytes &buf) const {
        if ((int)buf.size() != PublicKeySize()) throw ErrPubKeySize();
        if (r3_) return PublicKey{this, buf, nullptr};  // kem/kyber/kyber768/kyber.go:248-262: non-canonical encodings are accepted
        // parse once: the key, A^T and H(ek) stay on the device with the object.  The canonical check == cpapke.go:45-55 is the
        // status byte of an encapsulation to the resident key
        PublicKey pk{this, buf, resident_key(buf, 0)};
        Bytes ct(CiphertextSize()), ss(32), m(32, 0);
        uint8_t st = 0;
        check(circl_hip_mlkem_encaps_table(pk.resident.get(), nullptr, m.data(), ct.data(), ss.data(), &st, 1));
        if (st == CIRCL_HIP_ITEM_ERR_PUBKEY) throw ErrPubKey();
        return pk;
    }
    PrivateKey UnmarshalBinaryPrivateKey(const Bytes &buf) const {
        if ((int)buf.size() != PrivateKeySize()) throw ErrPrivKeySize();
        if (r3_) return PrivateKey{this, buf, nullptr};  // kyber.go:215-232: no hash check
        uint8_t st = 0;
        PrivateKey sk{this, buf, resident_key(buf, 1, &st)};  // the stored-hash check (kyber.go:219-228) is the table's verdict
        if (st == CIRCL_HIP_ITEM_ERR_PRIVKEY) throw ErrPrivKey();
        return sk;
    }

    std::pair<Bytes, Bytes> EncapsulateDeterministically(const PublicKey &pk, const Bytes &seed) const {
        if (pk.scheme != this) throw ErrTypeMismatch();
        if ((int)seed.size() != EncapsulationSeedSize()) throw ErrSeedSize();
        Bytes ct(CiphertextSize()), ss(32);
        uint8_t st = 0;
        if (r3_) check(circl_hip_kyber_encaps(param_, pk.packed.data(), seed.data(), ct.data(), ss.data(), 1, dev1()));
        else if (pk.resident) check(circl_hip_mlkem_encaps_table(pk.resident.get(), nullptr, seed.data(), ct.data(), ss.data(), &st, 1));
        else check(circl_hip_mlkem_encaps(param_, pk.packed.data(), seed.data(), ct.data(), ss.data(), &st, 1, dev1()));
        if (st) throw ErrPubKey();
        return {ct, ss};
    }
    std::pair<Bytes, Bytes> Encapsulate(const PublicKey &pk) const {
        return EncapsulateDeterministically(pk, random_bytes(EncapsulationSeedSize()));
    }
    Bytes Decapsulate(const PrivateKey &sk, const Bytes &ct) const {
        if (sk.scheme != this) throw ErrTypeMismatch();
        if ((int)ct.size() != CiphertextSize()) throw ErrCiphertextSize();
        Bytes ss(32);
        uint8_t st = 0;
        if (r3_) check(circl_hip_kyber_decaps(param_, sk.packed.data(), ct.data(), ss.data(), 1, dev1()));
        else if (sk.resident) check(circl_hip_mlkem_decaps_table(sk.resident.get(), nullptr, ct.data(), ss.data(), &st, 1));
        else check(circl_hip_mlkem_decaps(param_, sk.packed.data(), ct.data(), ss.data(), &st, 1, dev1()));
        if (st) throw ErrPrivKey();
        return ss;
    }

    // ---- batch API (new; rows are MarshalBinary-form keys) -------------------------------------
    // status[i]: 0 ok, 1 = ErrPubKey, 2 = ErrPrivKey; failed items have zeroed outputs.
    void EncapsulateBatch(const uint8_t *eks, const uint8_t *seeds, uint8_t *cts, uint8_t *sss, uint8_t *status, size_t n) const {
        if (r3_) {
            check(circl_hip_kyber_encaps(param_, eks, seeds, cts, sss, n, device));
            if (status) std::fill(status, status + n, (uint8_t)0);
        } else {
            check(circl_hip_mlkem_encaps(param_, eks, seeds, cts, sss, status, n, device));
        }
    }
    // n encapsulations to ONE key (the cached-key case of the reference, kem/mlkem/mlkem768/kyber.go:39-43)
    void EncapsulateSharedKeyBatch(const PublicKey &pk, const uint8_t *seeds, uint8_t *cts, uint8_t *sss, uint8_t *status, size_t n) const {
        if (pk.scheme != this) throw ErrTypeMismatch();
        if (r3_) {  // no shared-key fast path for round-3 Kyber: replicate the row
            Bytes eks((size_t)PublicKeySize() * n);
            for (size_t i = 0; i < n; i++) std::copy(pk.packed.begin(), pk.packed.end(), eks.begin() + (size_t)PublicKeySize() * i);
            EncapsulateBatch(eks.data(), seeds, cts, sss, status, n);
            return;
        }
        // (the object's table lives where the scheme's `device` says: one device, or -- CIRCL_HIP_ALL_DEVICES -- replicated, the batch sharded)
        if (pk.resident) check(circl_hip_mlkem_encaps_table(pk.resident.get(), nullptr, seeds, cts, sss, status, n));
        else check(circl_hip_mlkem_encaps_shared(param_, pk.packed.data(), seeds, cts, sss, status, n, device));
    }
    // n ciphertexts for ONE private key (ML-KEM only)
    void DecapsulateSharedKeyBatch(const PrivateKey &sk, const uint8_t *cts, uint8_t *sss, uint8_t *status, size_t n) const {
        if (sk.scheme != this || r3_) throw ErrTypeMismatch();
        if (sk.resident) check(circl_hip_mlkem_decaps_table(sk.resident.get(), nullptr, cts, sss, status, n));
        else check(circl_hip_mlkem_decaps_shared(param_, sk.packed.data(), cts, sss, status, n, device));
    }
    void DecapsulateBatch(const uint8_t *dks, const uint8_t *cts, uint8_t *sss, uint8_t *status, size_t n) const {
        if (r3_) {
            check(circl_hip_kyber_decaps(param_, dks, cts, sss, n, device));
            if (status) std::fill(status, status + n, (uint8_t)0);
        } else {
            check(circl_hip_mlkem_decaps(param_, dks, cts, sss, status, n, device));
        }
    }
    void DeriveKeyPairBatch(const uint8_t *seeds64, uint8_t *eks, uint8_t *dks, size_t n) const {
        check(r3_ ? circl_hip_kyber_keygen(param_, seeds64, eks, dks, n, device) : circl_hip_mlkem_keygen(param_, seeds64, eks, dks, n, device));
    }

    // PrivateKey.Public() over a batch of packed decapsulation keys (kem/mlkem/mlkem768/kyber.go:323-328; the same layout for round 3)
    void PublicBatch(const uint8_t *dks, uint8_t *eks, size_t n) const { check(circl_hip_mlkem_public_from_private(param_, dks, eks, n)); }

  private:
    int param_;
    const char *name_;
    bool r3_;
    int dev1() const { return device < 0 ? 0 : device; }
    ResidentKey resident_key(const Bytes &buf, int private_key, uint8_t *verdict = nullptr) const {
        circl_hip_keytable *t = nullptr;
        check(circl_hip_mlkem_keytable_new(param_, private_key, buf.data(), 1, device, verdict, &t));
        return ResidentKey(t, circl_hip_keytable_free);
    }
    static void check(int rc) {
        if (rc != CIRCL_HIP_OK) throw ErrDevice(std::string("error ") + std::to_string(rc) + " " + circl_hip_last_error());
    }
    static Bytes random_bytes(int n) {
        std::random_device rd;
        Bytes b((size_t)n);
        for (auto &x : b) x = (uint8_t)rd();
        return b;
    }
};

inline PublicKey PrivateKey::Public() const {
    const int k = (scheme->PublicKeySize() - 32) / 384;
    return PublicKey{scheme, Bytes(packed.begin() + 384 * k, packed.begin() + 384 * k + scheme->PublicKeySize()), nullptr};
}

// kem/schemes/schemes.go:35-75
inline const Scheme *ByName(const std::string &name) {
    static const Scheme s512(512, "ML-KEM-512"), s768(768, "ML-KEM-768"), s1024(1024, "ML-KEM-1024");
    static const Scheme k512(512, "Kyber512", true), k768(768, "Kyber768", true), k1024(1024, "Kyber1024", true);
    for (const Scheme *s : {&s512, &s768, &s1024, &k512, &k768, &k1024})
        if (s->Name() == name) return s;
    return nullptr;
}

}  // namespace kem
}  // namespace circl
