// circl/sign.hpp -- host-side mirror of cloudflare/circl's sign.Scheme for the HIP batch engine.
// sign/sign.go:48-94; ML-DSA wrappers sign/mldsa/mldsa65/dilithium.go:256-345.
//
//   Scheme.Name / PublicKeySize / SignatureSize          same names
//   UnmarshalBinaryPublicKey(buf)                        length check only (dilithium.go:330-343)
//   Verify(pk, msg, sig, opts) bool                      false for a bad signature, malformed encoding,
//                                                        wrong signature length or ctx > 255 bytes;
//                                                        throws ErrTypeMismatch on a foreign key
//                                                        (the reference panics, dilithium.go:311-314)
//   DeriveKey(seed) -> (PublicKey, PrivateKey)           throws std::invalid_argument on a bad seed length
//                                                        (the reference panics, dilithium.go:272-281)
//   Sign(sk, msg, opts) []byte                           deterministic (like the reference's scheme.Sign,
//                                                        dilithium.go:283-303, which passes randomized=false);
//                                                        throws ErrContextTooLong for ctx > 255 bytes
//   VerifyBatch / SignBatch / DeriveKeyBatch             new: the batch calls
#pragma once
#include <cstdint>
#include <stdexcept>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../circl_hip.h"

namespace circl {
namespace sign {

using Bytes = std::vector<uint8_t>;
struct ErrTypeMismatch : std::runtime_error { ErrTypeMismatch() : std::runtime_error("sign: type mismatch") {} };
struct ErrPubKeySize : std::runtime_error { ErrPubKeySize() : std::runtime_error("sign: wrong size for public key") {} };
struct ErrPrivKeySize : std::runtime_error { ErrPrivKeySize() : std::runtime_error("sign: wrong size for private key") {} };
struct ErrContextTooLong : std::runtime_error { ErrContextTooLong() : std::runtime_error("sign: context string too long") {} };
struct ErrContextNotSupported : std::runtime_error { ErrContextNotSupported() : std::runtime_error("context not supported") {} };  // sign/sign.go:113-115
struct ErrDevice : std::runtime_error { using std::runtime_error::runtime_error; };

struct SignatureOpts {
    std::string Context;  // sign/sign.go:24-31
};

class Scheme;
// A key object made by UnmarshalBinary* carries what the reference's parsed objects carry -- A and tr for a public key, A and the
// NTT-domain s1, s2, t0 for a private key (sign/mldsa/mldsa65/internal/dilithium.go:114-126, :149-179) -- as a resident table on the
// scheme's device (circl_hip_mldsa_keytable_new / circl_hip_mldsa_privkey_new); copies share it, the last one frees (and wipes) it.
using ResidentKey = std::shared_ptr<circl_hip_keytable>;
struct PublicKey {
    const Scheme *scheme = nullptr;
    Bytes packed;
    ResidentKey resident;
    Bytes MarshalBinary() const { return packed; }
};
struct PrivateKey {
    const Scheme *scheme = nullptr;
    Bytes packed;
    ResidentKey resident;
    Bytes MarshalBinary() const { return packed; }
    PublicKey Public() const;  // sign/mldsa/mldsa65/internal/dilithium.go:473-484: t1 recomputed from s1, s2 (on the device)
};

class Scheme {
  public:
    Scheme(int param, const char *name) : param_(param), name_(name) {}
    std::string Name() const { return name_; }
    int PublicKeySize() const { return (int)circl_hip_mldsa_pk_size(param_); }
    int SignatureSize() const { return (int)circl_hip_mldsa_sig_size(param_); }
    int PrivateKeySize() const { return (int)circl_hip_mldsa_sk_size(param_); }
    int SeedSize() const { return 32; }
    // round-3 Dilithium2/3/5 (param 2 / 3 / 5) has no context: sign/dilithium/mode3/dilithium.go:213-215, :232-234, :249-251
    bool SupportsContext() const { return param_ > 5; }
    int device = 0;

    PublicKey UnmarshalBinaryPublicKey(const Bytes &buf) const {
        if ((int)buf.size() != PublicKeySize()) throw ErrPubKeySize();
        circl_hip_keytable *t = nullptr;  // parse once: A and tr stay on the device with the object
        check(circl_hip_mldsa_keytable_new(param_, buf.data(), 1, device, &t));
        return PublicKey{this, buf, ResidentKey(t, circl_hip_keytable_free)};
    }
    PrivateKey UnmarshalBinaryPrivateKey(const Bytes &buf) const {
        if ((int)buf.size() != PrivateKeySize()) throw ErrPrivKeySize();
        circl_hip_keytable *t = nullptr;  // ... and A with the NTT-domain secrets
        check(circl_hip_mldsa_privkey_new(param_, buf.data(), device, &t));
        return PrivateKey{this, buf, ResidentKey(t, circl_hip_keytable_free)};
    }
    std::pair<PublicKey, PrivateKey> DeriveKey(const Bytes &seed) const {
        if ((int)seed.size() != SeedSize()) throw std::invalid_argument("seed must be of length SeedSize");
        PublicKey pk{this, Bytes(PublicKeySize())};
        PrivateKey sk{this, Bytes(PrivateKeySize())};
        check(circl_hip_mldsa_keygen(param_, seed.data(), pk.packed.data(), sk.packed.data(), 1, dev1()));
        return {pk, sk};
    }
    Bytes Sign(const PrivateKey &sk, const Bytes &msg, const SignatureOpts *opts = nullptr) const {
        if (sk.scheme != this) throw ErrTypeMismatch();
        const std::string ctx = opts ? opts->Context : std::string();
        if (!SupportsContext() && !ctx.empty()) throw ErrContextNotSupported();
        if (ctx.size() > 255) throw ErrContextTooLong();
        const uint64_t moff[2] = {0, msg.size()}, coff[2] = {0, ctx.size()};
        const uint8_t pad = 0;
        Bytes sig(SignatureSize());
        const uint8_t *cp = ctx.empty() ? &pad : reinterpret_cast<const uint8_t *>(ctx.data());
        if (sk.resident) check(circl_hip_mldsa_sign_table(sk.resident.get(), msg.empty() ? &pad : msg.data(), moff, cp, coff, nullptr, sig.data(), 1));
        else check(circl_hip_mldsa_sign(param_, sk.packed.data(), msg.empty() ? &pad : msg.data(), moff, cp, coff, nullptr, sig.data(), 1, dev1()));
        return sig;
    }
    bool Verify(const PublicKey &pk, const Bytes &msg, const Bytes &sig, const SignatureOpts *opts = nullptr) const {
        if (pk.scheme != this) throw ErrTypeMismatch();
        if ((int)sig.size() != SignatureSize()) return false;  // internal/dilithium.go:90-93
        const std::string ctx = opts ? opts->Context : std::string();
        if (!SupportsContext() && !ctx.empty()) throw ErrContextNotSupported();
        if (ctx.size() > 255) return false;                    // dilithium.go:116-118
        const uint64_t moff[2] = {0, msg.size()}, coff[2] = {0, ctx.size()};
        const uint8_t pad = 0;
        uint8_t ok = 0;
        const uint8_t *cp = ctx.empty() ? &pad : reinterpret_cast<const uint8_t *>(ctx.data());
        const int rc = pk.resident ? circl_hip_mldsa_verify_table(pk.resident.get(), nullptr, sig.data(), msg.empty() ? &pad : msg.data(), moff, cp, coff, &ok, 1)
                                   : circl_hip_mldsa_verify(param_, pk.packed.data(), sig.data(), msg.empty() ? &pad : msg.data(), moff, cp, coff, &ok, 1,
                                                            device < 0 ? 0 : device);
        if (rc != CIRCL_HIP_OK) throw ErrDevice(std::string("circl-hip: error ") + std::to_string(rc) + " " + circl_hip_last_error());
        return ok != 0;
    }
    // rows: pk[n][PublicKeySize], sig[n][SignatureSize]; messages / contexts as blobs + n+1 offsets
    void VerifyBatch(const uint8_t *pks, const uint8_t *sigs, const uint8_t *msg_blob, const uint64_t *msg_off,
                     const uint8_t *ctx_blob, const uint64_t *ctx_off, uint8_t *ok, size_t n) const {
        const int rc = circl_hip_mldsa_verify(param_, pks, sigs, msg_blob, msg_off, ctx_blob, ctx_off, ok, n, device);
        if (rc != CIRCL_HIP_OK) throw ErrDevice(std::string("circl-hip: error ") + std::to_string(rc) + " " + circl_hip_last_error());
    }

    // rnd = n x 32 random bytes (hedged signing) or nullptr (deterministic)
    // n messages signed with ONE private key (the parsed-key case: A and the NTT-domain secrets cached, internal/dilithium.go:149-179)
    void SignSharedKeyBatch(const PrivateKey &sk, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob,
                            const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sigs, size_t n) const {
        if (sk.scheme != this) throw ErrTypeMismatch();
        const int rc = sk.resident ? circl_hip_mldsa_sign_table(sk.resident.get(), msg_blob, msg_off, ctx_blob, ctx_off, rnd, sigs, n)
                                   : circl_hip_mldsa_sign_shared(param_, sk.packed.data(), msg_blob, msg_off, ctx_blob, ctx_off, rnd, sigs, n, device);
        if (rc == CIRCL_HIP_EPARAM) throw ErrContextTooLong();
        if (rc != CIRCL_HIP_OK) throw ErrDevice(std::string("circl-hip: error ") + std::to_string(rc) + " " + circl_hip_last_error());
    }
    // n signatures under ONE public key (the parsed-key case: A and tr cached, internal/dilithium.go:114-126)
    void VerifySharedKeyBatch(const PublicKey &pk, const uint8_t *sigs, const uint8_t *msg_blob, const uint64_t *msg_off,
                              const uint8_t *ctx_blob, const uint64_t *ctx_off, uint8_t *ok, size_t n) const {
        if (pk.scheme != this) throw ErrTypeMismatch();
        const int rc = pk.resident ? circl_hip_mldsa_verify_table(pk.resident.get(), nullptr, sigs, msg_blob, msg_off, ctx_blob, ctx_off, ok, n)
                                   : circl_hip_mldsa_verify_shared(param_, pk.packed.data(), sigs, msg_blob, msg_off, ctx_blob, ctx_off, ok, n, device);
        if (rc != CIRCL_HIP_OK) throw ErrDevice(std::string("circl-hip: error ") + std::to_string(rc) + " " + circl_hip_last_error());
    }
    void SignBatch(const uint8_t *sks, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off,
                   const uint8_t *rnd, uint8_t *sigs, size_t n) const {
        check(circl_hip_mldsa_sign(param_, sks, msg_blob, msg_off, ctx_blob, ctx_off, rnd, sigs, n, device));
    }
    void DeriveKeyBatch(const uint8_t *seeds32, uint8_t *pks, uint8_t *sks, size_t n) const {
        check(circl_hip_mldsa_keygen(param_, seeds32, pks, sks, n, device));
    }

    // PrivateKey.Public() over a batch of packed private keys -> packed public keys (dilithium.go:473-484)
    void PublicBatch(const uint8_t *sks, uint8_t *pks, size_t n) const { check(circl_hip_mldsa_public_from_private(param_, sks, pks, n, device)); }

  private:
    int param_;
    const char *name_;
    int dev1() const { return device < 0 ? 0 : device; }
    static void check(int rc) {
        if (rc == CIRCL_HIP_EPARAM) throw ErrContextTooLong();
        if (rc != CIRCL_HIP_OK) throw ErrDevice(std::string("circl-hip: error ") + std::to_string(rc) + " " + circl_hip_last_error());
    }
};

inline PublicKey PrivateKey::Public() const {
    Bytes pk((size_t)scheme->PublicKeySize());
    scheme->PublicBatch(packed.data(), pk.data(), 1);
    return scheme->UnmarshalBinaryPublicKey(pk);
}

// sign/schemes/schemes.go:31-74
inline const Scheme *ByName(const std::string &name) {
    static const Scheme s44(44, "ML-DSA-44"), s65(65, "ML-DSA-65"), s87(87, "ML-DSA-87");
    static const Scheme d2(2, "Dilithium2"), d3(3, "Dilithium3"), d5(5, "Dilithium5");
    for (const Scheme *s : {&s44, &s65, &s87, &d2, &d3, &d5})
        if (s->Name() == name) return s;
    return nullptr;
}

}  // namespace sign
}  // namespace circl
