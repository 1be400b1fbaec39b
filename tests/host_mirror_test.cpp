// Exercises the C++ host mirror of kem.Scheme / sign.Scheme the way kem/schemes/schemes_test.go:53-167
// and sign/schemes/schemes_test.go:17-108 exercise the Go interfaces.  Needs a GPU (run from
// tests/test_gpu_host_mirror.py).  Prints "OK" on success.
#include <cstdio>
#include <thread>
#include <future>
#include <atomic>
#include <cstring>

#include "circl/kem.hpp"
#include "circl/serving.hpp"
#include "circl/sign.hpp"

#define REQUIRE(c)                                                        \
    do {                                                                  \
        if (!(c)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } \
    } while (0)

template <class E, class F> bool throws(F &&f) {
    try { f(); } catch (const E &) { return true; } catch (...) { return false; }
    return false;
}

int main() {
    using namespace circl;
    for (const char *name : {"ML-KEM-512", "ML-KEM-768", "ML-KEM-1024", "Kyber512", "Kyber768", "Kyber1024"}) {
        const kem::Scheme *s = kem::ByName(name);
        const bool r3 = name[0] == 'K';  // round-3 Kyber validates neither key (kem/kyber/kyber768/kyber.go:215-262)
        REQUIRE(s && s->Name() == name);
        kem::Bytes seed(s->SeedSize());
        for (size_t i = 0; i < seed.size(); i++) seed[i] = (uint8_t)(3 * i + 1);
        auto [pk, sk] = s->DeriveKeyPair(seed);
        auto [pk2, sk2] = s->DeriveKeyPair(seed);
        REQUIRE(pk.Equal(pk2) && sk.Equal(sk2));
        REQUIRE((int)pk.MarshalBinary().size() == s->PublicKeySize() && (int)sk.MarshalBinary().size() == s->PrivateKeySize());
        REQUIRE(sk.Public().Equal(pk));
        kem::PublicKey pk3 = s->UnmarshalBinaryPublicKey(pk.MarshalBinary());
        kem::PrivateKey sk3 = s->UnmarshalBinaryPrivateKey(sk.MarshalBinary());
        REQUIRE(pk3.Equal(pk) && sk3.Equal(sk));
        kem::Bytes eseed(32, 7);
        auto [ct, ss] = s->EncapsulateDeterministically(pk, eseed);
        auto [ct2, ss2] = s->EncapsulateDeterministically(pk3, eseed);
        REQUIRE(ct == ct2 && ss == ss2 && (int)ct.size() == s->CiphertextSize() && (int)ss.size() == s->SharedKeySize());
        REQUIRE(s->Decapsulate(sk, ct) == ss);
        auto [ct3, ss3] = s->Encapsulate(pk);
        REQUIRE(s->Decapsulate(sk3, ct3) == ss3);
        // error behaviour (kem/kem.go:85-121)
        kem::Bytes shortbuf(10);
        REQUIRE(throws<kem::ErrPubKeySize>([&] { s->UnmarshalBinaryPublicKey(shortbuf); }));
        REQUIRE(throws<kem::ErrPrivKeySize>([&] { s->UnmarshalBinaryPrivateKey(shortbuf); }));
        REQUIRE(throws<kem::ErrCiphertextSize>([&] { s->Decapsulate(sk, shortbuf); }));
        REQUIRE(throws<kem::ErrSeedSize>([&] { s->EncapsulateDeterministically(pk, shortbuf); }));
        REQUIRE(throws<std::invalid_argument>([&] { s->DeriveKeyPair(shortbuf); }));
        kem::Bytes badpk = pk.MarshalBinary();
        badpk[0] = 0xff; badpk[1] |= 0x0f;
        if (r3) { (void)s->EncapsulateDeterministically(s->UnmarshalBinaryPublicKey(badpk), eseed); }
        else REQUIRE(throws<kem::ErrPubKey>([&] { s->UnmarshalBinaryPublicKey(badpk); }));
        kem::Bytes badsk = sk.MarshalBinary();
        badsk[badsk.size() - 40] ^= 1;
        if (r3) { (void)s->UnmarshalBinaryPrivateKey(badsk); }
        else REQUIRE(throws<kem::ErrPrivKey>([&] { s->UnmarshalBinaryPrivateKey(badsk); }));
        ct[3] ^= 1;  // invalid ciphertext: no error, different key
        REQUIRE(s->Decapsulate(sk, ct) != ss);
        const kem::Scheme *other = kem::ByName(std::strcmp(name, "ML-KEM-768") ? "ML-KEM-768" : "ML-KEM-512");
        REQUIRE(throws<kem::ErrTypeMismatch>([&] { other->EncapsulateDeterministically(pk, eseed); }));
        // batch
        const size_t n = 100;
        std::vector<uint8_t> seeds(64 * n), eks(n * s->PublicKeySize()), dks(n * s->PrivateKeySize()), ms(32 * n), cts(n * s->CiphertextSize()),
            sss(32 * n), sss2(32 * n), st(n);
        for (size_t i = 0; i < seeds.size(); i++) seeds[i] = (uint8_t)(i * 7 + i / 64);
        for (size_t i = 0; i < ms.size(); i++) ms[i] = (uint8_t)(i * 13);
        s->DeriveKeyPairBatch(seeds.data(), eks.data(), dks.data(), n);
        s->EncapsulateBatch(eks.data(), ms.data(), cts.data(), sss.data(), st.data(), n);
        for (auto x : st) REQUIRE(x == 0);
        s->DecapsulateBatch(dks.data(), cts.data(), sss2.data(), st.data(), n);
        REQUIRE(sss == sss2);
        // shared-key batch == the same key in every row
        std::vector<uint8_t> eks1(n * s->PublicKeySize()), cts1(cts.size()), sss1(sss.size()), cts2(cts.size()), sss3(sss.size());
        for (size_t i = 0; i < n; i++) std::copy(pk.packed.begin(), pk.packed.end(), eks1.begin() + i * s->PublicKeySize());
        s->EncapsulateBatch(eks1.data(), ms.data(), cts1.data(), sss1.data(), st.data(), n);
        s->EncapsulateSharedKeyBatch(pk, ms.data(), cts2.data(), sss3.data(), st.data(), n);
        REQUIRE(cts1 == cts2 && sss1 == sss3);
        if (!r3) {  // the same through PARSED key objects: their A^T / H(ek) live on the device with the object (and with its copies)
            REQUIRE(pk3.resident && sk3.resident && !pk.resident);
            kem::PublicKey pk4 = pk3;  // a copy shares the resident key
            std::fill(cts2.begin(), cts2.end(), 0);
            s->EncapsulateSharedKeyBatch(pk4, ms.data(), cts2.data(), sss3.data(), st.data(), n);
            REQUIRE(cts1 == cts2 && sss1 == sss3);
            s->DecapsulateSharedKeyBatch(sk3, cts2.data(), sss2.data(), st.data(), n);
            REQUIRE(sss2 == sss3);
            for (auto x : st) REQUIRE(x == 0);
        }
    }
    REQUIRE(kem::ByName("FrodoKEM-640-SHAKE") == nullptr);
    for (const char *name : {"ML-DSA-44", "ML-DSA-65", "ML-DSA-87"}) {
        const sign::Scheme *s = sign::ByName(name);
        REQUIRE(s && s->Name() == name && s->SupportsContext());
        sign::Bytes junk(s->PublicKeySize(), 1), sig(s->SignatureSize(), 2), msg{1, 2, 3};
        sign::PublicKey pk = s->UnmarshalBinaryPublicKey(junk);
        REQUIRE(!s->Verify(pk, msg, sig));
        sign::Bytes shortsig(5);
        REQUIRE(!s->Verify(pk, msg, shortsig));
        sign::SignatureOpts longctx{std::string(256, 'a')};
        REQUIRE(!s->Verify(pk, msg, sig, &longctx));
        REQUIRE(throws<sign::ErrPubKeySize>([&] { s->UnmarshalBinaryPublicKey(shortsig); }));
        // sign/schemes/schemes_test.go:17-108: derive, sign, verify, tamper
        sign::Bytes seed(s->SeedSize());
        for (size_t i = 0; i < seed.size(); i++) seed[i] = (uint8_t)(i * 5 + 2);
        auto [gpk, gsk] = s->DeriveKey(seed);
        auto [gpk2, gsk2] = s->DeriveKey(seed);
        REQUIRE(gpk.packed == gpk2.packed && gsk.packed == gsk2.packed);
        REQUIRE((int)gsk.MarshalBinary().size() == s->PrivateKeySize());
        sign::SignatureOpts ctx{"a context"};
        sign::Bytes sg = s->Sign(gsk, msg, &ctx);
        REQUIRE((int)sg.size() == s->SignatureSize());
        REQUIRE(sg == s->Sign(s->UnmarshalBinaryPrivateKey(gsk.MarshalBinary()), msg, &ctx));  // deterministic
        REQUIRE(s->Verify(gpk, msg, sg, &ctx));
        REQUIRE(!s->Verify(gpk, msg, sg));                    // wrong context
        sign::Bytes msg2 = msg; msg2[0] ^= 1;
        REQUIRE(!s->Verify(gpk, msg2, sg, &ctx));
        sg[sg.size() / 2] ^= 4;
        REQUIRE(!s->Verify(gpk, msg, sg, &ctx));
        REQUIRE(throws<sign::ErrContextTooLong>([&] { s->Sign(gsk, msg, &longctx); }));
        {   // shared-key batches: 20 messages under the one key pair, same bytes as the one-by-one calls
            const size_t nb = 20;
            std::vector<uint8_t> blob;
            std::vector<uint64_t> off{0};
            for (size_t i = 0; i < nb; i++) { for (size_t k = 0; k < 5 + i; k++) blob.push_back((uint8_t)(i * 17 + k)); off.push_back(blob.size()); }
            blob.resize(blob.size() + 16);
            std::vector<uint8_t> sigs(nb * s->SignatureSize()), ok(nb);
            s->SignSharedKeyBatch(gsk, blob.data(), off.data(), nullptr, nullptr, nullptr, sigs.data(), nb);
            for (size_t i = 0; i < nb; i++) {
                sign::Bytes mi(blob.begin() + off[i], blob.begin() + off[i + 1]);
                sign::Bytes one = s->Sign(gsk, mi);
                REQUIRE(std::equal(one.begin(), one.end(), sigs.begin() + i * s->SignatureSize()));
            }
            sigs[3 * s->SignatureSize() + 100] ^= 1;
            s->VerifySharedKeyBatch(gpk, sigs.data(), blob.data(), off.data(), nullptr, nullptr, ok.data(), nb);
            for (size_t i = 0; i < nb; i++) REQUIRE(ok[i] == (i == 3 ? 0 : 1));
            // the same through PARSED key objects (A, tr / the NTT-domain secrets resident with the object)
            const sign::PrivateKey psk = s->UnmarshalBinaryPrivateKey(gsk.MarshalBinary());
            const sign::PublicKey ppk = s->UnmarshalBinaryPublicKey(gpk.MarshalBinary());
            REQUIRE(psk.resident && ppk.resident && !gsk.resident);
            std::vector<uint8_t> sigs2(sigs.size());
            s->SignSharedKeyBatch(psk, blob.data(), off.data(), nullptr, nullptr, nullptr, sigs2.data(), nb);
            sigs[3 * s->SignatureSize() + 100] ^= 1;
            REQUIRE(sigs2 == sigs);
            sigs2[5 * s->SignatureSize() + 9] ^= 2;
            s->VerifySharedKeyBatch(ppk, sigs2.data(), blob.data(), off.data(), nullptr, nullptr, ok.data(), nb);
            for (size_t i = 0; i < nb; i++) REQUIRE(ok[i] == (i == 5 ? 0 : 1));
            REQUIRE(s->Verify(ppk, msg, s->Sign(psk, msg, &ctx), &ctx));
        }
        REQUIRE(throws<std::invalid_argument>([&] { s->DeriveKey(shortsig); }));
    }
    // round-3 Dilithium2/3/5 (sign/dilithium/mode{2,3,5}/dilithium.go:213-255): no context support
    for (const char *name : {"Dilithium2", "Dilithium3", "Dilithium5"}) {
        const sign::Scheme *s = sign::ByName(name);
        REQUIRE(s && s->Name() == name && !s->SupportsContext());
        sign::Bytes seed(s->SeedSize(), 9), msg{4, 5, 6, 7};
        auto [pk, sk] = s->DeriveKey(seed);
        REQUIRE((int)pk.MarshalBinary().size() == s->PublicKeySize() && (int)sk.MarshalBinary().size() == s->PrivateKeySize());
        sign::Bytes sg = s->Sign(sk, msg);
        REQUIRE((int)sg.size() == s->SignatureSize() && sg == s->Sign(sk, msg));
        REQUIRE(s->Verify(pk, msg, sg));
        sign::Bytes msg2 = msg; msg2[1] ^= 1;
        REQUIRE(!s->Verify(pk, msg2, sg));
        sign::SignatureOpts ctx{"x"};
        REQUIRE(throws<sign::ErrContextNotSupported>([&] { s->Sign(sk, msg, &ctx); }));
        REQUIRE(throws<sign::ErrContextNotSupported>([&] { s->Verify(pk, msg, sg, &ctx); }));
        REQUIRE(sk.Public().MarshalBinary() == pk.MarshalBinary());  // PrivateKey.Public(): dilithium.go:473-484
    }
    {
        // a scheme that targets EVERY device (device = -1): parsed key objects are replicated tables, shared-key batches shard over
        // the devices and give the bytes of the one-device scheme (sign/mldsa/mldsa65/dilithium.go:283-327, kem/mlkem/mlkem768/kyber.go:347-386)
        kem::Scheme all(768, "ML-KEM-768"), one(768, "ML-KEM-768");
        all.device = CIRCL_HIP_ALL_DEVICES;
        kem::Bytes seed(64, 3);
        auto kp = one.DeriveKeyPair(seed);
        const kem::PublicKey pa = all.UnmarshalBinaryPublicKey(kp.first.MarshalBinary()), po = one.UnmarshalBinaryPublicKey(kp.first.MarshalBinary());
        const kem::PrivateKey sa = all.UnmarshalBinaryPrivateKey(kp.second.MarshalBinary());
        REQUIRE(circl_hip_keytable_device(pa.resident.get()) == CIRCL_HIP_ALL_DEVICES && circl_hip_keytable_device(po.resident.get()) == 0);
        const size_t n = 777;
        std::vector<uint8_t> ms(32 * n), c1(n * 1088), s1(32 * n), c2(n * 1088), s2(32 * n), s3(32 * n), st(n);
        for (size_t i = 0; i < ms.size(); i++) ms[i] = (uint8_t)(i * 13 + 5);
        all.EncapsulateSharedKeyBatch(pa, ms.data(), c1.data(), s1.data(), st.data(), n);
        one.EncapsulateSharedKeyBatch(po, ms.data(), c2.data(), s2.data(), st.data(), n);
        REQUIRE(c1 == c2 && s1 == s2);
        all.DecapsulateSharedKeyBatch(sa, c1.data(), s3.data(), st.data(), n);
        REQUIRE(s3 == s1);
        sign::Scheme dall(65, "ML-DSA-65");
        dall.device = CIRCL_HIP_ALL_DEVICES;
        const sign::Scheme *d1 = sign::ByName("ML-DSA-65");
        auto dk = d1->DeriveKey(sign::Bytes(32, 7));
        const sign::PrivateKey dsa = dall.UnmarshalBinaryPrivateKey(dk.second.MarshalBinary());
        const sign::PublicKey dpa = dall.UnmarshalBinaryPublicKey(dk.first.MarshalBinary());
        sign::Bytes msg{9, 8, 7};
        REQUIRE(dall.Sign(dsa, msg) == d1->Sign(dk.second, msg) && dall.Verify(dpa, msg, dall.Sign(dsa, msg)));
    }
    {
        // kem::Serving (include/circl/serving.hpp): the compiled counterpart of go/kem/mlkem/hipbatch/{reactor,serving}.go -- key objects with
        // an asynchronous queue and ONE reactor thread each; six caller threads issue single operations (futures), every result must be
        // kem::Scheme's own (kem/mlkem/mlkem768/kyber.go:347-386)
        const kem::Scheme *s = kem::ByName("ML-KEM-768");
        kem::Serving srv(s, 0, 64, 128);
        auto kp = s->DeriveKeyPair(kem::Bytes(64, 11));
        auto spk = srv.UnmarshalBinaryPublicKey(kp.first.MarshalBinary());
        auto ssk = srv.UnmarshalBinaryPrivateKey(kp.second.MarshalBinary());
        constexpr int T = 6, N = 40;
        std::vector<std::thread> th;
        std::atomic<int> bad{0};
        for (int t = 0; t < T; t++) {
            th.emplace_back([&, t] {
                std::vector<std::future<kem::ServingKey::Result>> enc;
                std::vector<kem::Bytes> seeds;
                for (int i = 0; i < N; i++) {
                    kem::Bytes seed(32);
                    for (int b = 0; b < 32; b++) seed[b] = (uint8_t)(t * 41 + i * 7 + b);
                    seeds.push_back(seed);
                    enc.push_back(srv.EncapsulateAsync(*spk, seed));  // N requests outstanding from this thread alone
                }
                for (int i = 0; i < N; i++) {
                    const kem::ServingKey::Result r = enc[i].get();
                    const auto want = s->EncapsulateDeterministically(kp.first, seeds[i]);
                    if (r.first != want.first || r.second != want.second) bad++;
                    kem::Bytes ct = r.first;
                    if (i % 3 == 0) ct[5] ^= 1;  // implicit rejection: not an error, the reference's pseudo-random secret
                    if (srv.Decapsulate(*ssk, ct) != s->Decapsulate(kp.second, ct)) bad++;
                }
            });
        }
        for (auto &x : th) x.join();
        REQUIRE(bad.load() == 0);
        REQUIRE(throws<kem::ErrSeedSize>([&] { srv.EncapsulateAsync(*spk, kem::Bytes(5)); }));
        REQUIRE(throws<kem::ErrCiphertextSize>([&] { srv.DecapsulateAsync(*ssk, kem::Bytes(5)); }));
        REQUIRE(throws<kem::ErrTypeMismatch>([&] { srv.EncapsulateAsync(*ssk, kem::Bytes(32)); }));
        kem::Bytes badsk = kp.second.MarshalBinary();
        badsk[badsk.size() - 40] ^= 1;  // the stored H(ek) no longer matches: kem.ErrPrivKey at parse time (kyber.go:219-228)
        REQUIRE(throws<kem::ErrPrivKey>([&] { srv.UnmarshalBinaryPrivateKey(badsk); }));
        // a key object goes away with requests still in flight: they are finished first
        auto f = srv.EncapsulateAsync(*spk, kem::Bytes(32, 9));
        spk.reset();
        REQUIRE(f.get().second == s->EncapsulateDeterministically(kp.first, kem::Bytes(32, 9)).second);
    }
    std::printf("OK\n");
    return 0;
}
