// X25519MLKEM768 through include/circl/hybrid.hpp.  Mirrors kem/hybrid/xkem_test.go:35-69 (low-order X25519
// points are kem.ErrPubKey on both sides) and the generic scheme test kem/schemes/schemes_test.go:60-133
// (sizes, round trip).  With "dump <n>" it prints seed/eseed-derived pk, sk, ct, ss as hex lines so that the
// Python test can compare them with the oracle (ML-KEM, SHAKE256) + libcrypto (X25519).  Needs a GPU.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "circl/hybrid.hpp"

namespace H = circl::hybrid::x25519mlkem768;

static void hex(const char *tag, const std::vector<uint8_t> &v) {
    std::printf("%s ", tag);
    for (uint8_t b : v) std::printf("%02x", b);
    std::printf("\n");
}

int main(int argc, char **argv) {
    if (argc >= 3 && std::string(argv[1]) == "dump") {
        const int n = std::atoi(argv[2]);
        for (int i = 0; i < n; i++) {
            H::Bytes seed(H::SeedSize), eseed(H::EncapsulationSeedSize);
            for (int k = 0; k < H::SeedSize; k++) seed[k] = (uint8_t)(i * 131 + k * 7 + 1);
            for (int k = 0; k < H::EncapsulationSeedSize; k++) eseed[k] = (uint8_t)(i * 17 + k * 29 + 5);
            auto [pk, sk] = H::DeriveKeyPair(seed);
            auto [ct, ss] = H::EncapsulateDeterministically(pk, eseed);
            hex("seed", seed); hex("eseed", eseed); hex("pk", pk); hex("sk", sk); hex("ct", ct); hex("ss", ss);
            if (H::Decapsulate(sk, ct) != ss) { std::printf("FAILED decapsulate\n"); return 1; }
        }
        return 0;
    }
    static_assert(H::PublicKeySize == 1216 && H::PrivateKeySize == 2432 && H::CiphertextSize == 1120 && H::SharedKeySize == 64, "sizes");
    H::Bytes seed(H::SeedSize, 3), eseed(H::EncapsulationSeedSize, 9);
    auto [pk, sk] = H::DeriveKeyPair(seed);
    auto [ct, ss] = H::EncapsulateDeterministically(pk, eseed);
    if (H::Decapsulate(sk, ct) != ss) { std::printf("FAILED round trip\n"); return 1; }
    // order-8 point, xkem_test.go:21-25
    const uint8_t low[32] = {0xe0, 0xeb, 0x7a, 0x7c, 0x3b, 0x41, 0xb8, 0xae, 0x16, 0x56, 0xe3, 0xfa, 0xf1, 0x9f, 0xc4, 0x6a,
                             0xda, 0x09, 0x8d, 0xeb, 0x9c, 0x32, 0xb1, 0xfd, 0x86, 0x62, 0x05, 0x16, 0x5f, 0x49, 0xb8, 0x00};
    {   // TestLowOrderX25519PointEncapsulate
        H::Bytes bad = pk;
        std::copy(low, low + 32, bad.end() - 32);
        bool threw = false;
        try { (void)H::EncapsulateDeterministically(bad, eseed); } catch (const H::Error &e) { threw = std::string(e.what()) == "kem: invalid public key"; }
        if (!threw) { std::printf("FAILED low-order encapsulate\n"); return 1; }
    }
    {   // TestLowOrderX25519PointDecapsulate
        H::Bytes bad = ct;
        std::copy(low, low + 32, bad.end() - 32);
        bool threw = false;
        try { (void)H::Decapsulate(sk, bad); } catch (const H::Error &e) { threw = std::string(e.what()) == "kem: invalid public key"; }
        if (!threw) { std::printf("FAILED low-order decapsulate\n"); return 1; }
    }
    {   // wrong sizes: kem.ErrSeedSize / ErrCiphertextSize (hybrid.go:274-276, :303-305)
        bool a = false, b = false;
        try { (void)H::EncapsulateDeterministically(pk, H::Bytes(31)); } catch (const H::Error &) { a = true; }
        try { (void)H::Decapsulate(sk, H::Bytes(H::CiphertextSize + 1)); } catch (const H::Error &) { b = true; }
        if (!a || !b) { std::printf("FAILED size checks\n"); return 1; }
    }
    // batch with one bad item in the middle
    const size_t n = 333;  // several host threads for the X25519 half
    std::vector<uint8_t> seeds(H::SeedSize * n), es(H::EncapsulationSeedSize * n), pks(H::PublicKeySize * n), sks(H::PrivateKeySize * n),
        cts(H::CiphertextSize * n), s1(H::SharedKeySize * n), s2(H::SharedKeySize * n), st(n), st2(n);
    for (size_t i = 0; i < seeds.size(); i++) seeds[i] = (uint8_t)(i * 11 + 3);
    for (size_t i = 0; i < es.size(); i++) es[i] = (uint8_t)(i * 7 + 1);
    H::DeriveKeyPairBatch(seeds.data(), pks.data(), sks.data(), n);
    std::copy(low, low + 32, pks.begin() + H::PublicKeySize * 16 + H::MlkemEk);
    H::EncapsulateBatch(pks.data(), es.data(), cts.data(), s1.data(), st.data(), n);
    H::DecapsulateBatch(sks.data(), cts.data(), s2.data(), st2.data(), n);
    for (size_t i = 0; i < n; i++) {
        const bool bad = i == 16;
        if (st[i] != (bad ? H::ErrPubKey : H::Ok)) { std::printf("FAILED batch status %zu\n", i); return 1; }
        if (!bad && (st2[i] != 0 || !std::equal(s1.begin() + 64 * i, s1.begin() + 64 * i + 64, s2.begin() + 64 * i))) { std::printf("FAILED batch round trip %zu\n", i); return 1; }
    }
    for (size_t i : {size_t(0), size_t(200), n - 1}) {  // the batch forms equal the single-shot forms item by item
        const H::Bytes sd(seeds.begin() + H::SeedSize * i, seeds.begin() + H::SeedSize * (i + 1)), e(es.begin() + H::EncapsulationSeedSize * i, es.begin() + H::EncapsulationSeedSize * (i + 1));
        auto [pk1, sk1] = H::DeriveKeyPair(sd);
        auto [ct1, ss1] = H::EncapsulateDeterministically(pk1, e);
        if (!std::equal(pk1.begin(), pk1.end(), pks.begin() + H::PublicKeySize * i) || !std::equal(sk1.begin(), sk1.end(), sks.begin() + H::PrivateKeySize * i) ||
            !std::equal(ct1.begin(), ct1.end(), cts.begin() + H::CiphertextSize * i) || !std::equal(ss1.begin(), ss1.end(), s1.begin() + H::SharedKeySize * i)) {
            std::printf("FAILED batch item %zu differs from single-shot\n", i); return 1;
        }
    }
    std::printf("OK\n");
    return 0;
}
