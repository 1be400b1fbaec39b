// Reproduces kem/xwing/xwing_test.go:38-85 TestVectors with the C++ X-Wing layer: a SHAKE128 stream feeds
// three (seed, eseed) pairs, the formatted transcript is hashed with SHAKE128 and compared with the digest of
// the X-Wing draft's test-vectors.txt.  Also a batch round trip.  Needs a GPU; prints "OK".
#include <cstdio>
#include <string>

#include "circl/xwing.hpp"

static void write_hex(std::string &w, const char *prefix, const std::vector<uint8_t> &val) {
    static const char *d = "0123456789abcdef";
    std::string hex;
    for (uint8_t b : val) { hex += d[b >> 4]; hex += d[b & 15]; }
    const std::string indent = "  ";
    const size_t width = 74;
    const std::string p(prefix);
    if (p.size() + hex.size() + 5 < width) { w += p + "     " + hex + "\n"; return; }
    w += p + "\n";
    while (!hex.empty()) {
        if (hex.size() < width - indent.size()) { w += indent + hex + "\n"; hex.clear(); }
        else { w += indent + hex.substr(0, width - indent.size()) + "\n"; hex = hex.substr(width - indent.size()); }
    }
}

int main() {
    using namespace circl;
    uint8_t none = 0;
    std::vector<uint8_t> stream(3 * 96);
    if (circl_hip_shake(168, 0x1f, &none, 0, stream.data(), stream.size(), 1, 0) != 0) { std::printf("FAILED shake\n"); return 1; }
    std::string w;
    for (int i = 0; i < 3; i++) {
        xwing::Bytes seed(stream.begin() + 96 * i, stream.begin() + 96 * i + 32);
        xwing::Bytes eseed(stream.begin() + 96 * i + 32, stream.begin() + 96 * i + 96);
        write_hex(w, "seed", seed);
        auto [sk, pk] = xwing::DeriveKeyPairPacked(seed);
        write_hex(w, "sk", sk);
        write_hex(w, "pk", pk);
        write_hex(w, "eseed", eseed);
        auto [ss, ct] = xwing::Encapsulate(pk, eseed);
        write_hex(w, "ct", ct);
        write_hex(w, "ss", ss);
        if (xwing::Decapsulate(ct, sk) != ss) { std::printf("FAILED decapsulate\n"); return 1; }
        w += "\n";
    }
    uint8_t cs[32];
    if (circl_hip_shake(168, 0x1f, reinterpret_cast<const uint8_t *>(w.data()), w.size(), cs, 32, 1, 0) != 0) { std::printf("FAILED shake\n"); return 1; }
    std::string got;
    for (uint8_t b : cs) { char t[3]; std::snprintf(t, 3, "%02x", b); got += t; }
    const std::string want = "1bcd0057d861d6b866239936cadcaeee1ec0164dedc181c386e9e54fe46156fe";  // xwing_test.go:80
    if (got != want) { std::printf("FAILED transcript %s != %s\n", got.c_str(), want.c_str()); return 1; }
    // batch round trip
    const size_t n = 300;  // enough for the host-side X25519 loop to fan out over several threads
    std::vector<uint8_t> seeds(32 * n), es(64 * n), sks(32 * n), pks(xwing::PublicKeySize * n), ss1(32 * n), ss2(32 * n), cts(xwing::CiphertextSize * n), st(n);
    for (size_t i = 0; i < seeds.size(); i++) seeds[i] = (uint8_t)(i * 11 + 3);
    for (size_t i = 0; i < es.size(); i++) es[i] = (uint8_t)(i * 7 + 1);
    xwing::DeriveKeyPairBatch(seeds.data(), sks.data(), pks.data(), n);
    xwing::EncapsulateBatch(pks.data(), es.data(), ss1.data(), cts.data(), st.data(), n);
    xwing::DecapsulateBatch(cts.data(), sks.data(), ss2.data(), n);
    if (ss1 != ss2) { std::printf("FAILED batch round trip\n"); return 1; }
    for (size_t i : {size_t(0), size_t(77), n - 1}) {  // the batch forms equal the single-shot forms item by item
        const xwing::Bytes seed(seeds.begin() + 32 * i, seeds.begin() + 32 * (i + 1)), e(es.begin() + 64 * i, es.begin() + 64 * (i + 1));
        auto [sk1, pk1] = xwing::DeriveKeyPairPacked(seed);
        auto [s1, c1] = xwing::Encapsulate(pk1, e);
        if (!std::equal(pk1.begin(), pk1.end(), pks.begin() + xwing::PublicKeySize * i) || !std::equal(c1.begin(), c1.end(), cts.begin() + xwing::CiphertextSize * i) ||
            !std::equal(s1.begin(), s1.end(), ss1.begin() + 32 * i)) { std::printf("FAILED batch item %zu differs from single-shot\n", i); return 1; }
    }
    std::printf("OK\n");
    return 0;
}
