//go:build cgo && hip

package hipbatch

// The parity test a CIRCL maintainer runs after wiring libcirclhip.so in: every key, ciphertext and shared secret a batch
// produces equals what CIRCL's own scheme (kem/mlkem, kem/kyber: the reference) produces from the same seeds, byte for
// byte -- the Go-side counterpart of this repository's GPU-vs-oracle tests (tests/test_gpu_mlkem.py), in the shape of
// kem/schemes/schemes_test.go:53-140.
//
//	go test -tags hip ./kem/mlkem/hipbatch/ -run . -bench Batch
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain on any box); tools/gocheck.py checks it statically.

import (
	"bytes"
	"fmt"
	"testing"

	"github.com/cloudflare/circl/kem"
	"github.com/cloudflare/circl/kem/schemes"
	"github.com/cloudflare/circl/xof"
)

var names = []string{"ML-KEM-512", "ML-KEM-768", "ML-KEM-1024", "Kyber512", "Kyber768", "Kyber1024"}

// fill expands a label into n deterministic bytes (SHAKE128, as the reference's own tests do through internal/nist or xof).
func fill(label string, n int) []byte {
	h := xof.SHAKE128.New()
	_, _ = h.Write([]byte(label))
	out := make([]byte, n)
	_, _ = h.Read(out)
	return out
}

func row(flat []byte, size, i int) []byte { return flat[i*size : (i+1)*size] }

func TestBatchAgainstCIRCL(t *testing.T) {
	for _, name := range names {
		t.Run(name, func(t *testing.T) {
			s := schemes.ByName(name)
			if s == nil {
				t.Fatal("unknown scheme")
			}
			const n = 300
			kseeds := fill("keygen/"+name, n*s.SeedSize())
			eseeds := fill("encaps/"+name, n*s.EncapsulationSeedSize())
			eks, dks, err := DeriveKeyPairBatch(s, kseeds, 0)
			if err != nil {
				t.Fatal(err)
			}
			cts, sss, errs, err := EncapsulateBatch(s, eks, eseeds, 0)
			if err != nil {
				t.Fatal(err)
			}
			pks := make([]kem.PublicKey, n)
			sks := make([]kem.PrivateKey, n)
			for i := 0; i < n; i++ {
				pk, sk := s.DeriveKeyPair(row(kseeds, s.SeedSize(), i))
				pks[i], sks[i] = pk, sk
				ek, _ := pk.MarshalBinary()
				dk, _ := sk.MarshalBinary()
				if !bytes.Equal(ek, row(eks, s.PublicKeySize(), i)) || !bytes.Equal(dk, row(dks, s.PrivateKeySize(), i)) {
					t.Fatalf("key pair %d differs", i)
				}
				ct, ss, err := s.EncapsulateDeterministically(pk, row(eseeds, s.EncapsulationSeedSize(), i))
				if err != nil || errs[i] != nil {
					t.Fatal(err, errs[i])
				}
				if !bytes.Equal(ct, row(cts, s.CiphertextSize(), i)) || !bytes.Equal(ss, row(sss, s.SharedKeySize(), i)) {
					t.Fatalf("encapsulation %d differs", i)
				}
			}
			// decapsulation, every third ciphertext corrupted: implicit rejection must agree as well (kyber.go:144-184)
			bad := append([]byte(nil), cts...)
			for i := 0; i < n; i += 3 {
				bad[i*s.CiphertextSize()+7] ^= 0x10
			}
			got, derrs, err := DecapsulateBatch(s, dks, bad, 0)
			if err != nil {
				t.Fatal(err)
			}
			for i := 0; i < n; i++ {
				want, err := s.Decapsulate(sks[i], row(bad, s.CiphertextSize(), i))
				if err != nil || derrs[i] != nil {
					t.Fatal(err, derrs[i])
				}
				if !bytes.Equal(want, row(got, s.SharedKeySize(), i)) {
					t.Fatalf("decapsulation %d differs", i)
				}
				if i%3 != 0 && !bytes.Equal(want, row(sss, s.SharedKeySize(), i)) {
					t.Fatalf("round trip %d", i)
				}
			}
			// one key for the whole batch, and the same key resident on the device
			one, _, err := EncapsulateSharedKeyBatch(pks[0], eseeds, 0)
			if err != nil {
				t.Fatal(err)
			}
			ct0, _, _ := s.EncapsulateDeterministically(pks[0], row(eseeds, s.EncapsulationSeedSize(), 5))
			if !bytes.Equal(ct0, row(one, s.CiphertextSize(), 5)) {
				t.Fatal("shared-key encapsulation differs")
			}
		})
	}
}

func TestWrappedScheme(t *testing.T) {
	s := ByName("ML-KEM-768")
	if s == nil {
		t.Fatal("no kernels for ML-KEM-768")
	}
	const n = 64
	seeds := make([][]byte, n)
	for i := range seeds {
		seeds[i] = fill(fmt.Sprintf("wrapped/%d", i), s.SeedSize())
	}
	pks, sks, err := s.DeriveKeyPairBatch(seeds, AllDevices)
	if err != nil {
		t.Fatal(err)
	}
	cts, sss, errs, err := s.EncapsulateBatch(pks, AllDevices)
	if err != nil {
		t.Fatal(err)
	}
	back, derrs, err := s.DecapsulateBatch(sks, cts, AllDevices)
	if err != nil {
		t.Fatal(err)
	}
	for i := 0; i < n; i++ {
		if errs[i] != nil || derrs[i] != nil || !bytes.Equal(sss[i], back[i]) {
			t.Fatalf("round trip %d", i)
		}
		ss, err := s.Decapsulate(sks[i], cts[i]) // CIRCL's own path on the GPU's ciphertext
		if err != nil || !bytes.Equal(ss, sss[i]) {
			t.Fatalf("CIRCL decapsulates item %d differently", i)
		}
	}
	// a resident private key: a TLS server's static key
	rk, err := s.ResidentPrivateKey(sks[0], 0)
	if err != nil {
		t.Fatal(err)
	}
	defer rk.Close()
	got, err := rk.DecapsulateBatch(cts[0])
	if err != nil || !bytes.Equal(got, sss[0]) {
		t.Fatal("resident key decapsulation differs", err)
	}
}

func TestInvalidPublicKeyIsReportedPerItem(t *testing.T) {
	s := schemes.ByName("ML-KEM-768")
	eks, _, err := DeriveKeyPairBatch(s, fill("inv", 4*s.SeedSize()), 0)
	if err != nil {
		t.Fatal(err)
	}
	eks[2*s.PublicKeySize()] = 0xff // a coefficient >= q in key 2 (pke/kyber/kyber768/internal/cpapke.go:45-55)
	eks[2*s.PublicKeySize()+1] |= 0x0f
	_, _, errs, err := EncapsulateBatch(s, eks, fill("inv/m", 4*s.EncapsulationSeedSize()), 0)
	if err != nil {
		t.Fatal(err)
	}
	for i, e := range errs {
		if (i == 2) != (e == kem.ErrPubKey) {
			t.Fatalf("item %d: %v", i, e)
		}
	}
	if _, err := s.UnmarshalBinaryPublicKey(row(eks, s.PublicKeySize(), 2)); err != kem.ErrPubKey {
		t.Fatal("CIRCL accepts the key the GPU rejects")
	}
}

// BenchmarkEncapsulateBatch is kem/schemes/schemes_test.go:28-38 over batches: one op = one batch of 2^16 encapsulations to
// distinct keys (b.N batches), to be read against BenchmarkEncapsulate/ML-KEM-768 x 65536.
func BenchmarkEncapsulateBatch(b *testing.B) {
	s := schemes.ByName("ML-KEM-768")
	const n = 1 << 16
	eks, _, err := DeriveKeyPairBatch(s, fill("bench", n*s.SeedSize()), AllDevices)
	if err != nil {
		b.Fatal(err)
	}
	seeds := fill("bench/m", n*s.EncapsulationSeedSize())
	b.SetBytes(int64(n))
	b.ResetTimer()
	for i := 0; i < b.N; i++ {
		if _, _, _, err := EncapsulateBatch(s, eks, seeds, AllDevices); err != nil {
			b.Fatal(err)
		}
	}
}
