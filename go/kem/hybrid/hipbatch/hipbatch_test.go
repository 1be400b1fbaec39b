//go:build cgo && hip

package hipbatch

// The hybrids against CIRCL's own kem/hybrid and kem/xwing on the same seeds, byte for byte (the Go-side counterpart of
// tests/test_gpu_hybrid.py; shape of kem/schemes/schemes_test.go:53-140).
//
//	go test -tags hip ./kem/hybrid/hipbatch/
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain on any box); tools/gocheck.py checks it statically.

import (
	"bytes"
	"testing"

	circl "github.com/cloudflare/circl/kem/schemes"
	"github.com/cloudflare/circl/xof"
)

func fill(label string, n int) []byte {
	h := xof.SHAKE128.New()
	_, _ = h.Write([]byte(label))
	out := make([]byte, n)
	_, _ = h.Read(out)
	return out
}

func row(flat []byte, size, i int) []byte { return flat[i*size : (i+1)*size] }

func TestHybridsAgainstCIRCL(t *testing.T) {
	for _, name := range []string{"X-Wing", "X25519MLKEM768", "Kyber768-X25519", "Kyber512-X25519"} {
		t.Run(name, func(t *testing.T) {
			s := circl.ByName(name)
			if s == nil {
				t.Fatal("unknown scheme")
			}
			const n = 100
			kseeds := fill("keygen/"+name, n*s.SeedSize())
			eseeds := fill("encaps/"+name, n*s.EncapsulationSeedSize())
			pks, sks, err := DeriveKeyPairBatch(s, kseeds, 0)
			if err != nil {
				t.Fatal(err)
			}
			cts, sss, errs, err := EncapsulateBatch(s, pks, eseeds, 0)
			if err != nil {
				t.Fatal(err)
			}
			back, derrs, err := DecapsulateBatch(s, sks, cts, 0)
			if err != nil {
				t.Fatal(err)
			}
			for i := 0; i < n; i++ {
				pk, sk := s.DeriveKeyPair(row(kseeds, s.SeedSize(), i))
				pb, _ := pk.MarshalBinary()
				sb, _ := sk.MarshalBinary()
				if !bytes.Equal(pb, row(pks, s.PublicKeySize(), i)) || !bytes.Equal(sb, row(sks, s.PrivateKeySize(), i)) {
					t.Fatalf("key pair %d differs", i)
				}
				ct, ss, err := s.EncapsulateDeterministically(pk, row(eseeds, s.EncapsulationSeedSize(), i))
				if err != nil || errs[i] != nil || derrs[i] != nil {
					t.Fatal(err, errs[i], derrs[i])
				}
				if !bytes.Equal(ct, row(cts, s.CiphertextSize(), i)) || !bytes.Equal(ss, row(sss, s.SharedKeySize(), i)) {
					t.Fatalf("encapsulation %d differs", i)
				}
				if !bytes.Equal(ss, row(back, s.SharedKeySize(), i)) {
					t.Fatalf("round trip %d", i)
				}
				ss2, err := s.Decapsulate(sk, row(cts, s.CiphertextSize(), i))
				if err != nil || !bytes.Equal(ss2, ss) {
					t.Fatalf("CIRCL decapsulates item %d differently", i)
				}
			}
		})
	}
}
