//go:build cgo && hip

package hipbatch

// The parity test a CIRCL maintainer runs after wiring libcirclhip.so in: keys, signatures and verdicts of a batch equal what
// CIRCL's own scheme (sign/mldsa, sign/dilithium: the reference) gives for the same inputs, byte for byte -- the Go-side
// counterpart of tests/test_gpu_mldsa.py, in the shape of sign/schemes/schemes_test.go.
//
//	go test -tags hip ./sign/mldsa/hipbatch/ -run . -bench Batch
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain on any box); tools/gocheck.py checks it statically.

import (
	"bytes"
	"fmt"
	"testing"

	"github.com/cloudflare/circl/sign"
	"github.com/cloudflare/circl/sign/schemes"
	"github.com/cloudflare/circl/xof"
)

var names = []string{"ML-DSA-44", "ML-DSA-65", "ML-DSA-87", "Dilithium2", "Dilithium3", "Dilithium5"}

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
			const n = 200
			seeds := fill("keygen/"+name, n*s.SeedSize())
			pks, sks, err := DeriveKeyBatch(s, seeds, 0)
			if err != nil {
				t.Fatal(err)
			}
			pkRows := make([][]byte, n)
			skRows := make([][]byte, n)
			msgs := make([][]byte, n)
			ctxs := make([]string, n)
			keys := make([]sign.PrivateKey, n)
			pubs := make([]sign.PublicKey, n)
			for i := 0; i < n; i++ {
				pk, sk := s.DeriveKey(row(seeds, s.SeedSize(), i))
				pubs[i], keys[i] = pk, sk
				pb, _ := pk.MarshalBinary()
				sb, _ := sk.MarshalBinary()
				if !bytes.Equal(pb, row(pks, s.PublicKeySize(), i)) || !bytes.Equal(sb, row(sks, s.PrivateKeySize(), i)) {
					t.Fatalf("key pair %d differs", i)
				}
				pkRows[i], skRows[i] = pb, sb
				msgs[i] = fill(fmt.Sprintf("msg/%s/%d", name, i), 1+37*i%900) // ragged, some longer than a SHAKE256 block
				if s.SupportsContext() {
					ctxs[i] = string(fill(fmt.Sprintf("ctx/%d", i), i%256))
				}
			}
			sigs, err := SignBatch(s, skRows, msgs, ctxs, nil, 0) // deterministic, as scheme.Sign
			if err != nil {
				t.Fatal(err)
			}
			for i := 0; i < n; i++ {
				want := s.Sign(keys[i], msgs[i], &sign.SignatureOpts{Context: ctxs[i]})
				if !bytes.Equal(want, sigs[i]) {
					t.Fatalf("signature %d differs", i)
				}
			}
			// verification: valid, corrupted, and a signature of the wrong length
			bad := make([][]byte, n)
			for i := range bad {
				bad[i] = append([]byte(nil), sigs[i]...)
				if i%4 == 1 {
					bad[i][11] ^= 0x20
				}
			}
			bad[6] = bad[6][:len(bad[6])-1]
			ok, err := VerifyBatch(s, pkRows, msgs, bad, ctxs, 0)
			if err != nil {
				t.Fatal(err)
			}
			for i := 0; i < n; i++ {
				want := s.Verify(pubs[i], msgs[i], bad[i], &sign.SignatureOpts{Context: ctxs[i]})
				if ok[i] != want || want != (i%4 != 1 && i != 6) {
					t.Fatalf("verdict %d: gpu %v, circl %v", i, ok[i], want)
				}
			}
			// the same through resident keys (parsed once, used call after call)
			pubTable, err := NewResidentPublicKeys(s, pks, 0)
			if err != nil {
				t.Fatal(err)
			}
			defer pubTable.Close()
			prvTable, err := NewResidentPrivateKeys(s, sks, 0)
			if err != nil {
				t.Fatal(err)
			}
			defer prvTable.Close()
			idx := make([]uint32, n)
			for i := range idx {
				idx[i] = uint32(i)
			}
			sigs2, err := prvTable.Sign(idx, msgs, ctxs, nil)
			if err != nil {
				t.Fatal(err)
			}
			flat := make([]byte, 0, n*s.SignatureSize())
			for i := 0; i < n; i++ {
				if !bytes.Equal(sigs2[i], sigs[i]) {
					t.Fatalf("prepared-key signature %d differs", i)
				}
				flat = append(flat, sigs[i]...)
			}
			ok2, err := pubTable.Verify(idx, msgs, flat, ctxs)
			if err != nil {
				t.Fatal(err)
			}
			for i, v := range ok2 {
				if !v {
					t.Fatalf("resident-key verification %d", i)
				}
			}
		})
	}
}

func TestContextRules(t *testing.T) {
	s := schemes.ByName("ML-DSA-65")
	pks, sks, err := DeriveKeyBatch(s, fill("ctx", s.SeedSize()), 0)
	if err != nil {
		t.Fatal(err)
	}
	long := string(make([]byte, 256))
	if _, err := SignBatch(s, [][]byte{sks}, [][]byte{[]byte("m")}, []string{long}, nil, 0); err != sign.ErrContextTooLong {
		t.Fatal("a 256-byte context must be refused (mldsa65/dilithium.go:63-65)", err)
	}
	sig, err := SignBatch(s, [][]byte{sks}, [][]byte{[]byte("m")}, nil, nil, 0)
	if err != nil {
		t.Fatal(err)
	}
	ok, err := VerifyBatch(s, [][]byte{pks}, [][]byte{[]byte("m")}, sig, []string{long}, 0)
	if err != nil || ok[0] {
		t.Fatal("a 256-byte context never verifies (dilithium.go:116-118)", err)
	}
}

// BenchmarkVerifyBatch: one op = one batch of 2^14 verifications under distinct keys.
func BenchmarkVerifyBatch(b *testing.B) {
	s := schemes.ByName("ML-DSA-65")
	const n = 1 << 14
	pks, sks, err := DeriveKeyBatch(s, fill("bench", n*s.SeedSize()), AllDevices)
	if err != nil {
		b.Fatal(err)
	}
	pkRows := make([][]byte, n)
	skRows := make([][]byte, n)
	msgs := make([][]byte, n)
	for i := range msgs {
		pkRows[i], skRows[i] = row(pks, s.PublicKeySize(), i), row(sks, s.PrivateKeySize(), i)
		msgs[i] = fill(fmt.Sprintf("bench/%d", i), 32)
	}
	sigs, err := SignBatch(s, skRows, msgs, nil, nil, AllDevices)
	if err != nil {
		b.Fatal(err)
	}
	b.ResetTimer()
	for i := 0; i < b.N; i++ {
		if _, err := VerifyBatch(s, pkRows, msgs, sigs, nil, AllDevices); err != nil {
			b.Fatal(err)
		}
	}
}
