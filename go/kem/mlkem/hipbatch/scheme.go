//go:build cgo && hip

package hipbatch

// A kem.Scheme that drops in behind kem/mlkem (and kem/kyber): every method of kem.Scheme (kem/kem.go:33-82) is
// CIRCL's own -- single operations stay on the Go path, byte for byte what the wrapped scheme does -- and the batch
// methods below route whole batches to libcirclhip.so.  The parity tests of this repository establish that both paths
// give identical bytes, so a caller may mix them freely.
//
//	s := hipbatch.ByName("ML-KEM-768")        // or hipbatch.Wrap(mlkem768.Scheme())
//	pk, sk, _ := s.GenerateKeyPair()          // CIRCL, unchanged
//	cts, sss, errs, err := s.EncapsulateBatch(eks, seeds, hipbatch.AllDevices)
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain in the build image); tests/cgo_shape_test.c drives the same C
// entry points with exactly the argument shapes these methods produce.

import (
	"crypto/rand"

	"github.com/cloudflare/circl/kem"
	"github.com/cloudflare/circl/kem/schemes"
)

// Scheme is kem.Scheme plus batch operations.  The embedded scheme is the stateless singleton CIRCL hands out
// (kem/mlkem/mlkem768/kyber.go:267-272), so a Scheme is as cheap to copy and as safe for concurrent use.
type Scheme struct {
	kem.Scheme
}

// Wrap returns s with batch operations, or nil if libcirclhip.so has no kernels for it.
func Wrap(s kem.Scheme) *Scheme {
	if s == nil {
		return nil
	}
	if _, ok := params[s.Name()]; !ok {
		if _, ok3 := paramsRound3[s.Name()]; !ok3 {
			return nil
		}
	}
	return &Scheme{s}
}

// ByName looks the scheme up in CIRCL's registry (kem/schemes/schemes.go:35-55): "ML-KEM-512", "ML-KEM-768",
// "ML-KEM-1024", "Kyber512", "Kyber768", "Kyber1024".
func ByName(name string) *Scheme { return Wrap(schemes.ByName(name)) }

// EncapsulateBatch draws the encapsulation seeds from crypto/rand exactly as scheme.Encapsulate does per call
// (kyber.go:104-108) and encapsulates to every public key of pks.
func (s *Scheme) EncapsulateBatch(pks []kem.PublicKey, device int) (cts, sss [][]byte, errs []error, err error) {
	seeds := make([]byte, len(pks)*s.EncapsulationSeedSize())
	if _, err = rand.Read(seeds); err != nil {
		return nil, nil, nil, err
	}
	return s.EncapsulateDeterministicallyBatch(pks, seeds, device)
}

// EncapsulateDeterministicallyBatch is n times scheme.EncapsulateDeterministically(pks[i], seeds[i]) (kyber.go:359-370).
// A key of another scheme gives kem.ErrTypeMismatch for the whole call, as the single call does (kyber.go:351-354).
func (s *Scheme) EncapsulateDeterministicallyBatch(pks []kem.PublicKey, seeds []byte, device int) (cts, sss [][]byte, errs []error, err error) {
	n := len(pks)
	eks := make([]byte, 0, n*s.PublicKeySize())
	for _, pk := range pks {
		if pk.Scheme().Name() != s.Name() {
			return nil, nil, nil, kem.ErrTypeMismatch
		}
		b, e := pk.MarshalBinary()
		if e != nil {
			return nil, nil, nil, e
		}
		eks = append(eks, b...)
	}
	ct, ss, errs, err := EncapsulateBatch(s.Scheme, eks, seeds, device)
	if err != nil {
		return nil, nil, nil, err
	}
	return rows(ct, s.CiphertextSize()), rows(ss, s.SharedKeySize()), errs, nil
}

// DecapsulateBatch is n times scheme.Decapsulate(sks[i], cts[i]) (kyber.go:376-386): a ciphertext of the wrong length
// gives kem.ErrCiphertextSize for the whole call; an invalid ciphertext is not an error (implicit rejection).
func (s *Scheme) DecapsulateBatch(sks []kem.PrivateKey, cts [][]byte, device int) (sss [][]byte, errs []error, err error) {
	n := len(sks)
	if len(cts) != n {
		return nil, nil, kem.ErrCiphertextSize
	}
	dks := make([]byte, 0, n*s.PrivateKeySize())
	defer func() { clear(dks[:cap(dks)]) }() // the marshalled private keys are this wrapper's own copies: they do not outlive the call
	ctRows := make([]byte, 0, n*s.CiphertextSize())
	for i, sk := range sks {
		if sk.Scheme().Name() != s.Name() {
			return nil, nil, kem.ErrTypeMismatch
		}
		if len(cts[i]) != s.CiphertextSize() {
			return nil, nil, kem.ErrCiphertextSize
		}
		b, e := sk.MarshalBinary()
		if e != nil {
			return nil, nil, e
		}
		dks = append(dks, b...)
		clear(b)
		ctRows = append(ctRows, cts[i]...)
	}
	ss, errs, err := DecapsulateBatch(s.Scheme, dks, ctRows, device)
	if err != nil {
		return nil, nil, err
	}
	return rows(ss, s.SharedKeySize()), errs, nil
}

// DeriveKeyPairBatch is n times scheme.DeriveKeyPair(seeds[i]) followed by scheme.UnmarshalBinary*Key, so that the
// caller gets CIRCL's own key objects back.
func (s *Scheme) DeriveKeyPairBatch(seeds [][]byte, device int) (pks []kem.PublicKey, sks []kem.PrivateKey, err error) {
	flat := make([]byte, 0, len(seeds)*s.SeedSize())
	for _, sd := range seeds {
		if len(sd) != s.SeedSize() {
			panic(kem.ErrSeedSize) // kyber.go:341-343
		}
		flat = append(flat, sd...)
	}
	eks, dks, err := DeriveKeyPairBatch(s.Scheme, flat, device)
	if err != nil {
		return nil, nil, err
	}
	for i := range seeds {
		pk, e := s.UnmarshalBinaryPublicKey(eks[i*s.PublicKeySize() : (i+1)*s.PublicKeySize()])
		if e != nil {
			return nil, nil, e
		}
		sk, e := s.UnmarshalBinaryPrivateKey(dks[i*s.PrivateKeySize() : (i+1)*s.PrivateKeySize()])
		if e != nil {
			return nil, nil, e
		}
		pks, sks = append(pks, pk), append(sks, sk)
	}
	return
}

// KeyTable is a set of parsed public keys a batch refers to by index: what CIRCL's PublicKey object caches per key
// (A^T and H(ek), kyber.go:39-43) is expanded once per table entry on the device, however many items use it.
type KeyTable struct {
	scheme *Scheme
	rows   []byte // MarshalBinary forms, row-major
	n      int
}

func (s *Scheme) NewPublicKeyTable(pks []kem.PublicKey) (*KeyTable, error) {
	t := &KeyTable{scheme: s, n: len(pks)}
	for _, pk := range pks {
		if pk.Scheme().Name() != s.Name() {
			return nil, kem.ErrTypeMismatch
		}
		b, err := pk.MarshalBinary()
		if err != nil {
			return nil, err
		}
		t.rows = append(t.rows, b...)
	}
	return t, nil
}

// EncapsulateKeyedBatch: item i encapsulates seeds[i] to table entry idx[i].
func (t *KeyTable) EncapsulateKeyedBatch(idx []uint32, seeds []byte, device int) (cts, sss [][]byte, errs []error, err error) {
	s := t.scheme
	ct, ss, errs, err := EncapsulateKeyedBatch(s.Scheme, t.rows, idx, seeds, device)
	if err != nil {
		return nil, nil, nil, err
	}
	return rows(ct, s.CiphertextSize()), rows(ss, s.SharedKeySize()), errs, nil
}

func rows(flat []byte, size int) [][]byte {
	out := make([][]byte, 0, len(flat)/size)
	for i := 0; i+size <= len(flat); i += size {
		out = append(out, flat[i:i+size:i+size])
	}
	return out
}

// ResidentPublicKey / ResidentPrivateKey keep a CIRCL key object together with its GPU-side counterpart (keytable.go): the
// object answers every kem.PublicKey / kem.PrivateKey method as before, and batches addressed to it move only seeds,
// ciphertexts and secrets.
//
//	rpk, _ := s.ResidentPublicKey(pk, 0)          // parse on device 0 (A^T, H(ek) stay there)
//	cts, sss, _ := rpk.EncapsulateBatch(seeds)    // len(seeds) / 32 encapsulations to that key, ≈30 µs per call up to 1 024
type ResidentPublicKey struct {
	kem.PublicKey
	table *ResidentTable
	rx    *reactor // serving.go: the goroutine that owns the table's asynchronous queue (nil: none)
}
type ResidentPrivateKey struct {
	kem.PrivateKey
	table *ResidentTable
	rx    *reactor
}

// ResidentPublicKey parses pk on `device` (ML-KEM only: round-3 Kyber has no table route).
func (s *Scheme) ResidentPublicKey(pk kem.PublicKey, device int) (*ResidentPublicKey, error) {
	b, err := pk.MarshalBinary()
	if err != nil {
		return nil, err
	}
	t, err := NewResidentPublicKeys(s.Scheme, b, device)
	if err != nil {
		return nil, err
	}
	return &ResidentPublicKey{pk, t, nil}, nil
}

// ResidentPrivateKey parses sk on `device`; kem.ErrPrivKey if its stored hash does not match (kyber.go:219-228).
func (s *Scheme) ResidentPrivateKey(sk kem.PrivateKey, device int) (*ResidentPrivateKey, error) {
	b, err := sk.MarshalBinary()
	if err != nil {
		return nil, err
	}
	defer clear(b) // the marshalled copy of the private key does not outlive the call
	t, errs, err := NewResidentPrivateKeys(s.Scheme, b, device)
	if err != nil {
		return nil, err
	}
	if errs[0] != nil {
		t.Close()
		return nil, errs[0]
	}
	return &ResidentPrivateKey{sk, t, nil}, nil
}

// EncapsulateBatch is len(seeds)/EncapsulationSeedSize times EncapsulateDeterministically to this key.
func (k *ResidentPublicKey) EncapsulateBatch(seeds []byte) (cts, sss []byte, err error) {
	cts, sss, errs, err := k.table.Encapsulate(nil, seeds)
	if err != nil {
		return nil, nil, err
	}
	for _, e := range errs {
		if e != nil {
			return nil, nil, e
		}
	}
	return cts, sss, nil
}

// DecapsulateBatch is len(cts)/CiphertextSize times Decapsulate with this key.
func (k *ResidentPrivateKey) DecapsulateBatch(cts []byte) (sss []byte, err error) {
	sss, _, err = k.table.Decapsulate(nil, cts)
	return sss, err
}

// Close releases the device-side halves (also done by the tables' finalizers).
func (k *ResidentPublicKey) Close() {
	if k.rx != nil {
		k.rx.stop()
		k.rx = nil
	}
	k.table.Close()
}
func (k *ResidentPrivateKey) Close() {
	if k.rx != nil {
		k.rx.stop()
		k.rx = nil
	}
	k.table.Close()
}
