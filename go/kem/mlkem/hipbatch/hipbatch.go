//go:build cgo && hip

// Package hipbatch is the cgo bridge a CIRCL maintainer would add next to kem/mlkem to route
// batches of ML-KEM operations to libcirclhip.so (MI355X).  It keeps kem.Scheme untouched and
// adds batch calls on top of it.
//
// NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Go toolchain.  The same C symbols
// are exercised through ctypes (tests/) and C++ (include/circl/kem.hpp).  See INTEGRATION.md.
package hipbatch

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../include
#cgo LDFLAGS: -lcirclhip
#include <circl_hip.h>
*/
import "C"

import (
	"errors"
	"fmt"
	"unsafe"

	"github.com/cloudflare/circl/kem"
)

// AllDevices splits a batch into contiguous shards, one per visible GPU (no collective).
const AllDevices = -1

var params = map[string]C.int{"ML-KEM-512": 512, "ML-KEM-768": 768, "ML-KEM-1024": 1024}

// round-3 Kyber (kem/kyber/kyber{512,768,1024}) goes through circl_hip_kyber_*: same sizes, different hashing,
// no per-item failure (kem/kyber/kyber768/kyber.go:215-262).
var paramsRound3 = map[string]C.int{"Kyber512": 512, "Kyber768": 768, "Kyber1024": 1024}

func status(code C.int, where string) error {
	if code == 0 {
		return nil
	}
	return fmt.Errorf("circl-hip %s: error %d: %s", where, int(code), C.GoString(C.circl_hip_last_error()))
}

func itemErr(st byte) error {
	switch st {
	case 0:
		return nil
	case 1:
		return kem.ErrPubKey // pke/kyber/kyber768/internal/cpapke.go:45-55
	case 2:
		return kem.ErrPrivKey // kem/mlkem/mlkem768/kyber.go:219-228
	}
	return errors.New("circl-hip: unknown item status")
}

func ptr(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// EncapsulateBatch is n times scheme.UnmarshalBinaryPublicKey + scheme.EncapsulateDeterministically
// (kem/mlkem/mlkem768/kyber.go:390-396, :359-370).  eks, seeds are row-major [n][PublicKeySize],
// [n][EncapsulationSeedSize]; the returned cts, sss are [n][CiphertextSize], [n][SharedKeySize].
// errs[i] is nil, kem.ErrPubKey, ... per item; the Go slices are only borrowed for the call.
func EncapsulateBatch(s kem.Scheme, eks, seeds []byte, device int) (cts, sss []byte, errs []error, err error) {
	p, ok := params[s.Name()]
	p3, ok3 := paramsRound3[s.Name()]
	if !ok && !ok3 {
		return nil, nil, nil, kem.ErrTypeMismatch
	}
	if len(eks)%s.PublicKeySize() != 0 {
		return nil, nil, nil, kem.ErrPubKeySize
	}
	n := len(eks) / s.PublicKeySize()
	if len(seeds) != n*s.EncapsulationSeedSize() {
		return nil, nil, nil, kem.ErrSeedSize
	}
	cts = make([]byte, n*s.CiphertextSize())
	sss = make([]byte, n*s.SharedKeySize())
	st := make([]byte, n)
	if ok3 {
		err = status(C.circl_hip_kyber_encaps(p3, ptr(eks), ptr(seeds), ptr(cts), ptr(sss), C.size_t(n), C.int(device)), "kyber encaps")
	} else {
		err = status(C.circl_hip_mlkem_encaps(p, ptr(eks), ptr(seeds), ptr(cts), ptr(sss), ptr(st), C.size_t(n), C.int(device)), "encaps")
	}
	if err != nil {
		return nil, nil, nil, err
	}
	errs = make([]error, n)
	for i, c := range st {
		errs[i] = itemErr(c)
	}
	return
}

// EncapsulateSharedKeyBatch is n times scheme.EncapsulateDeterministically on ONE public key (the cached-key case,
// kem/mlkem/mlkem768/kyber.go:39-43): the device derives A^T and H(ek) once for the whole batch.
func EncapsulateSharedKeyBatch(pk kem.PublicKey, seeds []byte, device int) (cts, sss []byte, err error) {
	s := pk.Scheme()
	p, ok := params[s.Name()]
	if !ok {
		return nil, nil, kem.ErrTypeMismatch
	}
	if len(seeds)%s.EncapsulationSeedSize() != 0 {
		return nil, nil, kem.ErrSeedSize
	}
	n := len(seeds) / s.EncapsulationSeedSize()
	ek, err := pk.MarshalBinary()
	if err != nil {
		return nil, nil, err
	}
	cts = make([]byte, n*s.CiphertextSize())
	sss = make([]byte, n*s.SharedKeySize())
	st := make([]byte, n)
	if err = status(C.circl_hip_mlkem_encaps_shared(p, ptr(ek), ptr(seeds), ptr(cts), ptr(sss), ptr(st), C.size_t(n), C.int(device)), "encaps shared"); err != nil {
		return nil, nil, err
	}
	if n > 0 && st[0] != 0 {
		return nil, nil, itemErr(st[0])
	}
	return
}

// DecapsulateBatch is n times scheme.UnmarshalBinaryPrivateKey + scheme.Decapsulate
// (kyber.go:398-407, :376-386).  An invalid ciphertext is not an error (implicit rejection).
func DecapsulateBatch(s kem.Scheme, dks, cts []byte, device int) (sss []byte, errs []error, err error) {
	p, ok := params[s.Name()]
	p3, ok3 := paramsRound3[s.Name()]
	if !ok && !ok3 {
		return nil, nil, kem.ErrTypeMismatch
	}
	if len(dks)%s.PrivateKeySize() != 0 {
		return nil, nil, kem.ErrPrivKeySize
	}
	n := len(dks) / s.PrivateKeySize()
	if len(cts) != n*s.CiphertextSize() {
		return nil, nil, kem.ErrCiphertextSize
	}
	sss = make([]byte, n*s.SharedKeySize())
	st := make([]byte, n)
	if ok3 {
		err = status(C.circl_hip_kyber_decaps(p3, ptr(dks), ptr(cts), ptr(sss), C.size_t(n), C.int(device)), "kyber decaps")
	} else {
		err = status(C.circl_hip_mlkem_decaps(p, ptr(dks), ptr(cts), ptr(sss), ptr(st), C.size_t(n), C.int(device)), "decaps")
	}
	if err != nil {
		return nil, nil, err
	}
	errs = make([]error, n)
	for i, c := range st {
		errs[i] = itemErr(c)
	}
	return
}

// DeriveKeyPairBatch is n times scheme.DeriveKeyPair (kyber.go:340-345); seeds are [n][SeedSize].
func DeriveKeyPairBatch(s kem.Scheme, seeds []byte, device int) (eks, dks []byte, err error) {
	p, ok := params[s.Name()]
	p3, ok3 := paramsRound3[s.Name()]
	if !ok && !ok3 {
		return nil, nil, kem.ErrTypeMismatch
	}
	if len(seeds)%s.SeedSize() != 0 {
		panic(kem.ErrSeedSize) // DeriveKeyPair panics on a bad seed length (kyber.go:341-343)
	}
	n := len(seeds) / s.SeedSize()
	eks = make([]byte, n*s.PublicKeySize())
	dks = make([]byte, n*s.PrivateKeySize())
	if ok3 {
		err = status(C.circl_hip_kyber_keygen(p3, ptr(seeds), ptr(eks), ptr(dks), C.size_t(n), C.int(device)), "kyber keygen")
	} else {
		err = status(C.circl_hip_mlkem_keygen(p, ptr(seeds), ptr(eks), ptr(dks), C.size_t(n), C.int(device)), "keygen")
	}
	return
}

func ptr32(b []uint32) *C.uint32_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&b[0]))
}

// EncapsulateKeyedBatch: a table of nkeys public keys (row-major MarshalBinary forms) and one index per item.  The
// device expands A^T and H(ek) once per TABLE ENTRY -- what UnmarshalBinaryPublicKey caches in the key object
// (kem/mlkem/mlkem768/kyber.go:39-43, :247-263) -- and then does the per-message work of EncapsulateDeterministically.
func EncapsulateKeyedBatch(s kem.Scheme, ekTable []byte, idx []uint32, seeds []byte, device int) (cts, sss []byte, errs []error, err error) {
	p, ok := params[s.Name()]
	if !ok {
		return nil, nil, nil, kem.ErrTypeMismatch
	}
	if len(ekTable) == 0 || len(ekTable)%s.PublicKeySize() != 0 {
		return nil, nil, nil, kem.ErrPubKeySize
	}
	n := len(idx)
	if len(seeds) != n*s.EncapsulationSeedSize() {
		return nil, nil, nil, kem.ErrSeedSize
	}
	nkeys := len(ekTable) / s.PublicKeySize()
	cts = make([]byte, n*s.CiphertextSize())
	sss = make([]byte, n*s.SharedKeySize())
	st := make([]byte, n)
	if err = status(C.circl_hip_mlkem_encaps_keyed(p, ptr(ekTable), C.size_t(nkeys), ptr32(idx), ptr(seeds), ptr(cts), ptr(sss), ptr(st),
		C.size_t(n), C.int(device)), "encaps keyed"); err != nil {
		return nil, nil, nil, err
	}
	errs = make([]error, n)
	for i, c := range st {
		errs[i] = itemErr(c)
	}
	return
}

// DecapsulateKeyedBatch: a table of private keys and one index per ciphertext (a server with a handful of static keys).
func DecapsulateKeyedBatch(s kem.Scheme, dkTable []byte, idx []uint32, cts []byte, device int) (sss []byte, errs []error, err error) {
	p, ok := params[s.Name()]
	if !ok {
		return nil, nil, kem.ErrTypeMismatch
	}
	if len(dkTable) == 0 || len(dkTable)%s.PrivateKeySize() != 0 {
		return nil, nil, kem.ErrPrivKeySize
	}
	n := len(idx)
	if len(cts) != n*s.CiphertextSize() {
		return nil, nil, kem.ErrCiphertextSize
	}
	nkeys := len(dkTable) / s.PrivateKeySize()
	sss = make([]byte, n*s.SharedKeySize())
	st := make([]byte, n)
	if err = status(C.circl_hip_mlkem_decaps_keyed(p, ptr(dkTable), C.size_t(nkeys), ptr32(idx), ptr(cts), ptr(sss), ptr(st), C.size_t(n),
		C.int(device)), "decaps keyed"); err != nil {
		return nil, nil, err
	}
	errs = make([]error, n)
	for i, c := range st {
		errs[i] = itemErr(c)
	}
	return
}
