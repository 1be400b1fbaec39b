//go:build cgo && hip

// Package hipbatch is the cgo bridge a CIRCL maintainer would add next to kem/hybrid and kem/xwing to route batches of
// the X25519 hybrids -- "X25519MLKEM768", "Kyber768-X25519", "Kyber512-X25519" (kem/hybrid/hybrid.go) and "X-Wing" (kem/xwing) -- to
// libcirclhip.so (MI355X).  Both halves run on the GPU: ML-KEM-768, the X25519 ladders (one lane per item), the seed
// expansions and the X-Wing combiner; the kem.Scheme values stay untouched.
//
// NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Go toolchain.  The same C symbols are exercised through
// ctypes (tests/test_gpu_hybrid.py) and C++ (include/circl/{xwing,hybrid}.hpp).  See INTEGRATION.md.
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

var schemes = map[string]C.int{
	"X-Wing":          C.CIRCL_HIP_HYBRID_XWING,
	"X25519MLKEM768":  C.CIRCL_HIP_HYBRID_X25519MLKEM768,
	"Kyber768-X25519": C.CIRCL_HIP_HYBRID_KYBER768_X25519, // hybrid.Kyber768X25519(): X25519 first, round-3 Kyber768 second
	"Kyber512-X25519": C.CIRCL_HIP_HYBRID_KYBER512_X25519,
}

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
		return kem.ErrPubKey // non-canonical ek (cpapke.go:45-55) or a low-order X25519 point (kem/hybrid/xkem.go:144-146)
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

// DeriveKeyPairBatch is n times scheme.DeriveKeyPair(seed) + MarshalBinary (kem/hybrid/hybrid.go:236-250,
// kem/xwing/xwing.go:98-144).  seeds is row-major [n][SeedSize]; pks, sks are [n][PublicKeySize], [n][PrivateKeySize].
func DeriveKeyPairBatch(s kem.Scheme, seeds []byte, device int) (pks, sks []byte, err error) {
	id, ok := schemes[s.Name()]
	if !ok {
		return nil, nil, kem.ErrTypeMismatch
	}
	if len(seeds)%s.SeedSize() != 0 {
		return nil, nil, kem.ErrSeedSize
	}
	n := len(seeds) / s.SeedSize()
	pks = make([]byte, n*s.PublicKeySize())
	sks = make([]byte, n*s.PrivateKeySize())
	rc := C.circl_hip_hybrid_keygen(id, ptr(seeds), ptr(pks), ptr(sks), C.size_t(n), C.int(device))
	return pks, sks, status(rc, "hybrid keygen")
}

// EncapsulateBatch is n times UnmarshalBinaryPublicKey + EncapsulateDeterministically (hybrid.go:271-300,
// xwing.go:223-265).  errs[i] is nil or kem.ErrPubKey; a failed item's ct and ss are zero.
func EncapsulateBatch(s kem.Scheme, pks, seeds []byte, device int) (cts, sss []byte, errs []error, err error) {
	id, ok := schemes[s.Name()]
	if !ok {
		return nil, nil, nil, kem.ErrTypeMismatch
	}
	if len(pks)%s.PublicKeySize() != 0 {
		return nil, nil, nil, kem.ErrPubKeySize
	}
	n := len(pks) / s.PublicKeySize()
	if len(seeds) != n*s.EncapsulationSeedSize() {
		return nil, nil, nil, kem.ErrSeedSize
	}
	cts = make([]byte, n*s.CiphertextSize())
	sss = make([]byte, n*s.SharedKeySize())
	st := make([]byte, n)
	rc := C.circl_hip_hybrid_encaps(id, ptr(pks), ptr(seeds), ptr(cts), ptr(sss), ptr(st), C.size_t(n), C.int(device))
	if err = status(rc, "hybrid encaps"); err != nil {
		return nil, nil, nil, err
	}
	errs = make([]error, n)
	for i, b := range st {
		errs[i] = itemErr(b)
	}
	return cts, sss, errs, nil
}

// DecapsulateBatch is n times UnmarshalBinaryPrivateKey + Decapsulate (hybrid.go:302-323, xwing.go:270-299).
// errs[i] is nil, kem.ErrPrivKey (hash check of the ML-KEM private key) or kem.ErrPubKey (low-order X25519 ciphertext,
// X25519MLKEM768 only); implicit rejection of the ML-KEM half is not an error.
func DecapsulateBatch(s kem.Scheme, sks, cts []byte, device int) (sss []byte, errs []error, err error) {
	id, ok := schemes[s.Name()]
	if !ok {
		return nil, nil, kem.ErrTypeMismatch
	}
	if len(sks)%s.PrivateKeySize() != 0 {
		return nil, nil, kem.ErrPrivKeySize
	}
	n := len(sks) / s.PrivateKeySize()
	if len(cts) != n*s.CiphertextSize() {
		return nil, nil, kem.ErrCiphertextSize
	}
	sss = make([]byte, n*s.SharedKeySize())
	st := make([]byte, n)
	rc := C.circl_hip_hybrid_decaps(id, ptr(sks), ptr(cts), ptr(sss), ptr(st), C.size_t(n), C.int(device))
	if err = status(rc, "hybrid decaps"); err != nil {
		return nil, nil, err
	}
	errs = make([]error, n)
	for i, b := range st {
		errs[i] = itemErr(b)
	}
	return sss, errs, nil
}
