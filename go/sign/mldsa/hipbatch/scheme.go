//go:build cgo && hip

package hipbatch

// A sign.Scheme that drops in behind sign/mldsa (and sign/dilithium): every method of sign.Scheme (sign/sign.go:48-94)
// is CIRCL's own -- single operations stay on the Go path -- and the batch methods route whole batches to
// libcirclhip.so.  Both paths produce identical bytes (this repository's parity tests).
//
//	s := hipbatch.ByName("ML-DSA-65")          // or hipbatch.Wrap(mldsa65.Scheme())
//	ok := s.Verify(pk, msg, sig, opts)         // CIRCL, unchanged
//	oks, err := s.VerifyBatch(pks, msgs, sigs, ctxs, hipbatch.AllDevices)
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain in the build image); tests/cgo_shape_test.c drives the same C
// entry points with the argument shapes these methods produce.

import (
	"fmt"
	"unsafe"

	"github.com/cloudflare/circl/sign"
	"github.com/cloudflare/circl/sign/schemes"
)

/*
#include <circl_hip.h>
*/
import "C"

// AllDevices splits a batch into contiguous shards, one per visible GPU (no collective).
const AllDevices = -1

// Scheme is sign.Scheme plus batch operations; the embedded value is CIRCL's stateless singleton
// (sign/mldsa/mldsa65/dilithium.go:256-261).
type Scheme struct {
	sign.Scheme
}

// Wrap returns s with batch operations, or nil if libcirclhip.so has no kernels for it.
func Wrap(s sign.Scheme) *Scheme {
	if s == nil {
		return nil
	}
	if _, ok := params[s.Name()]; !ok {
		return nil
	}
	return &Scheme{s}
}

// ByName looks the scheme up in CIRCL's registry (sign/schemes/schemes.go:31-54): "ML-DSA-44", "ML-DSA-65",
// "ML-DSA-87", "Dilithium2", "Dilithium3", "Dilithium5".
func ByName(name string) *Scheme { return Wrap(schemes.ByName(name)) }

// VerifyBatch is n times scheme.Verify(pks[i], msgs[i], sigs[i], &sign.SignatureOpts{Context: ctxs[i]})
// (mldsa65/dilithium.go:305-327).  A key of another scheme panics with sign.ErrTypeMismatch as the single call does
// (:311-314).
func (s *Scheme) VerifyBatch(pks []sign.PublicKey, msgs, sigs [][]byte, ctxs []string, device int) ([]bool, error) {
	raw := make([][]byte, len(pks))
	for i, pk := range pks {
		if pk.Scheme().Name() != s.Name() {
			panic(sign.ErrTypeMismatch)
		}
		b, err := pk.MarshalBinary()
		if err != nil {
			return nil, err
		}
		raw[i] = b
	}
	return VerifyBatch(s.Scheme, raw, msgs, sigs, ctxs, device)
}

// SignBatch is n times scheme.Sign(sks[i], msgs[i], &sign.SignatureOpts{Context: ctxs[i]}) (deterministic, dilithium.go:283-303).
func (s *Scheme) SignBatch(sks []sign.PrivateKey, msgs [][]byte, ctxs []string, device int) ([][]byte, error) {
	raw := make([][]byte, len(sks))
	for i, sk := range sks {
		if sk.Scheme().Name() != s.Name() {
			panic(sign.ErrTypeMismatch)
		}
		b, err := sk.MarshalBinary()
		if err != nil {
			return nil, err
		}
		raw[i] = b
	}
	// the marshalled private keys are this wrapper's own copies: they do not outlive the call (the native library wipes its
	// staging and device copies; without this the Go heap would keep n packed private keys until the collector reuses them)
	defer func() {
		for _, b := range raw {
			clear(b)
		}
	}()
	return SignBatch(s.Scheme, raw, msgs, ctxs, nil, device)
}

// PublicKeyTable is a set of parsed public keys a batch refers to by index (a verifier that sees a few CA keys across a
// large batch): tr and the matrix A -- what PublicKey.Unpack caches per key object, internal/dilithium.go:114-126 -- are
// derived once per table entry on the device.
type PublicKeyTable struct {
	scheme *Scheme
	rows   []byte
	n      int
}

func (s *Scheme) NewPublicKeyTable(pks []sign.PublicKey) (*PublicKeyTable, error) {
	t := &PublicKeyTable{scheme: s, n: len(pks)}
	for _, pk := range pks {
		if pk.Scheme().Name() != s.Name() {
			panic(sign.ErrTypeMismatch)
		}
		b, err := pk.MarshalBinary()
		if err != nil {
			return nil, err
		}
		t.rows = append(t.rows, b...)
	}
	return t, nil
}

// VerifyKeyedBatch: signature i is checked under table entry idx[i].  Signatures of the wrong length and contexts longer
// than 255 bytes verify as false without reaching the device, like the single call (dilithium.go:116-118).
func (t *PublicKeyTable) VerifyKeyedBatch(idx []uint32, msgs, sigs [][]byte, ctxs []string, device int) ([]bool, error) {
	s := t.scheme
	p := params[s.Name()]
	n := len(idx)
	if len(msgs) != n || len(sigs) != n {
		return nil, sign.ErrTypeMismatch
	}
	res := make([]bool, n)
	sigRows := make([]byte, 0, n*s.SignatureSize())
	var msgBlob, ctxBlob []byte
	msgOff := make([]uint64, 1, n+1)
	ctxOff := make([]uint64, 1, n+1)
	keep := make([]int, 0, n)
	kidx := make([]uint32, 0, n)
	for i := 0; i < n; i++ {
		if int(idx[i]) >= t.n {
			return nil, fmt.Errorf("circl-hip: key index %d out of range", idx[i])
		}
		ctx := "" // ctxs may be nil or shorter than idx: the missing contexts are empty
		if i < len(ctxs) {
			ctx = ctxs[i]
		}
		if len(sigs[i]) != s.SignatureSize() || len(ctx) > 255 {
			continue
		}
		keep = append(keep, i)
		kidx = append(kidx, idx[i])
		sigRows = append(sigRows, sigs[i]...)
		msgBlob = append(msgBlob, msgs[i]...)
		ctxBlob = append(ctxBlob, ctx...)
		msgOff = append(msgOff, uint64(len(msgBlob)))
		ctxOff = append(ctxOff, uint64(len(ctxBlob)))
	}
	if len(keep) == 0 {
		return res, nil
	}
	msgBlob = append(msgBlob, 0) // keep the blobs non-empty so that &blob[0] is valid
	ctxBlob = append(ctxBlob, 0)
	okb := make([]byte, len(keep))
	rc := C.circl_hip_mldsa_verify_keyed(p, (*C.uint8_t)(unsafe.Pointer(&t.rows[0])), C.size_t(t.n), (*C.uint32_t)(unsafe.Pointer(&kidx[0])),
		(*C.uint8_t)(unsafe.Pointer(&sigRows[0])), (*C.uint8_t)(unsafe.Pointer(&msgBlob[0])), (*C.uint64_t)(unsafe.Pointer(&msgOff[0])),
		(*C.uint8_t)(unsafe.Pointer(&ctxBlob[0])), (*C.uint64_t)(unsafe.Pointer(&ctxOff[0])), (*C.uint8_t)(unsafe.Pointer(&okb[0])),
		C.size_t(len(keep)), C.int(device))
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip verify keyed: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	for k, i := range keep {
		res[i] = okb[k] == 1
	}
	return res, nil
}
