//go:build cgo && hip

// Package hipbatch is the cgo bridge for batched ML-DSA verification on libcirclhip.so.
// NOT COMPILED HERE (no Go toolchain in the build image); see INTEGRATION.md.
package hipbatch

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../include
#cgo LDFLAGS: -lcirclhip
#include <circl_hip.h>
*/
import "C"

import (
	"fmt"
	"unsafe"

	"github.com/cloudflare/circl/sign"
)

// ML-DSA, and the round-3 Dilithium modes of sign/dilithium/mode{2,3,5} (param 2 / 3 / 5 of the same C entry points:
// contexts must be empty, signing is deterministic; sign/dilithium/mode3/dilithium.go:213-255).
var params = map[string]C.int{"ML-DSA-44": 44, "ML-DSA-65": 65, "ML-DSA-87": 87, "Dilithium2": 2, "Dilithium3": 3, "Dilithium5": 5}

// VerifyBatch is n times scheme.UnmarshalBinaryPublicKey + scheme.Verify(pk, msg, sig, &opts)
// (sign/mldsa/mldsa65/dilithium.go:305-343).  Signatures of the wrong length and contexts longer
// than 255 bytes verify as false, exactly like the single-shot call.
func VerifyBatch(s sign.Scheme, pks [][]byte, msgs [][]byte, sigs [][]byte, ctxs []string, device int) ([]bool, error) {
	p, ok := params[s.Name()]
	if !ok {
		panic(sign.ErrTypeMismatch)
	}
	n := len(pks)
	if len(msgs) != n || len(sigs) != n {
		return nil, sign.ErrTypeMismatch
	}
	res := make([]bool, n)
	pkRows := make([]byte, 0, n*s.PublicKeySize())
	sigRows := make([]byte, 0, n*s.SignatureSize())
	var msgBlob, ctxBlob []byte
	msgOff := make([]uint64, 1, n+1)
	ctxOff := make([]uint64, 1, n+1)
	idx := make([]int, 0, n)
	for i := 0; i < n; i++ {
		if len(pks[i]) != s.PublicKeySize() {
			return nil, sign.ErrPubKeySize
		}
		ctx := "" // ctxs may be nil or shorter than pks: the missing contexts are empty
		if i < len(ctxs) {
			ctx = ctxs[i]
		}
		if len(sigs[i]) != s.SignatureSize() || len(ctx) > 255 {
			continue // false, without touching the device
		}
		idx = append(idx, i)
		pkRows = append(pkRows, pks[i]...)
		sigRows = append(sigRows, sigs[i]...)
		msgBlob = append(msgBlob, msgs[i]...)
		ctxBlob = append(ctxBlob, ctx...)
		msgOff = append(msgOff, uint64(len(msgBlob)))
		ctxOff = append(ctxOff, uint64(len(ctxBlob)))
	}
	if len(idx) == 0 {
		return res, nil
	}
	msgBlob = append(msgBlob, 0) // keep the blobs non-empty so that &blob[0] is valid
	ctxBlob = append(ctxBlob, 0)
	okb := make([]byte, len(idx))
	rc := C.circl_hip_mldsa_verify(p,
		(*C.uint8_t)(unsafe.Pointer(&pkRows[0])), (*C.uint8_t)(unsafe.Pointer(&sigRows[0])),
		(*C.uint8_t)(unsafe.Pointer(&msgBlob[0])), (*C.uint64_t)(unsafe.Pointer(&msgOff[0])),
		(*C.uint8_t)(unsafe.Pointer(&ctxBlob[0])), (*C.uint64_t)(unsafe.Pointer(&ctxOff[0])),
		(*C.uint8_t)(unsafe.Pointer(&okb[0])), C.size_t(len(idx)), C.int(device))
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip verify: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	for k, i := range idx {
		res[i] = okb[k] == 1
	}
	return res, nil
}

// SignBatch is n times scheme.Sign(sk, msg, &SignatureOpts{Context: ctx}) (sign/mldsa/mldsa65/dilithium.go:283-303).
// rnd is nil for deterministic signatures (what scheme.Sign produces) or n*32 bytes from crypto/rand
// for hedged ones (SignTo(..., randomized=true, ...), dilithium.go:56-88).
func SignBatch(s sign.Scheme, sks [][]byte, msgs [][]byte, ctxs []string, rnd []byte, device int) ([][]byte, error) {
	p, ok := params[s.Name()]
	if !ok {
		panic(sign.ErrTypeMismatch)
	}
	n := len(sks)
	if len(msgs) != n || (rnd != nil && len(rnd) != 32*n) {
		return nil, sign.ErrTypeMismatch
	}
	if n == 0 {
		return [][]byte{}, nil
	}
	skRows := make([]byte, 0, n*s.PrivateKeySize())
	defer func() { clear(skRows[:cap(skRows)]) }() // the contiguous copy of the private keys does not outlive the call
	var msgBlob, ctxBlob []byte
	msgOff := make([]uint64, 1, n+1)
	ctxOff := make([]uint64, 1, n+1)
	for i := 0; i < n; i++ {
		if len(sks[i]) != s.PrivateKeySize() {
			return nil, sign.ErrPrivKeySize
		}
		ctx := ""
		if i < len(ctxs) {
			ctx = ctxs[i]
		}
		if len(ctx) > 255 {
			return nil, sign.ErrContextTooLong
		}
		skRows = append(skRows, sks[i]...)
		msgBlob = append(msgBlob, msgs[i]...)
		ctxBlob = append(ctxBlob, ctx...)
		msgOff = append(msgOff, uint64(len(msgBlob)))
		ctxOff = append(ctxOff, uint64(len(ctxBlob)))
	}
	msgBlob = append(msgBlob, 0)
	ctxBlob = append(ctxBlob, 0)
	sigRows := make([]byte, n*s.SignatureSize())
	var rndPtr *C.uint8_t
	if len(rnd) != 0 {
		rndPtr = (*C.uint8_t)(unsafe.Pointer(&rnd[0]))
	}
	rc := C.circl_hip_mldsa_sign(p, (*C.uint8_t)(unsafe.Pointer(&skRows[0])),
		(*C.uint8_t)(unsafe.Pointer(&msgBlob[0])), (*C.uint64_t)(unsafe.Pointer(&msgOff[0])),
		(*C.uint8_t)(unsafe.Pointer(&ctxBlob[0])), (*C.uint64_t)(unsafe.Pointer(&ctxOff[0])),
		rndPtr, (*C.uint8_t)(unsafe.Pointer(&sigRows[0])), C.size_t(n), C.int(device))
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip sign: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	out := make([][]byte, n)
	for i := range out {
		out[i] = sigRows[i*s.SignatureSize() : (i+1)*s.SignatureSize()]
	}
	return out, nil
}

// DeriveKeyBatch is n times scheme.DeriveKey(seed) (dilithium.go:272-281); seeds are [n][SeedSize].
func DeriveKeyBatch(s sign.Scheme, seeds []byte, device int) (pks, sks []byte, err error) {
	p, ok := params[s.Name()]
	if !ok {
		panic(sign.ErrTypeMismatch)
	}
	if len(seeds)%s.SeedSize() != 0 {
		panic(sign.ErrSeedSize)
	}
	n := len(seeds) / s.SeedSize()
	if n == 0 {
		return []byte{}, []byte{}, nil
	}
	pks = make([]byte, n*s.PublicKeySize())
	sks = make([]byte, n*s.PrivateKeySize())
	rc := C.circl_hip_mldsa_keygen(p, (*C.uint8_t)(unsafe.Pointer(&seeds[0])), (*C.uint8_t)(unsafe.Pointer(&pks[0])),
		(*C.uint8_t)(unsafe.Pointer(&sks[0])), C.size_t(n), C.int(device))
	if rc != 0 {
		err = fmt.Errorf("circl-hip keygen: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	return
}
