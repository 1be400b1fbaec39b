//go:build cgo && hip

package hipbatch

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../include
#cgo LDFLAGS: -lcirclhip
#include <circl_hip.h>
*/
import "C"

import (
	"fmt"
	"runtime"
	"unsafe"

	"github.com/cloudflare/circl/sign"
)

// ResidentKeys is the GPU-side counterpart of parsed ML-DSA key objects: sign.Scheme.UnmarshalBinaryPublicKey keeps A and tr,
// UnmarshalBinaryPrivateKey keeps A and the NTT-domain s1, s2, t0 (sign/mldsa/mldsa65/internal/dilithium.go:114-126, :149-179).
// Built once on one device; Verify / Sign calls then move only messages and signatures.  Released by Close or the finalizer
// (a private key's table is wiped first).
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain); tests/test_gpu_keytable.py drives the same symbols.
type ResidentKeys struct {
	s       sign.Scheme
	t       *C.circl_hip_keytable
	private bool
}

// NewResidentPublicKeys parses n packed public keys ([n][PublicKeySize]) on `device`.
func NewResidentPublicKeys(s sign.Scheme, pks []byte, device int) (*ResidentKeys, error) {
	p, ok := params[s.Name()]
	if !ok {
		return nil, sign.ErrTypeMismatch
	}
	if len(pks) == 0 || len(pks)%s.PublicKeySize() != 0 {
		return nil, sign.ErrPubKeySize
	}
	r := &ResidentKeys{s: s}
	rc := C.circl_hip_mldsa_keytable_new(p, (*C.uint8_t)(unsafe.Pointer(&pks[0])), C.size_t(len(pks)/s.PublicKeySize()), C.int(device), &r.t)
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip keytable: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	runtime.SetFinalizer(r, func(k *ResidentKeys) { k.Close() })
	return r, nil
}

// NewResidentPrivateKey prepares ONE packed private key for signing on `device`.
func NewResidentPrivateKey(s sign.Scheme, sk []byte, device int) (*ResidentKeys, error) {
	p, ok := params[s.Name()]
	if !ok {
		return nil, sign.ErrTypeMismatch
	}
	if len(sk) != s.PrivateKeySize() {
		return nil, sign.ErrPrivKeySize
	}
	r := &ResidentKeys{s: s, private: true}
	rc := C.circl_hip_mldsa_privkey_new(p, (*C.uint8_t)(unsafe.Pointer(&sk[0])), C.int(device), &r.t)
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip privkey: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	runtime.SetFinalizer(r, func(k *ResidentKeys) { k.Close() })
	return r, nil
}

// Close releases the device memory (idempotent).
func (r *ResidentKeys) Close() {
	if r.t != nil {
		C.circl_hip_keytable_free(r.t)
		r.t = nil
	}
}

func blobs(msgs [][]byte, ctxs []string) (mb []byte, mo []uint64, cb []byte, co []uint64) {
	mo = make([]uint64, 1, len(msgs)+1)
	co = make([]uint64, 1, len(msgs)+1)
	for i := range msgs {
		mb = append(mb, msgs[i]...)
		cb = append(cb, ctxs[i]...)
		mo = append(mo, uint64(len(mb)))
		co = append(co, uint64(len(cb)))
	}
	return append(mb, 0), mo, append(cb, 0), co // (non-empty, so that &blob[0] is valid)
}

// Verify checks sigs[i] over msgs[i] with context ctxs[i] under table entry idx[i] (idx == nil: entry 0).  Signatures of the wrong
// length must be filtered by the caller (they verify as false, sign/mldsa/mldsa65/dilithium.go:305-343).
func (r *ResidentKeys) Verify(idx []uint32, msgs [][]byte, sigRows []byte, ctxs []string) ([]bool, error) {
	if r.t == nil || r.private {
		return nil, sign.ErrTypeMismatch
	}
	n := len(msgs)
	if len(sigRows) != n*r.s.SignatureSize() || (idx != nil && len(idx) != n) {
		return nil, sign.ErrTypeMismatch
	}
	mb, mo, cb, co := blobs(msgs, ctxs)
	okb := make([]byte, n)
	var ip *C.uint32_t
	if idx != nil {
		ip = (*C.uint32_t)(unsafe.Pointer(&idx[0]))
	}
	rc := C.circl_hip_mldsa_verify_table(r.t, ip, (*C.uint8_t)(unsafe.Pointer(&sigRows[0])), (*C.uint8_t)(unsafe.Pointer(&mb[0])),
		(*C.uint64_t)(unsafe.Pointer(&mo[0])), (*C.uint8_t)(unsafe.Pointer(&cb[0])), (*C.uint64_t)(unsafe.Pointer(&co[0])),
		(*C.uint8_t)(unsafe.Pointer(&okb[0])), C.size_t(n))
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip verify: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	res := make([]bool, n)
	for i := range okb {
		res[i] = okb[i] == 1
	}
	runtime.KeepAlive(r)
	return res, nil
}

// Sign is len(msgs) times scheme.Sign(sk, msg, &SignatureOpts{Context: ctx}) with the prepared key; rnd is nil (deterministic) or
// n*32 bytes from crypto/rand (hedged, dilithium.go:56-88).
func (r *ResidentKeys) Sign(msgs [][]byte, ctxs []string, rnd []byte) ([][]byte, error) {
	if r.t == nil || !r.private {
		return nil, sign.ErrTypeMismatch
	}
	n := len(msgs)
	for i := range ctxs {
		if len(ctxs[i]) > 255 {
			return nil, sign.ErrContextTooLong
		}
	}
	mb, mo, cb, co := blobs(msgs, ctxs)
	sigRows := make([]byte, n*r.s.SignatureSize())
	var rp *C.uint8_t
	if rnd != nil {
		rp = (*C.uint8_t)(unsafe.Pointer(&rnd[0]))
	}
	rc := C.circl_hip_mldsa_sign_table(r.t, (*C.uint8_t)(unsafe.Pointer(&mb[0])), (*C.uint64_t)(unsafe.Pointer(&mo[0])),
		(*C.uint8_t)(unsafe.Pointer(&cb[0])), (*C.uint64_t)(unsafe.Pointer(&co[0])), rp, (*C.uint8_t)(unsafe.Pointer(&sigRows[0])), C.size_t(n))
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip sign: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	out := make([][]byte, n)
	for i := range out {
		out[i] = sigRows[i*r.s.SignatureSize() : (i+1)*r.s.SignatureSize()]
	}
	runtime.KeepAlive(r)
	return out, nil
}
