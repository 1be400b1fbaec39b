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
	"sync"
	"unsafe"

	"github.com/cloudflare/circl/sign"
)

// (AllDevices as the `device` of a constructor below replicates the table on every GPU; calls then split their batch into
// contiguous shards, one per device: CIRCL_HIP_ALL_DEVICES, include/circl_hip.h.)

// ResidentKeys is the GPU-side counterpart of parsed ML-DSA key objects: sign.Scheme.UnmarshalBinaryPublicKey keeps A and tr,
// UnmarshalBinaryPrivateKey keeps A and the NTT-domain s1, s2, t0 (sign/mldsa/mldsa65/internal/dilithium.go:114-126, :149-179).
// Built once on one device (or replicated on all); Verify / Sign calls then move only messages and signatures.  A table is
// immutable: any number of goroutines may call Verify / Sign at once (they hold the read lock); Close takes the write lock, so it
// waits for calls in flight and later calls fail with sign.ErrTypeMismatch instead of touching freed memory.  Released by Close or
// the finalizer (a private key's table is wiped first).
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain); tests/test_gpu_keytable.py drives the same symbols.
type ResidentKeys struct {
	mu      sync.RWMutex
	s       sign.Scheme
	t       *C.circl_hip_keytable
	private bool
	n       int
}

func newResident(s sign.Scheme, rows []byte, rowSize int, private bool, device int) (*ResidentKeys, error) {
	p, ok := params[s.Name()]
	if !ok {
		return nil, sign.ErrTypeMismatch
	}
	if len(rows) == 0 || len(rows)%rowSize != 0 {
		if private {
			return nil, sign.ErrPrivKeySize
		}
		return nil, sign.ErrPubKeySize
	}
	r := &ResidentKeys{s: s, private: private, n: len(rows) / rowSize}
	var rc C.int
	if private {
		rc = C.circl_hip_mldsa_privkeys_new(p, (*C.uint8_t)(unsafe.Pointer(&rows[0])), C.size_t(r.n), C.int(device), &r.t)
	} else {
		rc = C.circl_hip_mldsa_keytable_new(p, (*C.uint8_t)(unsafe.Pointer(&rows[0])), C.size_t(r.n), C.int(device), &r.t)
	}
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip keytable: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	runtime.SetFinalizer(r, func(k *ResidentKeys) { k.Close() })
	return r, nil
}

// NewResidentPublicKeys parses n packed public keys ([n][PublicKeySize]) on `device` (AllDevices: on every device).
func NewResidentPublicKeys(s sign.Scheme, pks []byte, device int) (*ResidentKeys, error) {
	return newResident(s, pks, s.PublicKeySize(), false, device)
}

// NewResidentPrivateKeys prepares n packed private keys ([n][PrivateKeySize]) for signing: a signer that holds several identities.
func NewResidentPrivateKeys(s sign.Scheme, sks []byte, device int) (*ResidentKeys, error) {
	return newResident(s, sks, s.PrivateKeySize(), true, device)
}

// NewResidentPrivateKey prepares ONE packed private key.
func NewResidentPrivateKey(s sign.Scheme, sk []byte, device int) (*ResidentKeys, error) {
	if len(sk) != s.PrivateKeySize() {
		return nil, sign.ErrPrivKeySize
	}
	return newResident(s, sk, s.PrivateKeySize(), true, device)
}

// Len is the number of keys in the table.
func (r *ResidentKeys) Len() int { return r.n }

// Close releases the device memory (idempotent); it waits for calls that are using the table.
func (r *ResidentKeys) Close() {
	r.mu.Lock()
	defer r.mu.Unlock()
	if r.t != nil {
		C.circl_hip_keytable_free(r.t)
		r.t = nil
	}
}

// blobs lays messages and contexts out as the C ABI wants them (blob + n+1 offsets).  ctxs may be nil (all contexts empty) or
// shorter than msgs (the missing ones are empty).
func blobs(msgs [][]byte, ctxs []string) (mb []byte, mo []uint64, cb []byte, co []uint64) {
	mo = make([]uint64, 1, len(msgs)+1)
	co = make([]uint64, 1, len(msgs)+1)
	for i := range msgs {
		mb = append(mb, msgs[i]...)
		if i < len(ctxs) {
			cb = append(cb, ctxs[i]...)
		}
		mo = append(mo, uint64(len(mb)))
		co = append(co, uint64(len(cb)))
	}
	return append(mb, 0), mo, append(cb, 0), co // (non-empty, so that &blob[0] is valid)
}

func idxPtr(idx []uint32) *C.uint32_t {
	if len(idx) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&idx[0]))
}

// Verify checks sigs[i] over msgs[i] with context ctxs[i] under table entry idx[i] (idx == nil: entry 0).  Signatures of the wrong
// length must be filtered by the caller (they verify as false, sign/mldsa/mldsa65/dilithium.go:305-343).
func (r *ResidentKeys) Verify(idx []uint32, msgs [][]byte, sigRows []byte, ctxs []string) ([]bool, error) {
	r.mu.RLock()
	defer r.mu.RUnlock()
	if r.t == nil || r.private {
		return nil, sign.ErrTypeMismatch
	}
	n := len(msgs)
	if len(sigRows) != n*r.s.SignatureSize() || (idx != nil && len(idx) != n) {
		return nil, sign.ErrTypeMismatch
	}
	if n == 0 {
		return []bool{}, nil
	}
	mb, mo, cb, co := blobs(msgs, ctxs)
	okb := make([]byte, n)
	rc := C.circl_hip_mldsa_verify_table(r.t, idxPtr(idx), (*C.uint8_t)(unsafe.Pointer(&sigRows[0])), (*C.uint8_t)(unsafe.Pointer(&mb[0])),
		(*C.uint64_t)(unsafe.Pointer(&mo[0])), (*C.uint8_t)(unsafe.Pointer(&cb[0])), (*C.uint64_t)(unsafe.Pointer(&co[0])),
		(*C.uint8_t)(unsafe.Pointer(&okb[0])), C.size_t(n))
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip verify: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	res := make([]bool, n)
	for i := range okb {
		res[i] = okb[i] == 1
	}
	return res, nil
}

// Sign is len(msgs) times scheme.Sign(sk_idx[i], msg, &SignatureOpts{Context: ctx}) with the prepared keys (idx == nil: entry 0);
// rnd is nil (deterministic) or n*32 bytes from crypto/rand (hedged, dilithium.go:56-88).
func (r *ResidentKeys) Sign(idx []uint32, msgs [][]byte, ctxs []string, rnd []byte) ([][]byte, error) {
	r.mu.RLock()
	defer r.mu.RUnlock()
	if r.t == nil || !r.private {
		return nil, sign.ErrTypeMismatch
	}
	n := len(msgs)
	if (idx != nil && len(idx) != n) || (rnd != nil && len(rnd) != 32*n) {
		return nil, sign.ErrTypeMismatch
	}
	for i := range ctxs {
		if len(ctxs[i]) > 255 {
			return nil, sign.ErrContextTooLong
		}
	}
	if n == 0 {
		return [][]byte{}, nil
	}
	mb, mo, cb, co := blobs(msgs, ctxs)
	sigRows := make([]byte, n*r.s.SignatureSize())
	var rp *C.uint8_t
	if len(rnd) != 0 {
		rp = (*C.uint8_t)(unsafe.Pointer(&rnd[0]))
	}
	rc := C.circl_hip_mldsa_sign_table_keyed(r.t, idxPtr(idx), (*C.uint8_t)(unsafe.Pointer(&mb[0])), (*C.uint64_t)(unsafe.Pointer(&mo[0])),
		(*C.uint8_t)(unsafe.Pointer(&cb[0])), (*C.uint64_t)(unsafe.Pointer(&co[0])), rp, (*C.uint8_t)(unsafe.Pointer(&sigRows[0])), C.size_t(n))
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip sign: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	out := make([][]byte, n)
	for i := range out {
		out[i] = sigRows[i*r.s.SignatureSize() : (i+1)*r.s.SignatureSize()]
	}
	return out, nil
}

// PublicKeys is PrivateKey.Public() over a batch of packed private keys ([n][PrivateKeySize]) -> packed public keys
// (sign/mldsa/mldsa65/internal/dilithium.go:473-484: t1 recomputed from s1, s2 on the device).
func PublicKeys(s sign.Scheme, sks []byte, device int) ([]byte, error) {
	p, ok := params[s.Name()]
	if !ok {
		return nil, sign.ErrTypeMismatch
	}
	if len(sks)%s.PrivateKeySize() != 0 {
		return nil, sign.ErrPrivKeySize
	}
	n := len(sks) / s.PrivateKeySize()
	if n == 0 {
		return []byte{}, nil
	}
	pks := make([]byte, n*s.PublicKeySize())
	rc := C.circl_hip_mldsa_public_from_private(p, (*C.uint8_t)(unsafe.Pointer(&sks[0])), (*C.uint8_t)(unsafe.Pointer(&pks[0])), C.size_t(n), C.int(device))
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip public: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	return pks, nil
}
