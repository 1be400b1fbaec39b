//go:build cgo && hip

package hipbatch

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../include
#cgo LDFLAGS: -lcirclhip
#include <circl_hip.h>
*/
import "C"

import (
	"runtime"
	"sync"
	"unsafe"

	"github.com/cloudflare/circl/kem"
)

// ResidentKeys is the GPU-side counterpart of the hybrid schemes' parsed key objects: kem/xwing's PrivateKey keeps the expanded
// ML-KEM-768 key, the X25519 scalar and its public point next to the 32-byte seed (kem/xwing/xwing.go:20-25), its PublicKey the
// parsed ML-KEM key (:28-31); kem/hybrid's keys hold the component schemes' parsed keys (kem/hybrid/hybrid.go:101-114).  The table
// holds an ML-KEM key table of the lattice halves (A^T, H(ek), the private keys' hash verdicts) and the X25519 rows, built ONCE on
// `device` (AllDevices: replicated, calls shard their batch); an X-Wing private key -- the seed -- is expanded on the device when
// the table is built, not per decapsulation.  "X-Wing" and "X25519MLKEM768" only (round-3 Kyber has no key tables).
//
// Calls hold the read lock, Close the write lock.  NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain);
// tests/test_gpu_keytable.py::test_hybrid_key_tables_match_the_oracle_call_after_call drives the same symbols.
type ResidentKeys struct {
	mu      sync.RWMutex
	s       kem.Scheme
	t       *C.circl_hip_keytable
	private bool
	n       int
}

func newResident(s kem.Scheme, rows []byte, rowSize int, private bool, device int) (*ResidentKeys, []error, error) {
	id, ok := schemes[s.Name()]
	if !ok {
		return nil, nil, kem.ErrTypeMismatch
	}
	if len(rows) == 0 || len(rows)%rowSize != 0 {
		if private {
			return nil, nil, kem.ErrPrivKeySize
		}
		return nil, nil, kem.ErrPubKeySize
	}
	n := len(rows) / rowSize
	st := make([]byte, n)
	r := &ResidentKeys{s: s, private: private, n: n}
	priv := C.int(0)
	if private {
		priv = 1
	}
	if err := status(C.circl_hip_hybrid_keytable_new(id, priv, ptr(rows), C.size_t(n), C.int(device), ptr(st), &r.t), "hybrid keytable"); err != nil {
		return nil, nil, err
	}
	runtime.SetFinalizer(r, func(k *ResidentKeys) { k.Close() })
	errs := make([]error, n)
	for i := range st {
		errs[i] = itemErr(st[i])
	}
	return r, errs, nil
}

// NewResidentPublicKeys parses n packed public keys ([n][PublicKeySize]).
func NewResidentPublicKeys(s kem.Scheme, pks []byte, device int) (*ResidentKeys, error) {
	r, _, err := newResident(s, pks, s.PublicKeySize(), false, device)
	return r, err
}

// NewResidentPrivateKeys parses n packed private keys; errs[i] is kem.ErrPrivKey for a key whose ML-KEM half fails its hash check.
func NewResidentPrivateKeys(s kem.Scheme, sks []byte, device int) (*ResidentKeys, []error, error) {
	return newResident(s, sks, s.PrivateKeySize(), true, device)
}

// Close releases (and, for private keys, wipes) the device memory; idempotent; waits for calls in flight.
func (r *ResidentKeys) Close() {
	r.mu.Lock()
	defer r.mu.Unlock()
	if r.t != nil {
		C.circl_hip_keytable_free(r.t)
		r.t = nil
	}
}

func idxPtr(idx []uint32) *C.uint32_t {
	if len(idx) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&idx[0]))
}

// Encapsulate is len(seeds)/EncapsulationSeedSize times EncapsulateDeterministically to table entry idx[i] (idx == nil: entry 0).
func (r *ResidentKeys) Encapsulate(idx []uint32, seeds []byte) (cts, sss []byte, errs []error, err error) {
	r.mu.RLock()
	defer r.mu.RUnlock()
	if r.t == nil || r.private {
		return nil, nil, nil, kem.ErrTypeMismatch
	}
	if len(seeds)%r.s.EncapsulationSeedSize() != 0 {
		return nil, nil, nil, kem.ErrSeedSize
	}
	n := len(seeds) / r.s.EncapsulationSeedSize()
	if idx != nil && len(idx) != n {
		return nil, nil, nil, kem.ErrTypeMismatch
	}
	if n == 0 {
		return []byte{}, []byte{}, []error{}, nil
	}
	cts = make([]byte, n*r.s.CiphertextSize())
	sss = make([]byte, n*r.s.SharedKeySize())
	st := make([]byte, n)
	if err = status(C.circl_hip_hybrid_encaps_table(r.t, idxPtr(idx), ptr(seeds), ptr(cts), ptr(sss), ptr(st), C.size_t(n)), "hybrid encaps table"); err != nil {
		return nil, nil, nil, err
	}
	errs = make([]error, n)
	for i := range st {
		errs[i] = itemErr(st[i])
	}
	return cts, sss, errs, nil
}

// Decapsulate is len(cts)/CiphertextSize times Decapsulate with table entry idx[i] (idx == nil: entry 0).
func (r *ResidentKeys) Decapsulate(idx []uint32, cts []byte) (sss []byte, errs []error, err error) {
	r.mu.RLock()
	defer r.mu.RUnlock()
	if r.t == nil || !r.private {
		return nil, nil, kem.ErrTypeMismatch
	}
	if len(cts)%r.s.CiphertextSize() != 0 {
		return nil, nil, kem.ErrCiphertextSize
	}
	n := len(cts) / r.s.CiphertextSize()
	if idx != nil && len(idx) != n {
		return nil, nil, kem.ErrTypeMismatch
	}
	if n == 0 {
		return []byte{}, []error{}, nil
	}
	sss = make([]byte, n*r.s.SharedKeySize())
	st := make([]byte, n)
	if err = status(C.circl_hip_hybrid_decaps_table(r.t, idxPtr(idx), ptr(cts), ptr(sss), ptr(st), C.size_t(n)), "hybrid decaps table"); err != nil {
		return nil, nil, err
	}
	errs = make([]error, n)
	for i := range st {
		errs[i] = itemErr(st[i])
	}
	return sss, errs, nil
}
