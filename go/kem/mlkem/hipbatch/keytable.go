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

	"github.com/cloudflare/circl/kem"
)

// ResidentTable is the GPU-side counterpart of parsed key objects: what kem.Scheme.UnmarshalBinaryPublicKey /
// UnmarshalBinaryPrivateKey compute once and keep in the object (A^T, H(ek), the private key's hash check:
// kem/mlkem/mlkem768/kyber.go:39-43, :219-228, :247-263) is computed once on one device and stays there; an
// Encapsulate / Decapsulate call then moves only seeds, ciphertexts and shared secrets.  A table is immutable and may be
// used from several goroutines; it is released by Close or by the finalizer (private tables are wiped first).
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain); tests/test_gpu_keytable.py drives the same symbols.
type ResidentTable struct {
	mu      sync.RWMutex // calls hold the read lock, Close the write lock: a Close cannot free the table under a call
	s       kem.Scheme
	t       *C.circl_hip_keytable
	private bool
	n       int
}

func newResidentTable(s kem.Scheme, rows []byte, rowSize int, private bool, device int) (*ResidentTable, []error, error) {
	p, ok := params[s.Name()]
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
	kt := &ResidentTable{s: s, private: private, n: n}
	priv := C.int(0)
	if private {
		priv = 1
	}
	if err := status(C.circl_hip_mlkem_keytable_new(p, priv, ptr(rows), C.size_t(n), C.int(device), ptr(st), &kt.t), "mlkem_keytable_new"); err != nil {
		return nil, nil, err
	}
	runtime.SetFinalizer(kt, func(k *ResidentTable) { k.Close() })
	errs := make([]error, n)
	for i := range st {
		errs[i] = itemErr(st[i])
	}
	return kt, errs, nil
}

// (AllDevices as the `device` of a constructor below replicates the table on every GPU; calls then split their batch into
// contiguous shards, one per device: CIRCL_HIP_ALL_DEVICES, include/circl_hip.h.)

// PublicKeys is PrivateKey.Public() over a batch of packed private keys (kem/mlkem/mlkem768/kyber.go:323-328): the encapsulation
// key stored inside each decapsulation key.
func PublicKeys(s kem.Scheme, dks []byte) ([]byte, error) {
	p, ok := params[s.Name()]
	if !ok {
		return nil, kem.ErrTypeMismatch
	}
	if len(dks)%s.PrivateKeySize() != 0 {
		return nil, kem.ErrPrivKeySize
	}
	n := len(dks) / s.PrivateKeySize()
	if n == 0 {
		return []byte{}, nil
	}
	eks := make([]byte, n*s.PublicKeySize())
	if err := status(C.circl_hip_mlkem_public_from_private(p, ptr(dks), ptr(eks), C.size_t(n)), "mlkem_public_from_private"); err != nil {
		return nil, err
	}
	return eks, nil
}

// NewResidentPublicKeys parses n packed public keys ([n][PublicKeySize]) on `device` (AllDevices: on every device).
func NewResidentPublicKeys(s kem.Scheme, eks []byte, device int) (*ResidentTable, error) {
	kt, _, err := newResidentTable(s, eks, s.PublicKeySize(), false, device)
	return kt, err
}

// NewResidentPrivateKeys parses n packed private keys; errs[i] is kem.ErrPrivKey for a key whose stored hash does not match.
func NewResidentPrivateKeys(s kem.Scheme, dks []byte, device int) (*ResidentTable, []error, error) {
	return newResidentTable(s, dks, s.PrivateKeySize(), true, device)
}

// Close releases the device memory (idempotent).
func (k *ResidentTable) Close() {
	k.mu.Lock()
	defer k.mu.Unlock()
	if k.t != nil {
		C.circl_hip_keytable_free(k.t)
		k.t = nil
	}
}

// Encapsulate is len(seeds)/EncapsulationSeedSize times EncapsulateDeterministically on table entry idx[i]
// (idx == nil: every item uses entry 0 -- a table of one key object).
func (k *ResidentTable) Encapsulate(idx []uint32, seeds []byte) (cts, sss []byte, errs []error, err error) {
	k.mu.RLock()
	defer k.mu.RUnlock()
	if k.t == nil || k.private {
		return nil, nil, nil, kem.ErrTypeMismatch
	}
	if len(seeds)%k.s.EncapsulationSeedSize() != 0 {
		return nil, nil, nil, kem.ErrSeedSize
	}
	n := len(seeds) / k.s.EncapsulationSeedSize()
	if idx != nil && len(idx) != n {
		return nil, nil, nil, kem.ErrTypeMismatch
	}
	if n == 0 {
		return []byte{}, []byte{}, []error{}, nil
	}
	cts = make([]byte, n*k.s.CiphertextSize())
	sss = make([]byte, n*k.s.SharedKeySize())
	st := make([]byte, n)
	if err = status(C.circl_hip_mlkem_encaps_table(k.t, ptr32(idx), ptr(seeds), ptr(cts), ptr(sss), ptr(st), C.size_t(n)), "mlkem_encaps_table"); err != nil {
		return nil, nil, nil, err
	}
	errs = make([]error, n)
	for i := range st {
		errs[i] = itemErr(st[i])
	}
	runtime.KeepAlive(k)
	return cts, sss, errs, nil
}

// Decapsulate is len(cts)/CiphertextSize times Decapsulate with table entry idx[i] (idx == nil: entry 0).
func (k *ResidentTable) Decapsulate(idx []uint32, cts []byte) (sss []byte, errs []error, err error) {
	k.mu.RLock()
	defer k.mu.RUnlock()
	if k.t == nil || !k.private {
		return nil, nil, kem.ErrTypeMismatch
	}
	if len(cts)%k.s.CiphertextSize() != 0 {
		return nil, nil, kem.ErrCiphertextSize
	}
	n := len(cts) / k.s.CiphertextSize()
	if idx != nil && len(idx) != n {
		return nil, nil, kem.ErrTypeMismatch
	}
	if n == 0 {
		return []byte{}, []error{}, nil
	}
	sss = make([]byte, n*k.s.SharedKeySize())
	st := make([]byte, n)
	if err = status(C.circl_hip_mlkem_decaps_table(k.t, ptr32(idx), ptr(cts), ptr(sss), ptr(st), C.size_t(n)), "mlkem_decaps_table"); err != nil {
		return nil, nil, err
	}
	errs = make([]error, n)
	for i := range st {
		errs[i] = itemErr(st[i])
	}
	runtime.KeepAlive(k)
	return sss, errs, nil
}
