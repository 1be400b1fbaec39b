//go:build cgo && hip

package hipbatch

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../include
#cgo LDFLAGS: -lcirclhip
#include <circl_hip.h>
*/
import "C"

// ServingScheme is a kem.Scheme for UNMODIFIED callers: code that holds a kem.Scheme value and calls Encapsulate /
// Decapsulate one key and one item at a time from whatever goroutine owns the connection -- the shape of every consumer
// in the reference (kem/mlkem/mlkem768/kyber.go:347-386; hpke/algs.go:283-285 picks its kem.Scheme from a table).
//
// Its key objects are CIRCL's own with a GPU-side counterpart attached: UnmarshalBinaryPublicKey / UnmarshalBinaryPrivateKey /
// DeriveKeyPair / GenerateKeyPair parse the key ONCE on the device (a one-entry resident table with cross-caller coalescing
// switched on: circl_hip_keytable_set_coalesce), and every later single operation on that object is one row of whatever
// launch the concurrent callers of the same key share.  One caller alone pays a launch and a wait per call (tens of
// microseconds); hundreds of goroutines on one server key share launches.  A key object of the plain CIRCL scheme is
// accepted everywhere too and takes CIRCL's own path.  Results are identical either way.
//
//	s := hipbatch.Serving("ML-KEM-768", 0)          // device 0; hipbatch.AllDevices replicates every key object
//	sk, _ := s.UnmarshalBinaryPrivateKey(dkBytes)    // parsed on the GPU once
//	ss, _ := s.Decapsulate(sk, ct)                   // from any number of goroutines: coalesced
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain in the build image); tests/cgo_shape_test.c (coalesced_single_calls)
// and tests/test_gpu_coalesce.py drive the same C entry points in the same shape.

import (
	"crypto/rand"
	"time"

	"github.com/cloudflare/circl/kem"
	"github.com/cloudflare/circl/kem/schemes"
)

// SetCoalesce lets the small calls of concurrent goroutines through this table share launches: batches of up to maxItems
// items are flushed as soon as the device has room (maxWait = 0) or after maxWait at the latest.  maxItems = 0 switches it off.
// Call it before the table is shared between goroutines.
func (k *ResidentTable) SetCoalesce(maxItems int, maxWait time.Duration) error {
	k.mu.Lock()
	defer k.mu.Unlock()
	if k.t == nil {
		return kem.ErrTypeMismatch
	}
	return status(C.circl_hip_keytable_set_coalesce(k.t, C.size_t(maxItems), C.uint32_t(maxWait.Microseconds())), "keytable_set_coalesce")
}

type ServingScheme struct {
	Scheme
	device   int
	maxItems int
	maxWait  time.Duration
}

// Serving returns the serving form of an ML-KEM scheme of CIRCL's registry, or nil.  Batches hold up to 256 items and are
// never delayed (SetBatching changes both).
func Serving(name string, device int) *ServingScheme {
	s := Wrap(schemes.ByName(name))
	if s == nil {
		return nil
	}
	if _, ok := params[s.Name()]; !ok { // round-3 Kyber has no resident tables
		return nil
	}
	return &ServingScheme{Scheme: *s, device: device, maxItems: 256}
}

// SetBatching applies to key objects made afterwards.
func (s *ServingScheme) SetBatching(maxItems int, maxWait time.Duration) {
	s.maxItems, s.maxWait = maxItems, maxWait
}

func (s *ServingScheme) public(pk kem.PublicKey) (kem.PublicKey, error) {
	r, err := s.Scheme.ResidentPublicKey(pk, s.device)
	if err != nil {
		return nil, err
	}
	if err = r.table.SetCoalesce(s.maxItems, s.maxWait); err != nil {
		r.Close()
		return nil, err
	}
	return r, nil
}

func (s *ServingScheme) private(sk kem.PrivateKey) (kem.PrivateKey, error) {
	r, err := s.Scheme.ResidentPrivateKey(sk, s.device)
	if err != nil {
		return nil, err
	}
	if err = r.table.SetCoalesce(s.maxItems, s.maxWait); err != nil {
		r.Close()
		return nil, err
	}
	return r, nil
}

func (s *ServingScheme) UnmarshalBinaryPublicKey(b []byte) (kem.PublicKey, error) {
	pk, err := s.Scheme.UnmarshalBinaryPublicKey(b)
	if err != nil {
		return nil, err
	}
	return s.public(pk)
}

func (s *ServingScheme) UnmarshalBinaryPrivateKey(b []byte) (kem.PrivateKey, error) {
	sk, err := s.Scheme.UnmarshalBinaryPrivateKey(b)
	if err != nil {
		return nil, err
	}
	return s.private(sk)
}

func (s *ServingScheme) DeriveKeyPair(seed []byte) (kem.PublicKey, kem.PrivateKey) {
	pk, sk := s.Scheme.DeriveKeyPair(seed)
	rpk, err := s.public(pk)
	if err != nil {
		return pk, sk // no device: CIRCL's objects, CIRCL's path
	}
	rsk, err := s.private(sk)
	if err != nil {
		return pk, sk
	}
	return rpk, rsk
}

func (s *ServingScheme) GenerateKeyPair() (kem.PublicKey, kem.PrivateKey, error) {
	seed := make([]byte, s.SeedSize())
	if _, err := rand.Read(seed); err != nil {
		return nil, nil, err
	}
	pk, sk := s.DeriveKeyPair(seed)
	return pk, sk, nil
}

func (s *ServingScheme) Encapsulate(pk kem.PublicKey) (ct, ss []byte, err error) {
	seed := make([]byte, s.EncapsulationSeedSize())
	if _, err = rand.Read(seed); err != nil {
		return nil, nil, err
	}
	return s.EncapsulateDeterministically(pk, seed)
}

// EncapsulateDeterministically: one row of the key's coalesced batch when pk is one of this scheme's key objects.
func (s *ServingScheme) EncapsulateDeterministically(pk kem.PublicKey, seed []byte) (ct, ss []byte, err error) {
	r, ok := pk.(*ResidentPublicKey)
	if !ok {
		return s.Scheme.EncapsulateDeterministically(pk, seed)
	}
	if len(seed) != s.EncapsulationSeedSize() {
		return nil, nil, kem.ErrSeedSize
	}
	return r.EncapsulateBatch(seed)
}

func (s *ServingScheme) Decapsulate(sk kem.PrivateKey, ct []byte) ([]byte, error) {
	r, ok := sk.(*ResidentPrivateKey)
	if !ok {
		return s.Scheme.Decapsulate(sk, ct)
	}
	if len(ct) != s.CiphertextSize() {
		return nil, kem.ErrCiphertextSize
	}
	return r.DecapsulateBatch(ct)
}

// SetCallCoalescing is the same switch for keys that come WITH the call -- a TLS 1.3 server encapsulates once per handshake, to the
// client's ephemeral key share (kem/hybrid/hybrid.go:271-300), so there is no key object to attach a table to: small EncapsulateBatch /
// DecapsulateBatch calls (one or a few items) from concurrent goroutines share launches per scheme and device
// (circl_hip_set_coalesce; also covers sign/mldsa's VerifyBatch and kem/hybrid's batches, which live in the same library).
// Process-wide; call it once at start-up.  maxItems = 0 switches new joins off.
func SetCallCoalescing(maxItems int, maxWait time.Duration) error {
	return status(C.circl_hip_set_coalesce(C.size_t(maxItems), C.uint32_t(maxWait.Microseconds())), "set_coalesce")
}
