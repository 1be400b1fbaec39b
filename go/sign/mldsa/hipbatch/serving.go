//go:build cgo && hip

package hipbatch

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../include
#cgo LDFLAGS: -lcirclhip
#include <circl_hip.h>
*/
import "C"

// ServingScheme is a sign.Scheme for UNMODIFIED callers: code that holds a sign.Scheme value and calls Sign / Verify with one key
// and one message from whatever goroutine owns the connection (sign/sign.go:48-94; sign/mldsa/mldsa65/dilithium.go:283-343).
// UnmarshalBinaryPublicKey / UnmarshalBinaryPrivateKey / DeriveKey / GenerateKey return CIRCL's own key objects with a GPU-side
// counterpart attached -- a one-entry resident table (A, tr; for a private key also the NTT-domain s1, s2, t0).  A PUBLIC key's
// table gets an asynchronous queue and a reactor goroutine (reactor.go): Verify hands its request over and parks on a channel --
// no OS thread sleeps inside cgo per outstanding call -- and the requests that pile up while a launch runs share the next one.  A
// PRIVATE key's table keeps cross-caller coalescing of blocking calls (circl_hip_keytable_set_coalesce: the library has no submit
// form for signing).  A key object of the plain CIRCL scheme takes CIRCL's own path.  Signing is
// deterministic here exactly when it is in CIRCL (sign.SignatureOpts carries no randomness: dilithium.go:283-303 signs hedged only
// through SignTo with randomized = true, which this wrapper does not offer).
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain in the build image); tests/test_gpu_coalesce.py drives the same C entry points.

import (
	"crypto/rand"
	"fmt"
	"time"

	"github.com/cloudflare/circl/sign"
	"github.com/cloudflare/circl/sign/schemes"
)

// SetCoalesce lets the small Sign / Verify calls of concurrent goroutines through this table share launches (maxItems = 0: off).
func (r *ResidentKeys) SetCoalesce(maxItems int, maxWait time.Duration) error {
	r.mu.Lock()
	defer r.mu.Unlock()
	if r.t == nil {
		return sign.ErrTypeMismatch
	}
	if rc := C.circl_hip_keytable_set_coalesce(r.t, C.size_t(maxItems), C.uint32_t(maxWait.Microseconds())); rc != 0 {
		return fmt.Errorf("circl-hip keytable_set_coalesce: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	return nil
}

// ServingPublicKey / ServingPrivateKey: CIRCL's key object and its resident table of one.
type ServingPublicKey struct {
	sign.PublicKey
	table *ResidentKeys
	rx    *reactor
}
type ServingPrivateKey struct {
	sign.PrivateKey
	table *ResidentKeys
}

func (k *ServingPublicKey) Close() {
	if k.rx != nil {
		k.rx.stop()
		k.rx = nil
	}
	k.table.Close()
}
func (k *ServingPrivateKey) Close() { k.table.Close() }

type ServingScheme struct {
	Scheme
	device   int
	maxItems int
	maxWait  time.Duration
}

// Serving returns the serving form of an ML-DSA scheme of CIRCL's registry ("ML-DSA-44", "ML-DSA-65", "ML-DSA-87"), or nil.
func Serving(name string, device int) *ServingScheme {
	s := Wrap(schemes.ByName(name))
	if s == nil {
		return nil
	}
	return &ServingScheme{Scheme: *s, device: device, maxItems: 256}
}

// SetBatching applies to key objects made afterwards.
func (s *ServingScheme) SetBatching(maxItems int, maxWait time.Duration) {
	s.maxItems, s.maxWait = maxItems, maxWait
}

func (s *ServingScheme) public(pk sign.PublicKey) (sign.PublicKey, error) {
	b, err := pk.MarshalBinary()
	if err != nil {
		return nil, err
	}
	t, err := NewResidentPublicKeys(s.Scheme.Scheme, b, s.device)
	if err != nil {
		return nil, err
	}
	rx, err := t.startReactor(max(s.maxItems, 8), 1024)
	if err != nil {
		t.Close()
		return nil, err
	}
	return &ServingPublicKey{pk, t, rx}, nil
}

func (s *ServingScheme) private(sk sign.PrivateKey) (sign.PrivateKey, error) {
	b, err := sk.MarshalBinary()
	if err != nil {
		return nil, err
	}
	defer clear(b) // the marshalled copy of the private key does not outlive the call
	t, err := NewResidentPrivateKey(s.Scheme.Scheme, b, s.device)
	if err != nil {
		return nil, err
	}
	if err = t.SetCoalesce(s.maxItems, s.maxWait); err != nil {
		t.Close()
		return nil, err
	}
	return &ServingPrivateKey{sk, t}, nil
}

func (s *ServingScheme) UnmarshalBinaryPublicKey(b []byte) (sign.PublicKey, error) {
	pk, err := s.Scheme.UnmarshalBinaryPublicKey(b)
	if err != nil {
		return nil, err
	}
	return s.public(pk)
}

func (s *ServingScheme) UnmarshalBinaryPrivateKey(b []byte) (sign.PrivateKey, error) {
	sk, err := s.Scheme.UnmarshalBinaryPrivateKey(b)
	if err != nil {
		return nil, err
	}
	return s.private(sk)
}

func (s *ServingScheme) DeriveKey(seed []byte) (sign.PublicKey, sign.PrivateKey) {
	pk, sk := s.Scheme.DeriveKey(seed)
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

func (s *ServingScheme) GenerateKey() (sign.PublicKey, sign.PrivateKey, error) {
	seed := make([]byte, s.SeedSize())
	if _, err := rand.Read(seed); err != nil {
		return nil, nil, err
	}
	pk, sk := s.DeriveKey(seed)
	return pk, sk, nil
}

func context(opts *sign.SignatureOpts) string {
	if opts == nil {
		return ""
	}
	return opts.Context
}

// Sign: one row of the key's coalesced batch when sk is one of this scheme's key objects; panics like CIRCL's on a context longer
// than 255 bytes (dilithium.go:283-303).
func (s *ServingScheme) Sign(sk sign.PrivateKey, message []byte, opts *sign.SignatureOpts) []byte {
	r, ok := sk.(*ServingPrivateKey)
	if !ok {
		return s.Scheme.Sign(sk, message, opts)
	}
	sigs, err := r.table.Sign(nil, [][]byte{message}, []string{context(opts)}, nil)
	if err != nil {
		panic(err)
	}
	return sigs[0]
}

func (s *ServingScheme) Verify(pk sign.PublicKey, message, signature []byte, opts *sign.SignatureOpts) bool {
	r, ok := pk.(*ServingPublicKey)
	if !ok {
		return s.Scheme.Verify(pk, message, signature, opts)
	}
	if len(signature) != s.SignatureSize() || len(context(opts)) > 255 {
		return false
	}
	if r.rx == nil {
		res, err := r.table.Verify(nil, [][]byte{message}, signature, []string{context(opts)})
		return err == nil && res[0]
	}
	return r.rx.verify(message, signature, context(opts))
}
