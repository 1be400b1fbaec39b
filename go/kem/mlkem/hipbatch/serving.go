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
// DeriveKeyPair / GenerateKeyPair parse the key ONCE on the device (a one-entry resident table with an ASYNCHRONOUS queue:
// circl_hip_keytable_async_start) and start the key's reactor (reactor.go): ONE goroutine that submits and polls.  Every later
// single operation on that object hands its request to the reactor and PARKS ON A CHANNEL -- in the Go scheduler, not in a
// blocking cgo call: round 5's form held one OS thread per outstanding call (a goroutine inside cgo keeps its M), ten thousand
// concurrent handshakes were ten thousand sleeping threads and 13 us of host CPU per call (profiles/r05_concurrent_final.txt);
// now the number of threads does not depend on the number of callers, and the requests that pile up while a launch runs share the
// next one.  One caller alone pays a launch and a completion per call (tens of microseconds).  A key object of the plain CIRCL
// scheme is accepted everywhere too and takes CIRCL's own path.  Results are identical either way.
//
//	s := hipbatch.Serving("ML-KEM-768", 0)          // device 0
//	sk, _ := s.UnmarshalBinaryPrivateKey(dkBytes)    // parsed on the GPU once; its reactor starts
//	ss, _ := s.Decapsulate(sk, ct)                   // from any number of goroutines: each parks until its row is back
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain in the build image); tests/cgo_shape_test.c (async_submit_poll)
// and tests/test_gpu_async.py drive the same C entry points in the same shape.

import (
	"crypto/rand"
	"sync"
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
	window   int

	callOnce sync.Once // the call queue for encapsulations to keys that are NOT resident: opened at the first such call
	call     *reactor
	callErr  error
}

// Serving returns the serving form of an ML-KEM scheme of CIRCL's registry, or nil.  Device batches hold up to 2048 items, a key's
// reactor keeps up to 4096 requests in flight (SetBatching changes both).  device must name ONE device (a reactor owns one
// queue); for several GPUs make one ServingScheme per device and spread the key objects.
func Serving(name string, device int) *ServingScheme {
	s := Wrap(schemes.ByName(name))
	if s == nil {
		return nil
	}
	if _, ok := params[s.Name()]; !ok { // round-3 Kyber has no resident tables
		return nil
	}
	if device < 0 {
		return nil
	}
	return &ServingScheme{Scheme: *s, device: device, maxItems: 2048, window: 4096}
}

// Close stops the scheme's call queue (key objects are closed one by one).
func (s *ServingScheme) Close() {
	s.callOnce.Do(func() {}) // (a queue that was never opened stays unopened)
	if s.call != nil {
		s.call.stop()
		s.call = nil
	}
}

// SetBatching applies to key objects made afterwards: the largest device batch and the requests one key keeps in flight.
func (s *ServingScheme) SetBatching(maxItems, window int) {
	s.maxItems, s.window = maxItems, max(window, 1)
}

func (s *ServingScheme) public(pk kem.PublicKey) (kem.PublicKey, error) {
	r, err := s.Scheme.ResidentPublicKey(pk, s.device)
	if err != nil {
		return nil, err
	}
	if r.rx, err = r.table.startReactor(s.maxItems, s.window); err != nil {
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
	if r.rx, err = r.table.startReactor(s.maxItems, s.window); err != nil {
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

// EncapsulateDeterministically: one row of whatever launch the key's reactor submits next, when pk is one of this scheme's key
// objects; the calling goroutine parks on a channel meanwhile (reactor.go).
func (s *ServingScheme) EncapsulateDeterministically(pk kem.PublicKey, seed []byte) (ct, ss []byte, err error) {
	r, ok := pk.(*ResidentPublicKey)
	if !ok {
		// a key that was not made by this scheme -- a TLS server's case: every handshake encapsulates once, to the CLIENT'S ephemeral key
		// (kem/hybrid/hybrid.go:271-300).  Its packed form travels with the request through the scheme's call queue (circl_hip_queue):
		// the requests of the connections that arrive while a launch runs share the next one.
		if len(seed) != s.EncapsulationSeedSize() {
			return nil, nil, kem.ErrSeedSize
		}
		s.callOnce.Do(func() { s.call, s.callErr = newCallReactor(s.Scheme.Scheme, s.device, s.maxItems, s.window) })
		if s.callErr != nil {
			return s.Scheme.EncapsulateDeterministically(pk, seed) // no device: CIRCL's own path
		}
		b, err := pk.MarshalBinary()
		if err != nil {
			return nil, nil, err
		}
		rp := s.call.doKeyed(b, seed)
		return rp.ct, rp.ss, rp.err
	}
	if len(seed) != s.EncapsulationSeedSize() {
		return nil, nil, kem.ErrSeedSize
	}
	if r.rx == nil { // a resident key made without a reactor (Scheme.ResidentPublicKey): the blocking call
		return r.EncapsulateBatch(seed)
	}
	rp := r.rx.do(seed)
	return rp.ct, rp.ss, rp.err
}

func (s *ServingScheme) Decapsulate(sk kem.PrivateKey, ct []byte) ([]byte, error) {
	r, ok := sk.(*ResidentPrivateKey)
	if !ok {
		return s.Scheme.Decapsulate(sk, ct)
	}
	if len(ct) != s.CiphertextSize() {
		return nil, kem.ErrCiphertextSize
	}
	if r.rx == nil {
		return r.DecapsulateBatch(ct)
	}
	rp := r.rx.do(ct)
	return rp.ss, rp.err
}

// SetCallCoalescing is the same switch for keys that come WITH the call -- a TLS 1.3 server encapsulates once per handshake, to the
// client's ephemeral key share (kem/hybrid/hybrid.go:271-300), so there is no key object to attach a table to: small EncapsulateBatch /
// DecapsulateBatch calls (one or a few items) from concurrent goroutines share launches per scheme and device
// (circl_hip_set_coalesce; also covers sign/mldsa's VerifyBatch and kem/hybrid's batches, which live in the same library).
// Process-wide; call it once at start-up.  maxItems = 0 switches new joins off.
func SetCallCoalescing(maxItems int, maxWait time.Duration) error {
	return status(C.circl_hip_set_coalesce(C.size_t(maxItems), C.uint32_t(maxWait.Microseconds())), "set_coalesce")
}
