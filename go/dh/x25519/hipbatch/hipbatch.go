//go:build cgo && hip

// Package hipbatch routes batches of X25519 operations (dh/x25519 KeyGen / Shared) to libcirclhip.so (MI355X): one
// Montgomery ladder (Shared) or one fixed-base comb (KeyGen) per GPU lane.
//
// NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Go toolchain (see INTEGRATION.md); the C symbols are
// exercised by tests/test_gpu_x25519.py against the reference's RFC 7748 and Wycheproof vectors.
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

	"github.com/cloudflare/circl/dh/x25519"
)

// AllDevices splits a batch into contiguous shards, one per visible GPU (no collective).
const AllDevices = -1

func keyPtr(k []x25519.Key) *C.uint8_t {
	if len(k) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&k[0])) // []x25519.Key is n contiguous 32-byte arrays
}

// KeyGenBatch: public[i] = x25519.KeyGen(secret[i])  (dh/x25519/key.go:34-36)
func KeyGenBatch(public, secret []x25519.Key, device int) error {
	if len(public) != len(secret) {
		return fmt.Errorf("circl-hip x25519: %d public keys for %d secrets", len(public), len(secret))
	}
	rc := C.circl_hip_x25519(keyPtr(secret), nil, keyPtr(public), nil, C.size_t(len(secret)), C.int(device))
	if rc != 0 {
		return fmt.Errorf("circl-hip x25519: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	return nil
}

// SharedBatch: ok[i] = x25519.Shared(&shared[i], &secret[i], &public[i])  (dh/x25519/key.go:41-47); ok[i] is false for
// a low-order public key, and shared[i] is then all zero.
func SharedBatch(shared, secret, public []x25519.Key, device int) (ok []bool, err error) {
	n := len(secret)
	if len(shared) != n || len(public) != n {
		return nil, fmt.Errorf("circl-hip x25519: slice lengths differ")
	}
	st := make([]byte, n)
	var stp *C.uint8_t
	if n > 0 {
		stp = (*C.uint8_t)(unsafe.Pointer(&st[0]))
	}
	rc := C.circl_hip_x25519(keyPtr(secret), keyPtr(public), keyPtr(shared), stp, C.size_t(n), C.int(device))
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip x25519: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	ok = make([]bool, n)
	for i, b := range st {
		ok[i] = b != 0
	}
	return ok, nil
}
