//go:build cgo && hip

// Package hipbatch is the batch counterpart of simd/keccakf1600: where StateX2 / StateX4 permute two or four interleaved
// states with AVX2 / NEON (simd/keccakf1600/f1600x.go:30-44, :77-129), PermuteBatch permutes n independent states on
// libcirclhip.so (MI355X), one state per GPU lane.  States are PLAIN [25]uint64 arrays (not interleaved): state i is
// states[25*i : 25*i+25].
//
// NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Go toolchain (see INTEGRATION.md); the C symbol is exercised
// by tests/test_gpu_prims.py against the reference's zero-state vector (simd/keccakf1600/f1600x_test.go:9-19) and the oracle.
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
)

// AllDevices splits a batch into contiguous shards, one per visible GPU (no collective).
const AllDevices = -1

// PermuteBatch applies Keccak-f[1600] (turbo = false, 24 rounds) or the 12-round TurboSHAKE permutation (turbo = true:
// StateX4.Initialize(true), f1600x.go:60-75) to every state in place.
func PermuteBatch(states []uint64, turbo bool, device int) error {
	if len(states)%25 != 0 {
		return fmt.Errorf("circl-hip keccakf1600: %d words is not a whole number of states", len(states))
	}
	if len(states) == 0 {
		return nil
	}
	rounds := 24
	if turbo {
		rounds = 12
	}
	rc := C.circl_hip_keccak_f1600((*C.uint64_t)(unsafe.Pointer(&states[0])), C.size_t(len(states)/25), C.int(rounds), C.int(device))
	if rc != 0 {
		return fmt.Errorf("circl-hip keccakf1600: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	return nil
}

// IsEnabled mirrors keccakf1600.IsEnabledX4 (f1600x.go:46-50): true when a GPU is visible to the library.
func IsEnabled() bool { return C.circl_hip_device_count() > 0 }
