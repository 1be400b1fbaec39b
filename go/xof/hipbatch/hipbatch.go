//go:build cgo && hip

// Package hipbatch routes batches of extendable-output-function computations (xof.SHAKE128 / SHAKE256, TurboSHAKE, and
// KangarooTwelve draft -10 = xof.K12D10) to libcirclhip.so (MI355X): one sponge per GPU lane, KangarooTwelve's leaves as
// one TurboSHAKE128 batch.  It serves callers that hash MANY independent messages (SLH-DSA's hypertree nodes, FrodoKEM's
// matrix rows, Merkle leaves); a single streaming xof.XOF stays on the CPU path (xof/xof.go:44-66).
//
// NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Go toolchain (see INTEGRATION.md); the C symbols are
// exercised by tests/test_gpu_prims.py (SHA-3 ShortMsgKATs, TurboSHAKE and KangarooTwelve I-D vectors) and by
// tests/cgo_shape_test.c with cgo's argument shapes.
package hipbatch

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -lcirclhip
#include <circl_hip.h>
*/
import "C"

import (
	"fmt"
	"unsafe"

	"github.com/cloudflare/circl/xof"
)

// AllDevices splits a batch into contiguous shards, one per visible GPU (no collective).
const AllDevices = -1

func ptr(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// blob lays messages out as one byte slice plus n+1 offsets (the ABI's ragged-row form) with the spare bytes the kernels'
// aligned loads may touch behind the last message.
func blob(msgs [][]byte) ([]byte, []C.uint64_t) {
	total := 0
	for _, m := range msgs {
		total += len(m)
	}
	b := make([]byte, 0, total+16)
	off := make([]C.uint64_t, len(msgs)+1)
	for i, m := range msgs {
		off[i] = C.uint64_t(len(b))
		b = append(b, m...)
	}
	off[len(msgs)] = C.uint64_t(len(b))
	return b[:len(b) : total+16], off
}

func offPtr(o []C.uint64_t) *C.uint64_t { return (*C.uint64_t)(unsafe.Pointer(&o[0])) }

// SumBatch computes outLen bytes of id over every message: out[i] = first outLen bytes of
// x := id.New(); x.Write(msgs[i]); x.Read(...)  (xof/xof.go:44-66).  id is xof.SHAKE128, xof.SHAKE256 or xof.K12D10
// (BLAKE2X has no GPU path here and returns an error).
func SumBatch(id xof.ID, msgs [][]byte, outLen int, device int) ([][]byte, error) {
	n := len(msgs)
	if n == 0 || outLen <= 0 {
		return make([][]byte, n), nil
	}
	b, off := blob(msgs)
	out := make([]byte, n*outLen)
	var rc C.int
	switch id {
	case xof.SHAKE128:
		rc = C.circl_hip_xof(168, 0x1f, 24, ptr(b[:cap(b)]), offPtr(off), ptr(out), C.size_t(outLen), C.size_t(n), C.int(device))
	case xof.SHAKE256:
		rc = C.circl_hip_xof(136, 0x1f, 24, ptr(b[:cap(b)]), offPtr(off), ptr(out), C.size_t(outLen), C.size_t(n), C.int(device))
	case xof.K12D10:
		rc = C.circl_hip_k12(ptr(b[:cap(b)]), offPtr(off), nil, nil, ptr(out), C.size_t(outLen), C.size_t(n), C.int(device))
	default:
		return nil, fmt.Errorf("circl-hip xof: no GPU path for %v", id)
	}
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip xof: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	rows := make([][]byte, n)
	for i := range rows {
		rows[i] = out[i*outLen : (i+1)*outLen : (i+1)*outLen]
	}
	return rows, nil
}

// TurboShakeBatch: TurboSHAKE128 (security 128) or TurboSHAKE256 (security 256) with domain-separation byte ds in
// 0x01..0x7f over every message (internal/sha3/shake.go:56-100 NewTurboShake128 / NewTurboShake256).
func TurboShakeBatch(security int, ds byte, msgs [][]byte, outLen int, device int) ([][]byte, error) {
	rate := 168
	if security == 256 {
		rate = 136
	} else if security != 128 {
		return nil, fmt.Errorf("circl-hip xof: TurboSHAKE security %d", security)
	}
	n := len(msgs)
	if n == 0 || outLen <= 0 {
		return make([][]byte, n), nil
	}
	b, off := blob(msgs)
	out := make([]byte, n*outLen)
	rc := C.circl_hip_xof(C.int(rate), C.int(ds), 12, ptr(b[:cap(b)]), offPtr(off), ptr(out), C.size_t(outLen), C.size_t(n), C.int(device))
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip xof: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	rows := make([][]byte, n)
	for i := range rows {
		rows[i] = out[i*outLen : (i+1)*outLen : (i+1)*outLen]
	}
	return rows, nil
}

// K12Batch: xof/k12 Draft10Sum(out[i], msgs[i], ctxs[i]) for every i (xof/k12/k12.go); ctxs may be nil (all contexts empty).
func K12Batch(msgs, ctxs [][]byte, outLen int, device int) ([][]byte, error) {
	n := len(msgs)
	if ctxs != nil && len(ctxs) != n {
		return nil, fmt.Errorf("circl-hip xof: %d contexts for %d messages", len(ctxs), n)
	}
	if n == 0 || outLen <= 0 {
		return make([][]byte, n), nil
	}
	b, off := blob(msgs)
	out := make([]byte, n*outLen)
	var rc C.int
	if ctxs == nil {
		rc = C.circl_hip_k12(ptr(b[:cap(b)]), offPtr(off), nil, nil, ptr(out), C.size_t(outLen), C.size_t(n), C.int(device))
	} else {
		cb, coff := blob(ctxs)
		rc = C.circl_hip_k12(ptr(b[:cap(b)]), offPtr(off), ptr(cb[:cap(cb)]), offPtr(coff), ptr(out), C.size_t(outLen), C.size_t(n), C.int(device))
	}
	if rc != 0 {
		return nil, fmt.Errorf("circl-hip xof: error %d: %s", int(rc), C.GoString(C.circl_hip_last_error()))
	}
	rows := make([][]byte, n)
	for i := range rows {
		rows[i] = out[i*outLen : (i+1)*outLen : (i+1)*outLen]
	}
	return rows, nil
}
