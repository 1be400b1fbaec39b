//go:build cgo && hip

package hipbatch

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../include
#cgo LDFLAGS: -lcirclhip
#include <stdlib.h>
#include <circl_hip.h>
*/
import "C"

// reactor.go -- sign.Scheme.Verify for ordinary goroutines on the asynchronous table API (circl_hip_keytable_async_start,
// circl_hip_mldsa_verify_table_submit, circl_hip_poll: include/circl_hip.h).  The design and its reasons are those of
// go/kem/mlkem/hipbatch/reactor.go: a request goroutine hands {message, signature, context} to the key's ONE reactor goroutine and
// parks on a channel (no OS thread asleep inside a cgo call per outstanding Verify); the reactor packs what is pending into one
// submitted call (messages and contexts as blobs with offsets, signatures as rows: all Go memory, copied by the library before the
// submit returns), polls the head of its ticket FIFO and sends the verdicts back.  The one array the library writes LATER -- the
// verdict bytes -- lives in a C.malloc'ed ring, because cgo forbids C to keep a Go pointer after the call returns.  An eventfd read
// through the runtime poller nudges the reactor when a batch is done.  Signing keeps the blocking coalesced call (serving.go): the
// library has no submit form for it (a signature is rounds of launches; a server signs once per handshake, it verifies chains).
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain in the build image); tests/test_gpu_async.py
// (test_mldsa_verify_submit_equals_the_oracle) and tests/race_driver.cpp (Dsa::async_calls) drive the same entry points.

import (
	"errors"
	"os"
	"sync"
	"syscall"
	"time"
	"unsafe"

	"github.com/cloudflare/circl/sign"
)

type vrequest struct {
	msg, sig []byte
	ctx      string
	done     chan bool // capacity 1
}

type vflight struct {
	ticket C.uint64_t
	slot   int
	reqs   []*vrequest
}

type reactor struct {
	r       *ResidentKeys
	reqs    chan *vrequest
	wake    chan struct{}
	quit    chan struct{}
	stopped sync.WaitGroup
	efd     *os.File

	window, callMax int
	sigSize         int
	arenaOK         unsafe.Pointer // window verdict bytes
	tail, inUse     int
	fifo            []vflight
	sigs, mblob     []byte
	cblob           []byte
	moff, coff      []uint64
}

var errClosed = errors.New("circl-hip: the key object was closed")

func (k *ResidentKeys) startReactor(maxItems, window int) (*reactor, error) {
	k.mu.Lock()
	defer k.mu.Unlock()
	if k.t == nil || k.private {
		return nil, sign.ErrTypeMismatch
	}
	if rc := C.circl_hip_keytable_async_start(k.t, C.size_t(maxItems), 0, 1); rc != 0 {
		return nil, errors.New("circl-hip keytable_async_start: " + C.GoString(C.circl_hip_last_error()))
	}
	x := &reactor{r: k, reqs: make(chan *vrequest, window), wake: make(chan struct{}, 1), quit: make(chan struct{}),
		window: window, callMax: max(1, maxItems/4), sigSize: k.s.SignatureSize()}
	x.arenaOK = C.malloc(C.size_t(window))
	x.sigs = make([]byte, x.callMax*x.sigSize)
	if fd := int(C.circl_hip_keytable_eventfd(k.t, 0)); fd >= 0 {
		if d, err := syscall.Dup(fd); err == nil { // the library owns its descriptor: the os.File gets a dup (same counter)
			syscall.SetNonblock(d, true)
			x.efd = os.NewFile(uintptr(d), "circl-hip-eventfd")
		}
	}
	x.stopped.Add(1)
	go x.loop()
	if x.efd != nil {
		go func() {
			var cnt [8]byte
			for {
				if _, err := x.efd.Read(cnt[:]); err != nil {
					return
				}
				select {
				case x.wake <- struct{}{}:
				default:
				}
			}
		}()
	}
	return x, nil
}

// verify is what a request goroutine runs: false also when the key object is closed under it.
func (x *reactor) verify(msg, sig []byte, ctx string) bool {
	rq := &vrequest{msg, sig, ctx, make(chan bool, 1)}
	select {
	case x.reqs <- rq:
	case <-x.quit:
		return false
	}
	select {
	case ok := <-rq.done:
		return ok
	case <-x.quit:
		return false
	}
}

func (x *reactor) loop() {
	defer x.stopped.Done()
	var pending []*vrequest
	blocked := false
	for {
		switch {
		case len(pending) == 0 && len(x.fifo) == 0:
			select {
			case rq := <-x.reqs:
				pending = append(pending, rq)
			case <-x.quit:
				return
			}
		case len(pending) == 0 || x.inUse == x.window || blocked:
			var tick <-chan time.Time
			if x.efd == nil {
				tick = time.After(50 * time.Microsecond)
			}
			select {
			case rq := <-x.reqs:
				pending = append(pending, rq)
			case <-x.wake:
			case <-tick:
			case <-x.quit:
				for _, rq := range pending {
					rq.done <- false
				}
				return
			}
		}
		blocked = false
	more:
		for len(pending) < x.window {
			select {
			case rq := <-x.reqs:
				pending = append(pending, rq)
			default:
				break more
			}
		}
		x.reap()
		for len(pending) > 0 && x.inUse < x.window {
			n := min(len(pending), x.callMax, x.window-x.inUse, x.window-x.tail)
			if !x.submit(pending[:n]) {
				blocked = true
				break
			}
			pending = pending[n:]
		}
		x.reap()
	}
}

func (x *reactor) submit(reqs []*vrequest) bool {
	n := len(reqs)
	x.mblob, x.cblob, x.moff, x.coff = x.mblob[:0], x.cblob[:0], x.moff[:0], x.coff[:0]
	for i, rq := range reqs {
		copy(x.sigs[i*x.sigSize:], rq.sig)
		x.moff = append(x.moff, uint64(len(x.mblob)))
		x.mblob = append(x.mblob, rq.msg...)
		x.coff = append(x.coff, uint64(len(x.cblob)))
		x.cblob = append(x.cblob, rq.ctx...)
	}
	x.moff = append(x.moff, uint64(len(x.mblob)))
	x.coff = append(x.coff, uint64(len(x.cblob)))
	x.mblob = append(x.mblob, 0) // (never empty: &blob[0] below)
	x.cblob = append(x.cblob, 0)
	var ticket C.uint64_t
	x.r.mu.RLock()
	if x.r.t == nil {
		x.r.mu.RUnlock()
		for _, rq := range reqs {
			rq.done <- false
		}
		return true
	}
	rc := C.circl_hip_mldsa_verify_table_submit(x.r.t, nil, (*C.uint8_t)(unsafe.Pointer(&x.sigs[0])), (*C.uint8_t)(unsafe.Pointer(&x.mblob[0])),
		(*C.uint64_t)(unsafe.Pointer(&x.moff[0])), (*C.uint8_t)(unsafe.Pointer(&x.cblob[0])), (*C.uint64_t)(unsafe.Pointer(&x.coff[0])),
		(*C.uint8_t)(unsafe.Add(x.arenaOK, x.tail)), C.size_t(n), &ticket)
	x.r.mu.RUnlock()
	if rc == C.CIRCL_HIP_EAGAIN {
		return false
	}
	if rc != 0 { // (a message beyond a quarter of the queue's blob area, say): the blocking call decides
		for _, rq := range reqs {
			res, err := x.r.Verify(nil, [][]byte{rq.msg}, rq.sig, []string{rq.ctx})
			rq.done <- err == nil && res[0]
		}
		return true
	}
	x.fifo = append(x.fifo, vflight{ticket, x.tail, append([]*vrequest(nil), reqs...)})
	x.tail = (x.tail + n) % x.window
	x.inUse += n
	return true
}

func (x *reactor) reap() {
	for len(x.fifo) > 0 {
		f := &x.fifo[0]
		var state C.int8_t
		x.r.mu.RLock()
		if x.r.t != nil {
			C.circl_hip_poll(x.r.t, &f.ticket, 1, &state)
		} else {
			state = -1
		}
		x.r.mu.RUnlock()
		if state == 0 {
			return
		}
		for i, rq := range f.reqs {
			rq.done <- state == 1 && *(*byte)(unsafe.Add(x.arenaOK, f.slot+i)) == 1 // a failed batch verifies nothing
		}
		x.inUse -= len(f.reqs)
		x.fifo = x.fifo[1:]
	}
}

func (x *reactor) stop() {
	close(x.quit)
	x.stopped.Wait()
	x.r.mu.Lock()
	if x.r.t != nil {
		C.circl_hip_keytable_async_stop(x.r.t) // finishes what was submitted: nothing writes the arena afterwards
	}
	x.r.mu.Unlock()
	if x.efd != nil {
		x.efd.Close()
	}
	for _, f := range x.fifo {
		for _, rq := range f.reqs {
			rq.done <- false
		}
	}
	C.free(x.arenaOK)
}
