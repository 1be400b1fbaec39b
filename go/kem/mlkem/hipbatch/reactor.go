//go:build cgo && hip

package hipbatch

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../include
#cgo LDFLAGS: -lcirclhip
#include <stdlib.h>
#include <string.h>
#include <circl_hip.h>
*/
import "C"

// reactor.go -- the asynchronous table API (circl_hip_keytable_async_start / *_table_submit / circl_hip_poll,
// include/circl_hip.h) under a kem.Scheme whose callers are ordinary goroutines.
//
// Why not one blocking cgo call per goroutine (what serving.go did in round 5): a goroutine inside a blocking cgo call holds
// an OS thread (an M) for as long as the call sleeps.  Ten thousand concurrent handshakes -- the callers of
// kem.Scheme.Encapsulate / Decapsulate in hpke/algs.go:283-285 and kem/hybrid/hybrid.go:95-99 -- were ten thousand threads
// asleep on a futex inside the library, and the sleep + wake of each was most of the call's host cost (7.6 of 13 us,
// profiles/r05_concurrent_final.txt).  Here nothing ever blocks in C:
//
//	request goroutine   sends its request on the reactor's channel and PARKS on its own reply channel (Go scheduler, no M)
//	reactor goroutine   ONE per resident table (= per device queue): takes everything that is pending, copies the inputs
//	                    into one contiguous buffer and submits them with ONE cgo call (n items, one ticket; the call copies
//	                    the inputs and returns at once), polls the head of its ticket FIFO (one atomic load in C), copies
//	                    finished rows out of the C arena into Go slices and sends them to their owners
//	eventfd goroutine   reads the queue's eventfd through the runtime's network poller (os.File on a non-blocking fd): it
//	                    parks in the scheduler too and nudges the reactor whenever the library has finished a batch
//
// cgo's pointer rules forbid C to keep a Go pointer after the call returns, and the library writes a submitted call's
// results LATER: the output rows therefore live in a C-allocated arena (C.malloc, one slot ring per reactor), never in Go
// memory.  Inputs are plain Go slices: the library copies them before the submit returns.
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain in the build image).  tests/cgo_shape_test.c (async_submit_poll)
// drives the same C entry points in this shape -- producers parked on a condition variable, one submitting / polling thread,
// a malloc'ed output arena -- and tests/test_gpu_async.py checks the bytes against the oracle.

import (
	"errors"
	"os"
	"sync"
	"syscall"
	"time"
	"unsafe"

	"github.com/cloudflare/circl/kem"
)

type reply struct {
	ct, ss []byte
	err    error
}

type request struct {
	key  []byte     // call queues only: the packed key that comes with the item
	in   []byte     // encapsulation seed or ciphertext
	done chan reply // capacity 1: the reactor never blocks on an owner
}

// one submitted call: `n` requests in arena slots [slot, slot + n), one ticket
type flight struct {
	ticket C.uint64_t
	slot   int
	reqs   []*request
}

// A reactor serves ONE queue of the library: a resident table's (k != nil: circl_hip_keytable_async_start) or a call queue's (q != nil:
// circl_hip_queue_open -- every item brings its own key: a TLS server's encapsulation to the client's ephemeral share).
type reactor struct {
	k       *ResidentTable
	q       *C.circl_hip_queue
	decaps  bool
	keySize int
	keys    []byte // call queues: the pending calls' key rows, side by side
	reqs    chan *request
	wake    chan struct{}
	quit    chan struct{}
	stopped sync.WaitGroup
	efd     *os.File

	window, callMax    int // arena slots; items per submitted call (max_items / 4)
	inSize, ctSize     int
	arenaCT, arenaSS   unsafe.Pointer
	arenaST            unsafe.Pointer
	head, tail, inUse  int // the arena is a ring of slots: calls finish in issue order
	fifo               []flight
	scratch            []byte
	errQueueClosed     error
}

// startReactor switches the table's asynchronous queue on and starts its goroutines.  maxItems: largest device batch
// (a submitted call holds at most maxItems / 4 requests); window: requests in flight at most.
func (k *ResidentTable) startReactor(maxItems, window int) (*reactor, error) {
	k.mu.Lock()
	defer k.mu.Unlock()
	if k.t == nil {
		return nil, kem.ErrTypeMismatch
	}
	if err := status(C.circl_hip_keytable_async_start(k.t, C.size_t(maxItems), 0, 1), "keytable_async_start"); err != nil {
		return nil, err
	}
	r := &reactor{k: k, decaps: k.private, reqs: make(chan *request, window), wake: make(chan struct{}, 1), quit: make(chan struct{}),
		window: window, callMax: max(1, maxItems/4), ctSize: k.s.CiphertextSize(), errQueueClosed: errors.New("circl-hip: the key object was closed")}
	if r.decaps {
		r.inSize = k.s.CiphertextSize()
	} else {
		r.inSize = k.s.EncapsulationSeedSize()
		r.arenaCT = C.malloc(C.size_t(window * r.ctSize))
	}
	r.arenaSS = C.malloc(C.size_t(window * 32))
	r.arenaST = C.malloc(C.size_t(window))
	r.scratch = make([]byte, r.callMax*r.inSize)
	r.start(int(C.circl_hip_keytable_eventfd(k.t, 0)))
	return r, nil
}

// newCallReactor opens a call queue for s's encapsulations with the key in the call (circl_hip_queue_open) and starts its goroutines.
func newCallReactor(s kem.Scheme, device, maxItems, window int) (*reactor, error) {
	p, ok := params[s.Name()]
	if !ok {
		return nil, kem.ErrTypeMismatch
	}
	r := &reactor{reqs: make(chan *request, window), wake: make(chan struct{}, 1), quit: make(chan struct{}), window: window,
		callMax: max(1, maxItems/4), ctSize: s.CiphertextSize(), keySize: s.PublicKeySize(), inSize: s.EncapsulationSeedSize(),
		errQueueClosed: errors.New("circl-hip: the scheme's call queue was closed")}
	if err := status(C.circl_hip_queue_open(C.int(C.CIRCL_HIP_QUEUE_MLKEM_ENCAPS), p, C.int(device), C.size_t(maxItems), 1, &r.q), "queue_open"); err != nil {
		return nil, err
	}
	r.arenaCT = C.malloc(C.size_t(window * r.ctSize))
	r.arenaSS = C.malloc(C.size_t(window * 32))
	r.arenaST = C.malloc(C.size_t(window))
	r.scratch = make([]byte, r.callMax*r.inSize)
	r.keys = make([]byte, r.callMax*r.keySize)
	r.start(int(C.circl_hip_queue_eventfd(r.q)))
	return r, nil
}

// start: the queue's eventfd through the runtime poller -- a Read parks the goroutine, not a thread.  The library owns its descriptor, so
// the os.File gets a dup of it (same counter; closed by stop, which is also what ends the completions goroutine) -- and the goroutines.
func (r *reactor) start(fd int) {
	if fd >= 0 {
		if d, err := syscall.Dup(fd); err == nil {
			syscall.SetNonblock(d, true)
			r.efd = os.NewFile(uintptr(d), "circl-hip-eventfd")
		}
	}
	r.stopped.Add(1)
	go r.loop()
	if r.efd != nil {
		go r.completions()
	}
}

func (r *reactor) completions() {
	var cnt [8]byte
	for {
		if _, err := r.efd.Read(cnt[:]); err != nil {
			return // the queue is gone (stop)
		}
		select {
		case r.wake <- struct{}{}:
		default: // a nudge is already pending
		}
	}
}

// do is what a request goroutine runs: hand the request over, park until the reply is there (key: call queues only).
func (r *reactor) do(in []byte) reply { return r.doKeyed(nil, in) }

func (r *reactor) doKeyed(key, in []byte) reply {
	rq := &request{key: key, in: in, done: make(chan reply, 1)}
	select {
	case r.reqs <- rq:
	case <-r.quit:
		return reply{err: r.errQueueClosed}
	}
	select {
	case rp := <-rq.done:
		return rp
	case <-r.quit:
		return reply{err: r.errQueueClosed}
	}
}

func (r *reactor) loop() {
	defer r.stopped.Done()
	var pending []*request
	blocked := false // the last submit found every device batch busy (CIRCL_HIP_EAGAIN): a completion has to come first
	for {
		// ---- wait, parked in the scheduler, until there is something to do ----
		switch {
		case len(pending) == 0 && len(r.fifo) == 0:
			select {
			case rq := <-r.reqs:
				pending = append(pending, rq)
			case <-r.quit:
				return
			}
		case len(pending) == 0 || r.inUse == r.window || blocked:
			var tick <-chan time.Time
			if r.efd == nil { // no eventfd: look again shortly instead of being nudged
				tick = time.After(50 * time.Microsecond)
			}
			select {
			case rq := <-r.reqs:
				pending = append(pending, rq)
			case <-r.wake: // the library finished a batch
			case <-tick:
			case <-r.quit:
				r.drain(pending)
				return
			}
		}
		blocked = false
		// everything else that is pending right now comes along
	more:
		for len(pending) < r.window {
			select {
			case rq := <-r.reqs:
				pending = append(pending, rq)
			default:
				break more
			}
		}
		r.reap()
		// ---- submit: as many requests per cgo call as a call may hold and the arena ring has contiguous room for ----
		for len(pending) > 0 && r.inUse < r.window {
			n := min(len(pending), r.callMax, r.window-r.inUse, r.window-r.tail)
			if !r.submit(pending[:n]) {
				blocked = true
				break
			}
			pending = pending[n:]
		}
		r.reap()
	}
}

func (r *reactor) submit(reqs []*request) bool {
	n := len(reqs)
	for i, rq := range reqs {
		copy(r.scratch[i*r.inSize:], rq.in)
	}
	var ticket C.uint64_t
	var rc C.int
	ss := (*C.uint8_t)(unsafe.Add(r.arenaSS, r.tail*32))
	st := (*C.uint8_t)(unsafe.Add(r.arenaST, r.tail))
	if r.q != nil { // a call queue: the key rows travel with the call (public keys: nothing to clear)
		for i, rq := range reqs {
			copy(r.keys[i*r.keySize:], rq.key)
		}
		ct := (*C.uint8_t)(unsafe.Add(r.arenaCT, r.tail*r.ctSize))
		rc = C.circl_hip_queue_submit(r.q, ptr(r.keys[:n*r.keySize]), ptr(r.scratch[:n*r.inSize]), ct, ss, st, C.size_t(n), &ticket)
		clear(r.scratch[:n*r.inSize])
		return r.submitted(rc, ticket, reqs)
	}
	r.k.mu.RLock()
	if r.k.t == nil {
		r.k.mu.RUnlock()
		for _, rq := range reqs {
			rq.done <- reply{err: r.errQueueClosed}
		}
		return true
	}
	if r.decaps {
		rc = C.circl_hip_mlkem_decaps_table_submit(r.k.t, nil, ptr(r.scratch[:n*r.inSize]), ss, st, C.size_t(n), &ticket)
	} else {
		ct := (*C.uint8_t)(unsafe.Add(r.arenaCT, r.tail*r.ctSize))
		rc = C.circl_hip_mlkem_encaps_table_submit(r.k.t, nil, ptr(r.scratch[:n*r.inSize]), ct, ss, st, C.size_t(n), &ticket)
	}
	r.k.mu.RUnlock()
	clear(r.scratch[:n*r.inSize]) // (the library has its copy; encapsulation seeds are secret)
	return r.submitted(rc, ticket, reqs)
}

// submitted books a submit's outcome: false = CIRCL_HIP_EAGAIN (nothing was taken; a completion has to come first)
func (r *reactor) submitted(rc C.int, ticket C.uint64_t, reqs []*request) bool {
	n := len(reqs)
	if rc == C.CIRCL_HIP_EAGAIN {
		return false
	}
	if rc != 0 {
		err := status(rc, "submit")
		for _, rq := range reqs {
			rq.done <- reply{err: err}
		}
		return true
	}
	r.fifo = append(r.fifo, flight{ticket, r.tail, append([]*request(nil), reqs...)})
	r.tail = (r.tail + n) % r.window
	r.inUse += n
	return true
}

// reap hands out every finished call at the head of the FIFO (tickets of one queue finish in issue order).
func (r *reactor) reap() {
	for len(r.fifo) > 0 {
		f := &r.fifo[0]
		var state C.int8_t
		if r.q != nil {
			C.circl_hip_queue_poll(r.q, &f.ticket, 1, &state)
		} else {
			r.k.mu.RLock()
			if r.k.t != nil {
				C.circl_hip_poll(r.k.t, &f.ticket, 1, &state)
			} else {
				state = C.int8_t(C.CIRCL_HIP_EPARAM)
			}
			r.k.mu.RUnlock()
		}
		if state == 0 {
			return
		}
		for i, rq := range f.reqs {
			s := f.slot + i
			var rp reply
			switch {
			case state != 1:
				rp.err = status(C.int(state), "table batch")
			default:
				rp.err = itemErr(*(*byte)(unsafe.Add(r.arenaST, s)))
			}
			if rp.err == nil {
				rp.ss = C.GoBytes(unsafe.Add(r.arenaSS, s*32), 32)
				if !r.decaps {
					rp.ct = C.GoBytes(unsafe.Add(r.arenaCT, s*r.ctSize), C.int(r.ctSize))
				}
			}
			C.memset(unsafe.Add(r.arenaSS, s*32), 0, 32) // the arena's copy of the shared secret
			rq.done <- rp
		}
		r.inUse -= len(f.reqs)
		r.head = (r.head + len(f.reqs)) % r.window
		r.fifo = r.fifo[1:]
	}
}

func (r *reactor) drain(pending []*request) {
	for _, rq := range pending {
		rq.done <- reply{err: r.errQueueClosed}
	}
}

// stop ends the goroutines and the queue; requests still in flight are answered with an error (their owners also see quit).
// circl_hip_keytable_async_stop finishes every submitted call before it frees the queue, so the arena is not written afterwards.
func (r *reactor) stop() {
	close(r.quit)
	r.stopped.Wait()
	if r.q != nil {
		C.circl_hip_queue_close(r.q) // (the reactor goroutine has left: nobody is inside the queue; what was submitted is finished first)
		r.q = nil
	} else {
		r.k.mu.Lock()
		if r.k.t != nil {
			C.circl_hip_keytable_async_stop(r.k.t) // (finishes what was submitted: nothing writes the arena afterwards)
		}
		r.k.mu.Unlock()
	}
	if r.efd != nil {
		r.efd.Close() // the completions goroutine's Read fails and it leaves
	}
	for _, f := range r.fifo {
		r.drain(f.reqs)
	}
	if r.arenaCT != nil {
		C.free(r.arenaCT)
	}
	C.memset(r.arenaSS, 0, C.size_t(r.window*32))
	C.free(r.arenaSS)
	C.free(r.arenaST)
}
