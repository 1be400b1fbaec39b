//go:build cgo && hip

package hipbatch

// Batched SHAKE / KangarooTwelve against CIRCL's own xof package (xof/xof.go:44-66) on ragged messages.
//
//	go test -tags hip ./xof/hipbatch/
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain on any box); tools/gocheck.py checks it statically.

import (
	"bytes"
	"testing"

	"github.com/cloudflare/circl/xof"
)

func TestSumBatchAgainstCIRCL(t *testing.T) {
	const n = 500
	msgs := make([][]byte, n)
	seed := xof.SHAKE256.New()
	_, _ = seed.Write([]byte("xof/hipbatch"))
	for i := range msgs {
		msgs[i] = make([]byte, (i*37)%1000) // empty, shorter and longer than one block, block-aligned
		_, _ = seed.Read(msgs[i])
	}
	for _, id := range []xof.ID{xof.SHAKE128, xof.SHAKE256, xof.K12D10} {
		for _, outLen := range []int{32, 200} { // one block and more than one block of output
			got, err := SumBatch(id, msgs, outLen, 0)
			if err != nil {
				t.Fatal(err)
			}
			want := make([]byte, outLen)
			for i := range msgs {
				x := id.New()
				_, _ = x.Write(msgs[i])
				_, _ = x.Read(want)
				if !bytes.Equal(got[i], want) {
					t.Fatalf("xof %d, message %d (%d bytes), %d bytes of output", id, i, len(msgs[i]), outLen)
				}
			}
		}
	}
	if _, err := SumBatch(xof.BLAKE2XB, msgs, 32, 0); err == nil {
		t.Fatal("BLAKE2X has no GPU path: an error is expected")
	}
}
