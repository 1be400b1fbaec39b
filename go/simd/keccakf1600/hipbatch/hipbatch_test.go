//go:build cgo && hip

package hipbatch

// Batched Keccak-f[1600] against CIRCL's simd/keccakf1600 (f1600x.go:30-44, 77-91): the permutation of the zero state
// (f1600x_test.go:9-19) and random states through StateX4 / the scalar fallback.
//
//	go test -tags hip ./simd/keccakf1600/hipbatch/
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain on any box); tools/gocheck.py checks it statically.

import (
	"testing"

	"github.com/cloudflare/circl/simd/keccakf1600"
)

func TestPermuteBatchAgainstCIRCL(t *testing.T) {
	if !IsEnabled() {
		t.Skip("no HIP device")
	}
	const n = 1000 // a multiple of 4: CIRCL permutes four interleaved states at a time
	for _, turbo := range []bool{false, true} {
		states := make([]uint64, 25*n)
		for i := range states {
			states[i] = uint64(i)*0x9e3779b97f4a7c15 + 1
		}
		for i := 0; i < 25; i++ {
			states[i] = 0 // item 0: the zero state
		}
		want := append([]uint64(nil), states...)
		if err := PermuteBatch(states, turbo, 0); err != nil {
			t.Fatal(err)
		}
		for g := 0; g < n; g += 4 {
			var s keccakf1600.StateX4
			a := s.Initialize(turbo)
			for k := 0; k < 4; k++ {
				for w := 0; w < 25; w++ {
					a[4*w+k] = want[25*(g+k)+w]
				}
			}
			s.Permute()
			for k := 0; k < 4; k++ {
				for w := 0; w < 25; w++ {
					if a[4*w+k] != states[25*(g+k)+w] {
						t.Fatalf("state %d word %d (turbo %v)", g+k, w, turbo)
					}
				}
			}
		}
	}
}
