//go:build cgo && hip

package hipbatch

// Batched X25519 against CIRCL's dh/x25519 (key.go:33-50): public keys, shared secrets and the low-order-point verdict.
//
//	go test -tags hip ./dh/x25519/hipbatch/
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain on any box); tools/gocheck.py checks it statically.

import (
	"testing"

	"github.com/cloudflare/circl/dh/x25519"
)

func TestBatchAgainstCIRCL(t *testing.T) {
	const n = 300
	secret := make([]x25519.Key, n)
	peer := make([]x25519.Key, n)
	for i := range secret {
		for j := range secret[i] {
			secret[i][j] = byte(7*i + 13*j + 1)
			peer[i][j] = byte(11*i + 3*j + 5)
		}
	}
	peer[4] = x25519.Key{}    // the point of order 1 ...
	peer[9] = x25519.Key{1}   // ... and of order 4: Shared reports false (key.go:41-50)
	public := make([]x25519.Key, n)
	if err := KeyGenBatch(public, secret, 0); err != nil {
		t.Fatal(err)
	}
	shared := make([]x25519.Key, n)
	ok, err := SharedBatch(shared, secret, peer, 0)
	if err != nil {
		t.Fatal(err)
	}
	for i := 0; i < n; i++ {
		var pub, sh x25519.Key
		x25519.KeyGen(&pub, &secret[i])
		if pub != public[i] {
			t.Fatalf("public key %d differs", i)
		}
		want := x25519.Shared(&sh, &secret[i], &peer[i])
		if ok[i] != want || (want && sh != shared[i]) {
			t.Fatalf("shared secret %d: gpu %v, circl %v", i, ok[i], want)
		}
	}
	if ok[4] || ok[9] {
		t.Fatal("low-order points must be reported")
	}
}
