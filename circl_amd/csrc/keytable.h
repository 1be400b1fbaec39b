// keytable.h -- a parsed-key cache that lives ACROSS calls (circl_hip_*_keytable_new / _free, include/circl_hip.h).
//
// The reference parses a key once and keeps what every operation needs in the key object: kem/mlkem PublicKey / PrivateKey
// hold A^T and H(ek) (kem/mlkem/mlkem768/kyber.go:39-43, :247-263, pke/kyber/kyber768/internal/cpapke.go:19-25), sign/mldsa
// PublicKey holds A and tr (sign/mldsa/mldsa65/internal/dilithium.go:114-126).  The per-call key tables (circl_hip_*_keyed)
// rebuild that material on every call; this object is the same table, expanded once, resident on ONE device.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

struct circl_hip_keytable {
    uint32_t magic;        // kKeytableMagic while alive
    int family;            // 1 = ML-KEM, 2 = ML-DSA
    int param;
    int device;            // LOGICAL device the table lives on
    int private_keys;      // ML-KEM: rows are decapsulation keys
    size_t nkeys;
    size_t row;            // bytes per key row
    uint8_t *d_keys;       // nkeys rows (+ slack)
    uint8_t *d_table;      // expanded material (layout of the family's key-table workspace tail)
    size_t keys_bytes, table_bytes;
};
constexpr uint32_t kKeytableMagic = 0x4b544231u;  // "KTB1"
