// keytable.h -- a parsed-key cache that lives ACROSS calls (circl_hip_*_keytable_new / _free, include/circl_hip.h).
//
// The reference parses a key once and keeps what every operation needs in the key object: kem/mlkem PublicKey / PrivateKey
// hold A^T and H(ek) (kem/mlkem/mlkem768/kyber.go:39-43, :247-263, pke/kyber/kyber768/internal/cpapke.go:19-25), sign/mldsa
// PublicKey holds A and tr (sign/mldsa/mldsa65/internal/dilithium.go:114-126).  The per-call key tables (circl_hip_*_keyed)
// rebuild that material on every call; this object is the same table, expanded once, resident on ONE device -- or, made with
// device = CIRCL_HIP_ALL_DEVICES, replicated on every device so that the host-buffer calls shard a batch over all of them.
#pragma once
#include "host_common.h"

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <functional>

struct circl_hip_keytable {
    uint32_t magic;        // kKeytableMagic while alive
    int family;            // 1 = ML-KEM, 2 = ML-DSA, 3 = hybrid KEM (ML-KEM table + X25519 rows)
    int param;
    int device;            // LOGICAL device the table lives on; CIRCL_HIP_ALL_DEVICES: a set of replicas, one per logical device
    int private_keys;      // ML-KEM / hybrid: rows are decapsulation keys; ML-DSA: prepared private keys
    size_t nkeys;
    size_t row;            // bytes per key row
    uint8_t *d_keys;       // nkeys rows (+ slack)
    uint8_t *d_table;      // expanded material (layout of the family's key-table workspace tail)
    size_t keys_bytes, table_bytes;
    // family 3: the lattice half is an ML-KEM table of its own; d_x holds the X25519 rows (public: pk_X[nkeys][32]; private:
    // sk_X[nkeys][32] then pk_X[nkeys][32])
    int scheme;
    circl_hip_keytable *inner;
    uint8_t *d_x;
    size_t x_bytes;
    // device == CIRCL_HIP_ALL_DEVICES: replica[d] is the same table built on logical device d (owned by this object)
    circl_hip_keytable **replica;
    int nreplica;
    // circl_hip_keytable_set_coalesce: small host-buffer calls through this table join cross-caller batches (host_common.h)
    circl::host::Coalescer *coalescer;
};
constexpr uint32_t kKeytableMagic = 0x4b544232u;  // "KTB2"

namespace circl {
namespace host {
// the table to use on logical device `dev`: a replica of a replicated table, the table itself when it lives there, else nullptr
inline const circl_hip_keytable *keytable_on(const circl_hip_keytable *t, int dev) {
    if (!t || t->magic != kKeytableMagic) return nullptr;
    if (t->device < 0) return (dev >= 0 && dev < t->nreplica) ? t->replica[dev] : nullptr;
    return t->device == dev ? t : nullptr;
}
// for the *_dev entry points (pointers live on the calling thread's current HIP device): a single-device table is taken as given
// (its device is the caller's contract), a replicated one resolves to the replica on the current HIP device
const circl_hip_keytable *keytable_here(const circl_hip_keytable *t);
// device >= 0: make(device, out).  CIRCL_HIP_ALL_DEVICES: one table per logical device under a parent object.
int keytable_replicate(int device, const std::function<int(int dev, circl_hip_keytable **one)> &make, circl_hip_keytable **out);
// the host-buffer form of a table call: items [lo, lo + cnt) on the table's device -- or, with a replicated table, the batch split
// into contiguous shards, one per device, each on that device's replica (no collective)
// A SMALL call through a replicated table goes to ONE replica, taken round-robin: splitting a handful of items over every device
// costs a host thread and a launch per device for no gain (and the contiguous split sent every one-item call to the last device).
constexpr size_t kSmallTableCall = 1024;
int next_replica(int nreplica);
template <class F> int table_shard(const circl_hip_keytable *t, size_t n, F one) {
    if (t->device >= 0) return one(t, size_t(0), n);
    if (n <= kSmallTableCall && t->nreplica > 0) return one(t->replica[next_replica(t->nreplica)], size_t(0), n);
    return shard(n, CIRCL_HIP_ALL_DEVICES, [&](int dev, size_t lo, size_t cnt) { return one(t->replica[dev], lo, cnt); });
}
}  // namespace host
}  // namespace circl
