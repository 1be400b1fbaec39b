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

#include <atomic>
#include <functional>

namespace circl {
namespace host {
// How many calls are inside a table right now.  Striped: a caller counts itself in the slot of its thread (a cache line of its own), so
// that the reactors of the asynchronous form -- millions of submits per second from a handful of threads -- do not pass ONE line around
// (profiles/r06_async.txt: four reactors on one counter paid 1.4 us per item where one paid 0.2); a setter sums the slots.
constexpr int kUseSlots = 16;
struct alignas(64) UseSlot {
    std::atomic<uint32_t> n{0};
};
struct UseCount {
    UseSlot slot[kUseSlots];
    uint32_t load() const {
        uint32_t s = 0;
        for (auto &x : slot) s += x.n.load();  // (seq_cst, like the callers' increments: see `frozen`)
        return s;
    }
};
inline int use_slot() {
    static std::atomic<unsigned> next{0};
    thread_local const int mine = (int)(next.fetch_add(1) % kUseSlots);
    return mine;
}
}  // namespace host
}  // namespace circl

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
    // Lifetime of the coalescer under concurrent callers: every host-buffer call through the table counts itself
    // in `users` while it is inside (TableUse) and skips the coalescer while `frozen` is set; a setter sets `frozen`, then requires
    // users == 0 and an idle coalescer (else CIRCL_HIP_EBUSY) before it frees anything.  Both seq_cst: either the caller sees the
    // freeze, or the setter sees the caller.
    std::atomic<bool> frozen{false};
    mutable circl::host::UseCount users;  // (cache lines of its own, behind the read-mostly fields above)
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
// a call's stay inside a table (see circl_hip_keytable::users)
struct TableUse {
    const circl_hip_keytable *t;  // nullptr: nothing to count (a part that is the table itself is counted once)
    int s;
    explicit TableUse(const circl_hip_keytable *tt) : t(tt), s(use_slot()) { if (t) t->users.slot[s].n.fetch_add(1); }
    ~TableUse() { if (t) t->users.slot[s].n.fetch_sub(1); }
    TableUse(const TableUse &) = delete;
    TableUse &operator=(const TableUse &) = delete;
};
// the coalescer a call inside `r` (TableUse held) may use: none while a setter has the table frozen
inline Coalescer *usable_coalescer(const circl_hip_keytable *r) { return r->frozen.load() ? nullptr : r->coalescer; }
template <class F> int table_shard(const circl_hip_keytable *t, size_t n, F one, size_t one_replica_max = kSmallTableCall) {
    TableUse top(t);
    if (t->device >= 0) return one(t, size_t(0), n);
    auto part = [&](const circl_hip_keytable *r, size_t lo, size_t cnt) { TableUse use(r); return one(r, lo, cnt); };
    if (n <= one_replica_max && t->nreplica > 0) return part(t->replica[next_replica(t->nreplica)], size_t(0), n);
    return shard(n, CIRCL_HIP_ALL_DEVICES, [&](int dev, size_t lo, size_t cnt) { return part(t->replica[dev], lo, cnt); });
}
// the asynchronous queues of the two families (api_mlkem.hip, api_mldsa.hip): fix the queue's arrays and launch on `co`, start its dispatcher
int kem_table_async_start(const circl_hip_keytable *r, Coalescer *co, bool want_eventfd);
int dsa_table_async_start(const circl_hip_keytable *r, Coalescer *co, bool want_eventfd);
int hyb_table_async_start(const circl_hip_keytable *r, Coalescer *co, bool want_eventfd);
// circl_hip_keytable_free: waits (bounded) until no call is inside the table; false = still busy
bool keytable_quiesce(circl_hip_keytable *t);
// a submitted call picks its part like a small blocking call does and says which in the ticket's top byte
inline uint64_t make_ticket(int replica, uint64_t seq) { return ((uint64_t)(unsigned)replica << 56) | seq; }
// One part of a table takes a submitted call: the table itself, or -- replicated -- one replica, round-robin; the ticket says which.
template <class F> int table_submit(const circl_hip_keytable *t, uint64_t *ticket, F one) {
    TableUse top(t);
    int rep = 0;
    const circl_hip_keytable *r = t;
    if (t->device < 0) {
        if (t->nreplica <= 0) return CIRCL_HIP_EPARAM;
        rep = next_replica(t->nreplica);
        r = t->replica[rep];
    }
    TableUse use(r != t ? r : nullptr);
    Coalescer *co = usable_coalescer(r);
    if (!co || !coalescer_is_async(co)) { g_err = "the table has no asynchronous queue (circl_hip_keytable_async_start)"; return CIRCL_HIP_EPARAM; }
    uint64_t seq = 0;
    const int rc = one(r, co, &seq);
    if (rc == CIRCL_HIP_OK) *ticket = make_ticket(rep, seq);
    return rc;
}
}  // namespace host
}  // namespace circl
