/* circl_hip.h -- C ABI of libcirclhip.so, the MI355X (gfx950) batch ML-KEM / ML-DSA engine.
 *
 * This is the drop-in boundary for cloudflare/circl's lattice hot path.  The reference has no
 * FFI of its own (it is pure Go + Go assembler), so each entry point below cites the reference
 * Go interface it replaces; INTEGRATION.md shows the cgo binding a CIRCL maintainer would add
 * under kem/mlkem and sign/mldsa.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every buffer; nothing is retained after
 *     return (cgo pointer rule); no exceptions or aborts cross the ABI.
 *   - batches are contiguous row-major arrays: ek[n][EK], m[n][32], ct[n][CT], ss[n][32] ...
 *   - the ABI is deterministic: randomness (kem.Scheme.Encapsulate's 32 random bytes,
 *     kem/mlkem/mlkem768/kyber.go:104-108) stays on the Go side and arrives as `m` / seeds.
 *   - functions return 0 or a negative CIRCL_HIP_E* code; data-dependent, per-item failures
 *     (the reference's kem.ErrPubKey / kem.ErrPrivKey) are written to status[n] and the
 *     item's outputs are zero-filled.
 *   - `device` >= 0 selects one GPU; CIRCL_HIP_ALL_DEVICES (-1) splits the batch into
 *     contiguous shards, one per visible GPU, with no collective (items are independent).
 *     A SMALL call is not split: it goes to ONE device, taken round-robin (a host thread and a launch per device for a
 *     handful of items cost more than they return) -- up to 1024 items for ML-KEM and ML-DSA verification, up to 64 for the
 *     operations whose small batches are latency-bound for hundreds of microseconds (ML-DSA signing and key generation,
 *     X25519 and the hybrids), so that a few hundred of those already use every device.
 *   - the *_dev variants take DEVICE pointers and a hipStream_t (as void*), enqueue the
 *     kernels and return without synchronising: this is what bench.py times with inputs
 *     already resident in HBM.  All device pointers must be 16-byte aligned.
 *     The device the pointers live on must be the calling thread's CURRENT device (hipSetDevice); workspace sizes are
 *     valid for every visible device.  Rows that are not a multiple of 4 bytes (ML-DSA signatures, contexts and
 *     messages) are read as whole aligned dwords: device arrays must have at least 4 bytes of addressable slack
 *     behind their last row (the host-buffer entry points provide it themselves).
 *   - host-buffer entry points accept ANY alignment and ordinary pageable memory (a Go []byte sub-slice): every
 *     device owns a pool of staging slots (page-locked host staging + device staging + a stream each) that a small
 *     per-device thread pool, pinned to the GPU's NUMA node, fills and drains, so H2D, kernels and D2H of
 *     successive chunks overlap and a pageable caller reaches the PCIe-bound rate.  Page-locked caller buffers
 *     (circl_hip_alloc_host / hipHostRegister) are DMA-ed directly.  Entry points are re-entrant: concurrent
 *     callers take different slots.  Staging copies of secret inputs / outputs (seeds, private keys, m, shared
 *     secrets) are wiped before a slot is recycled.  Tuning aids: CIRCL_HIP_HOST_THREADS, CIRCL_HIP_HOST_CHUNK
 *     (log2 items per chunk), CIRCL_HIP_HOST_SLOTS.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point
 *     returns CIRCL_HIP_ENODEV.
 */
#ifndef CIRCL_HIP_H
#define CIRCL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CIRCL_HIP_OK 0
#define CIRCL_HIP_EPARAM (-1)   /* unknown parameter set / bad argument */
#define CIRCL_HIP_ENODEV (-2)   /* no HIP device, or device index out of range */
#define CIRCL_HIP_EHIP (-3)     /* a HIP runtime call failed (see circl_hip_last_error) */
#define CIRCL_HIP_ENOMEM (-4)
#define CIRCL_HIP_EWORKSPACE (-5) /* workspace too small / misaligned pointer */
#define CIRCL_HIP_EBUSY (-6)    /* the object has calls in flight (circl_hip_keytable_set_coalesce, _async_start / _stop, _close) */
#define CIRCL_HIP_EAGAIN (-7)   /* a *_submit call found every batch of the queue busy: poll / wait for a ticket, then submit again */

#define CIRCL_HIP_ALL_DEVICES (-1)

/* per-item status codes */
#define CIRCL_HIP_ITEM_OK 0
#define CIRCL_HIP_ITEM_ERR_PUBKEY 1  /* kem.ErrPubKey : pke/kyber/kyber768/internal/cpapke.go:45-55 */
#define CIRCL_HIP_ITEM_ERR_PRIVKEY 2 /* kem.ErrPrivKey: kem/mlkem/mlkem768/kyber.go:219-228 */

/* ---- library / device management ------------------------------------------------------ */
int circl_hip_init(void);               /* idempotent; returns the number of devices or <0 */
int circl_hip_device_count(void);
const char *circl_hip_last_error(void); /* thread-local message for the last CIRCL_HIP_EHIP */
const char *circl_hip_version(void);
/* compute units and NUMA node (-1 unknown) of one device; either pointer may be NULL */
int circl_hip_device_info(int device, int *cus, int *numa_node);
/* Device indices of this ABI are LOGICAL.  By default they are the process's HIP devices.  With the environment variable
 * CIRCL_HIP_LOGICAL_DEVICES=L (read once, at the first call; never fewer than the HIP devices) the library presents L
 * devices, logical device d running on HIP device d mod the HIP device count, each with its own staging pool, copy /
 * compute streams and byte movers: a one-GPU box then executes exactly the host-side code of an L-GPU node
 * (CIRCL_HIP_ALL_DEVICES splits into L contiguous shards driven by L threads), and an 8-GPU node can be driven as 16
 * half-shards.  Returns the HIP device behind a logical one, or CIRCL_HIP_ENODEV. */
int circl_hip_physical_device(int device);

/* ---- sizes (kem.Scheme.PublicKeySize etc., kem/kem.go:33-82; sign/sign.go:48-94) -------- */
size_t circl_hip_mlkem_ek_size(int param); /* 512|768|1024 -> 800|1184|1568, 0 if unknown */
size_t circl_hip_mlkem_dk_size(int param); /*                1632|2400|3168 */
size_t circl_hip_mlkem_ct_size(int param); /*                 768|1088|1568 */
size_t circl_hip_mldsa_pk_size(int param); /* 44|65|87 -> 1312|1952|2592 */
size_t circl_hip_mldsa_sig_size(int param);/*             2420|3309|4627 */
size_t circl_hip_mldsa_sk_size(int param); /*             2560|4032|4896 */

/* ---- ML-KEM, host buffers (what cgo binds) ---------------------------------------------
 * circl_hip_mlkem_encaps: scheme.UnmarshalBinaryPublicKey(ek_i) followed by
 *   scheme.EncapsulateDeterministically(pk_i, m_i) for every i
 *   (kem/mlkem/mlkem768/kyber.go:390-396, :359-370 -> :103-137 EncapsulateTo).
 *   status[i] = CIRCL_HIP_ITEM_ERR_PUBKEY when ek_i holds a coefficient >= q.
 * circl_hip_mlkem_decaps: scheme.UnmarshalBinaryPrivateKey(dk_i) followed by
 *   scheme.Decapsulate(sk_i, ct_i) (kyber.go:398-407 -> :209-230 ; :376-386 -> :144-184).
 *   status[i] = CIRCL_HIP_ITEM_ERR_PRIVKEY when H(ek) != the hash stored in dk_i.  An invalid
 *   ciphertext is NOT an error (implicit rejection).
 * circl_hip_mlkem_keygen: scheme.DeriveKeyPair(seed_i), seed = d || z, 64 bytes
 *   (kyber.go:340-345 -> :57-78 NewKeyFromSeed), keys in MarshalBinary form.
 * status may be NULL.
 */
int circl_hip_mlkem_encaps(int param, const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss,
                           uint8_t *status, size_t n, int device);
int circl_hip_mlkem_decaps(int param, const uint8_t *dk, const uint8_t *ct, uint8_t *ss,
                           uint8_t *status, size_t n, int device);
int circl_hip_mlkem_keygen(int param, const uint8_t *seed64, uint8_t *ek, uint8_t *dk, size_t n,
                           int device);

/* ---- ML-KEM, device-resident ------------------------------------------------------------
 * Same semantics; every pointer is a device pointer on the device `stream` belongs to.
 * `workspace` must hold circl_hip_mlkem_workspace_size(param, n) bytes.  status may NOT be NULL.
 * After a call the workspace still holds per-item intermediates that are as secret as the call's secret inputs (the
 * encryption coins r, the decrypted m'): a caller that hands the memory on should zero it (the host-buffer forms do).
 * circl_hip_mlkem_workspace_size is MONOTONE in n: 129 B per item + 32 KB of matrix scratch per resident workgroup (about
 * 134 MB) + a row cache of 8 KB per item for the first 2^15 items (the small-batch routes; at most 256 MB).  A workspace
 * sized once for the largest batch serves every smaller batch.  A call whose workspace lacks the row cache for its n (but
 * has the first two parts and 128 KB) still succeeds, through the big-batch routes; results are identical either way.
 */
size_t circl_hip_mlkem_workspace_size(int param, size_t n);
int circl_hip_mlkem_encaps_dev(int param, const uint8_t *d_ek, const uint8_t *d_m, uint8_t *d_ct,
                               uint8_t *d_ss, uint8_t *d_status, size_t n, void *d_workspace,
                               size_t workspace_bytes, void *stream);
int circl_hip_mlkem_decaps_dev(int param, const uint8_t *d_dk, const uint8_t *d_ct, uint8_t *d_ss,
                               uint8_t *d_status, size_t n, void *d_workspace,
                               size_t workspace_bytes, void *stream);
int circl_hip_mlkem_keygen_dev(int param, const uint8_t *d_seed64, uint8_t *d_ek, uint8_t *d_dk,
                               size_t n, void *d_workspace, size_t workspace_bytes, void *stream);

/* Shared-key encapsulation: all n items use the ONE encapsulation key at `ek` -- n times
 * scheme.EncapsulateDeterministically(pk, m_i) on one parsed key, the shape of the reference's BenchmarkEncapsulate
 * (kem/schemes/schemes_test.go:28-38), where the cached key amortises A^T and H(ek) (kyber.go:39-43).  Same
 * outputs as circl_hip_mlkem_encaps on n copies of the key; status[i] is the key's verdict for every i. */
int circl_hip_mlkem_encaps_shared(int param, const uint8_t *ek, const uint8_t *m, uint8_t *ct, uint8_t *ss,
                                  uint8_t *status, size_t n, int device);
int circl_hip_mlkem_encaps_shared_dev(int param, const uint8_t *d_ek, const uint8_t *d_m, uint8_t *d_ct,
                                      uint8_t *d_ss, uint8_t *d_status, size_t n, void *d_workspace,
                                      size_t workspace_bytes, void *stream);
/* Shared-key decapsulation: n ciphertexts for the ONE private key at `dk` (n times scheme.Decapsulate(sk, ct_i) on one
 * parsed key).  status[i] = 2 (kem.ErrPrivKey) for every i if the key fails its hash check. */
int circl_hip_mlkem_decaps_shared(int param, const uint8_t *dk, const uint8_t *ct, uint8_t *ss, uint8_t *status,
                                  size_t n, int device);
int circl_hip_mlkem_decaps_shared_dev(int param, const uint8_t *d_dk, const uint8_t *d_ct, uint8_t *d_ss,
                                      uint8_t *d_status, size_t n, void *d_workspace, size_t workspace_bytes,
                                      void *stream);

/* ---- ML-KEM key tables (grouped keys) -----------------------------------------------------------
 * A batch over a handful of distinct keys: item i uses row key_idx[i] of a table of nkeys keys.  This is the
 * reference's parsed-key object applied to a batch: kem.Scheme.UnmarshalBinaryPublicKey expands A^T and H(ek) once
 * per KEY (kem/mlkem/mlkem768/kyber.go:39-43, :247-263; pke/kyber/kyber768/internal/cpapke.go:19-25) and every later
 * EncapsulateDeterministically / Decapsulate on that object reuses them.  Here: matrix expansion, H(ek) and the
 * private key's hash check once per TABLE ENTRY, then the shared-key work per item.  Results are identical to
 * circl_hip_mlkem_encaps / _decaps on the gathered rows ek_table[key_idx[i]] / dk_table[key_idx[i]]; status[i] is the
 * verdict of item i's key.  Host entry points return CIRCL_HIP_EPARAM for an index >= nkeys; the _dev variants
 * bound d_key_idx (4-byte aligned) to the table -- an index >= nkeys uses the last entry -- and need
 * circl_hip_mlkem_keyed_workspace_size(param, n, nkeys) bytes. */
int circl_hip_mlkem_encaps_keyed(int param, const uint8_t *ek_table, size_t nkeys, const uint32_t *key_idx,
                                 const uint8_t *m, uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, int device);
int circl_hip_mlkem_decaps_keyed(int param, const uint8_t *dk_table, size_t nkeys, const uint32_t *key_idx,
                                 const uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, int device);
size_t circl_hip_mlkem_keyed_workspace_size(int param, size_t n, size_t nkeys);
int circl_hip_mlkem_encaps_keyed_dev(int param, const uint8_t *d_ek_table, size_t nkeys, const uint32_t *d_key_idx,
                                     const uint8_t *d_m, uint8_t *d_ct, uint8_t *d_ss, uint8_t *d_status, size_t n,
                                     void *d_workspace, size_t workspace_bytes, void *stream);
int circl_hip_mlkem_decaps_keyed_dev(int param, const uint8_t *d_dk_table, size_t nkeys, const uint32_t *d_key_idx,
                                     const uint8_t *d_ct, uint8_t *d_ss, uint8_t *d_status, size_t n,
                                     void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- key tables that LIVE ACROSS CALLS: the reference's parsed key objects ----------------------
 * kem.Scheme.UnmarshalBinaryPublicKey / UnmarshalBinaryPrivateKey return objects that keep A^T and H(ek) (kem/mlkem/mlkem768/
 * kyber.go:39-43, :247-263; the private key's hash check :219-228 happens there); sign.Scheme.UnmarshalBinaryPublicKey keeps A
 * and tr (sign/mldsa/mldsa65/internal/dilithium.go:114-126).  A circl_hip_keytable is that cache for nkeys keys, built ONCE on
 * one device: the key bytes, the expanded matrices and the hashes stay resident, and a call moves only its per-item data.
 * Item i of a call uses entry key_idx[i]; key_idx == NULL: every item uses entry 0 (a table of one key = one key object).
 * Results are identical to the per-call key-table entry points above.  A Go bridge keeps one table per key object (or per
 * key set) and frees it from the object's finalizer; a table is immutable and may be used by concurrent calls.
 *   circl_hip_mlkem_keytable_new : private_keys = 0: rows are encapsulation keys; 1: decapsulation keys, whose stored-hash
 *       verdicts (0 | 2 = kem.ErrPrivKey) are written to key_status[nkeys] if it is not NULL.  A public key's canonicity is
 *       reported per item by the encapsulation (status 1 = kem.ErrPubKey), as everywhere in this ABI.
 *   _dev variants: pointers are device memory on the TABLE's device; workspace = circl_hip_mlkem_workspace_size(param, n) /
 *       circl_hip_mldsa_workspace_size(param, n) bytes (no table tail: the table brings its own).  Like every keyed _dev entry point
 *       they cannot REPORT a bad d_key_idx (the host forms check it: an index >= nkeys is CIRCL_HIP_EPARAM; a device array cannot be
 *       checked without a synchronisation), so the kernels BOUND it: an index >= nkeys uses the table's LAST entry -- never memory
 *       behind the table.
 *   circl_hip_keytable_free wipes the key rows of a private table before releasing them.
 *   device = CIRCL_HIP_ALL_DEVICES in any *_new below REPLICATES the table: it is built once on every device, and the host-buffer
 *       *_table calls then split a batch into contiguous shards, one per device, like every other entry point (a table made for
 *       one device runs its calls there).  The _dev variants use the replica on the calling thread's current device;
 *       circl_hip_keytable_on_device returns that replica (or the table itself if it lives on `device`, else NULL) and
 *       circl_hip_keytable_device the table's device (CIRCL_HIP_ALL_DEVICES for a replicated one; CIRCL_HIP_ENODEV for NULL or a
 *       freed table), circl_hip_keytable_nkeys its number of entries (0 likewise). */
typedef struct circl_hip_keytable circl_hip_keytable;
int circl_hip_keytable_device(const circl_hip_keytable *table);
size_t circl_hip_keytable_nkeys(const circl_hip_keytable *table);
const circl_hip_keytable *circl_hip_keytable_on_device(const circl_hip_keytable *table, int device);
int circl_hip_mlkem_keytable_new(int param, int private_keys, const uint8_t *keys, size_t nkeys, int device,
                                 uint8_t *key_status, circl_hip_keytable **out);
int circl_hip_mldsa_keytable_new(int param, const uint8_t *pks, size_t nkeys, int device, circl_hip_keytable **out);
/* ONE ML-DSA private key prepared once -- A and the NTT-domain s1, s2, t0 of the reference's parsed PrivateKey (sign/mldsa/mldsa65/
 * internal/dilithium.go:149-179) -- then any number of scheme.Sign calls with it: circl_hip_mldsa_sign_table[_dev] = circl_hip_mldsa_
 * sign_shared[_dev] without the per-call ExpandA and transforms (workspace: circl_hip_mldsa_sign_workspace_size(param, n)). */
int circl_hip_mldsa_privkey_new(int param, const uint8_t *sk, int device, circl_hip_keytable **out);
/* ... and a table of nkeys prepared private keys (a signer that holds several identities): message i is signed with entry
 * key_idx[i]; key_idx == NULL: entry 0.  Same signatures as circl_hip_mldsa_sign on the gathered rows sks[key_idx[i]].  The host
 * form returns CIRCL_HIP_EPARAM for an index >= nkeys; the _dev form bounds d_key_idx (4-byte aligned) to the table (>= nkeys: the last entry). */
int circl_hip_mldsa_privkeys_new(int param, const uint8_t *sks, size_t nkeys, int device, circl_hip_keytable **out);
int circl_hip_mldsa_sign_table_keyed(const circl_hip_keytable *table, const uint32_t *key_idx, const uint8_t *msg_blob, const uint64_t *msg_off,
                                     const uint8_t *ctx_blob, const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sig, size_t n);
int circl_hip_mldsa_sign_table_keyed_dev(const circl_hip_keytable *table, const uint32_t *d_key_idx, const uint8_t *d_msg_blob,
                                         const uint64_t *d_msg_off, const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off, const uint8_t *d_rnd,
                                         int internal, uint8_t *d_sig, size_t n, void *d_workspace, size_t workspace_bytes, void *stream);
int circl_hip_mldsa_sign_table(const circl_hip_keytable *table, const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob,
                               const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sig, size_t n);
int circl_hip_mldsa_sign_table_dev(const circl_hip_keytable *table, const uint8_t *d_msg_blob, const uint64_t *d_msg_off,
                                   const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off, const uint8_t *d_rnd, int internal, uint8_t *d_sig,
                                   size_t n, void *d_workspace, size_t workspace_bytes, void *stream);
void circl_hip_keytable_free(circl_hip_keytable *table);
int circl_hip_mlkem_encaps_table(const circl_hip_keytable *table, const uint32_t *key_idx, const uint8_t *m, uint8_t *ct,
                                 uint8_t *ss, uint8_t *status, size_t n);
int circl_hip_mlkem_decaps_table(const circl_hip_keytable *table, const uint32_t *key_idx, const uint8_t *ct, uint8_t *ss,
                                 uint8_t *status, size_t n);
int circl_hip_mldsa_verify_table(const circl_hip_keytable *table, const uint32_t *key_idx, const uint8_t *sig,
                                 const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob,
                                 const uint64_t *ctx_off, uint8_t *ok, size_t n);
int circl_hip_mlkem_encaps_table_dev(const circl_hip_keytable *table, const uint32_t *d_key_idx, const uint8_t *d_m,
                                     uint8_t *d_ct, uint8_t *d_ss, uint8_t *d_status, size_t n, void *d_workspace,
                                     size_t workspace_bytes, void *stream);
int circl_hip_mlkem_decaps_table_dev(const circl_hip_keytable *table, const uint32_t *d_key_idx, const uint8_t *d_ct,
                                     uint8_t *d_ss, uint8_t *d_status, size_t n, void *d_workspace, size_t workspace_bytes,
                                     void *stream);
int circl_hip_mldsa_verify_table_dev(const circl_hip_keytable *table, const uint32_t *d_key_idx, const uint8_t *d_sig,
                                     const uint8_t *d_msg_blob, const uint64_t *d_msg_off, const uint8_t *d_ctx_blob,
                                     const uint64_t *d_ctx_off, uint8_t *d_ok, size_t n, void *d_workspace,
                                     size_t workspace_bytes, void *stream);

/* ---- many concurrent ONE-ITEM callers: cross-caller coalescing (opt-in, per table) ------------------------------------------
 * The reference's consumers do not batch: kem/hybrid (kem/hybrid/hybrid.go:95-99), kem/xwing (kem/xwing/xwing.go:259,288), hpke
 * (hpke/algs.go:283-285) and every user of kem.Scheme / sign.Scheme call Encapsulate / Decapsulate / Verify with one key and one
 * item (kem/mlkem/mlkem768/kyber.go:347-386, sign/mldsa/mldsa65/dilithium.go:305) from whichever goroutine owns the connection.
 * One such call through a resident table costs a launch and a wait (tens of microseconds) however little the kernel does.
 * circl_hip_keytable_set_coalesce(table, max_items, max_wait_us) lets the small host-buffer *_table calls of CONCURRENT callers
 * (calls of at most max_items / 4 items) share launches: callers copy their rows into an open batch, the first of them flushes it
 * as soon as the device has room for another batch -- so a batch holds exactly the calls that arrived while the previous ones ran,
 * nothing is delayed when the table is idle -- or after max_wait_us microseconds if that is not 0; each caller returns with its own
 * rows.  Results are byte for byte those of the same calls made one by one (same kernels; items are independent).
 *   max_items: largest batch (2 .. 8192; 0 switches coalescing off).  CIRCL_HIP_EBUSY (nothing changed) while calls are in flight
 *   through the table: set it before the table is shared, or retry.  A replicated table coalesces per replica; its small calls (<= 1024
 *   items) go to one replica each, round-robin.  circl_hip_keytable_coalesce_stats: calls and items that joined batches and the
 *   launches they became (items / launches = mean batch).  Served today: circl_hip_mlkem_encaps_table, circl_hip_mlkem_decaps_table,
 *   circl_hip_mldsa_verify_table, circl_hip_mldsa_sign_table / _sign_table_keyed (a server signing one handshake transcript per call),
 *   circl_hip_hybrid_encaps_table / _decaps_table. */
int circl_hip_keytable_set_coalesce(circl_hip_keytable *table, size_t max_items, uint32_t max_wait_us);
/* Lifetime under concurrent callers: circl_hip_keytable_set_coalesce, circl_hip_keytable_async_start / _stop and circl_hip_keytable_close
 * return CIRCL_HIP_EBUSY -- and change nothing -- while a call is inside the table or one of its batches is open or running; nothing is
 * ever freed under a caller.  circl_hip_keytable_close(table) = circl_hip_keytable_free with that verdict (CIRCL_HIP_OK: freed);
 * circl_hip_keytable_free itself waits up to two seconds for the calls inside to return and, should they not, marks the table dead
 * (later calls get CIRCL_HIP_EPARAM) and LEAKS its memory instead of freeing it under them.
 *
 * ---- the ASYNCHRONOUS form: submit / poll, no sleeping thread per call ---------------------------------------------------------
 * The blocking form above holds one OS thread per outstanding call (a goroutine inside a blocking cgo call holds an M: ten thousand
 * concurrent handshakes, hpke/algs.go:283-285 or kem/hybrid/hybrid.go:95-99, would be ten thousand threads), and the sleep + wake of
 * each of them is most of a coalesced call's host cost (profiles/r05_concurrent_final.txt: 7.6 of 13 us).  With
 *   circl_hip_keytable_async_start(table, max_items, max_wait_us, want_eventfd)
 * the table gets a QUEUE served by one library thread per device (the dispatcher): it closes the open batch whenever the device has
 * room -- same group commit as above: a batch holds what arrived while its predecessors ran, nothing waits on an idle device unless
 * max_wait_us says so -- launches it, copies the finished batch's rows STRAIGHT INTO THE SUBMITTERS' OUTPUT BUFFERS, and publishes the
 * batch's completion.  Served: circl_hip_mlkem_encaps_table_submit (public ML-KEM table), circl_hip_mlkem_decaps_table_submit (private
 * ML-KEM table), circl_hip_mldsa_verify_table_submit (ML-DSA public-key table), circl_hip_hybrid_encaps_table_submit / _decaps_table_submit
 * (X-Wing / X25519MLKEM768 / Kyber-X25519 tables: kem/hybrid/hybrid.go:95-99, kem/xwing/xwing.go:259,288 -- one hybrid launch costs an
 * X25519 ladder, ~0.8 ms, whatever it holds: the calls that gain most from sharing it).
 *   *_submit(...same arrays as the blocking call..., n, &ticket): copies the n <= max_items / 4 items' inputs into the open batch, notes
 *       the output pointers and returns AT ONCE: CIRCL_HIP_OK and a ticket; CIRCL_HIP_EAGAIN when every batch of the queue is busy
 *       (nothing was taken: poll, then submit again); CIRCL_HIP_EPARAM for what the blocking call rejects (a NULL required input, a
 *       key index >= nkeys, an unsupported context) or a table without a queue.  The output buffers (and nothing else: inputs are
 *       copied) must stay valid and untouched until the ticket is done.  Results are byte for byte those of the blocking calls; per-
 *       item verdicts (status / ok rows, a bad decapsulation-key entry, a non-canonical public key) arrive in the rows as always.
 *   circl_hip_poll(table, tickets, n, state): state[i] = 1 done, 0 pending, < 0 the ticket's batch failed (a CIRCL_HIP_E* code; its
 *       output rows were zeroed: no shared secret, ok = 0) or the ticket is not one of this table's (CIRCL_HIP_EPARAM); returns how
 *       many are not pending.  One atomic load per ticket, no lock, no system call.  Tickets of one queue complete in the order
 *       they were issued, so a host that keeps its outstanding tickets in a FIFO only ever needs to look at the head.
 *   circl_hip_wait(table, ticket, timeout_us): blocks the calling thread until the ticket is done (returns its state) or timeout_us
 *       microseconds passed (returns 0); timeout_us < 0: no limit.  ONE thread per device is all a host needs to block.
 *   circl_hip_keytable_eventfd(table, replica): with want_eventfd != 0 the queue owns an eventfd(2) (non-blocking, counter semantics)
 *       that the dispatcher adds 1 to after every finished batch -- put it into the host's own epoll / kqueue set and call
 *       circl_hip_poll when it becomes readable; -1 if the table has none.  replica: 0 for a one-device table, the logical device
 *       for a replicated one (one queue and one dispatcher per replica; a submitted call goes to one replica, round-robin, and its
 *       ticket says which: the top byte).
 *   circl_hip_keytable_async_stop(table): flushes and finishes every submitted call, stops the dispatchers, frees the queues
 *       (CIRCL_HIP_EBUSY while a call is inside the table: a *_submit, a circl_hip_poll, a thread blocked in circl_hip_wait); circl_hip_keytable_free does the same on the way.
 * The blocking *_table calls keep working on such a table (they submit and wait), so one table can serve both kinds of caller. */
int circl_hip_keytable_close(circl_hip_keytable *table);
int circl_hip_keytable_async_start(circl_hip_keytable *table, size_t max_items, uint32_t max_wait_us, int want_eventfd);
int circl_hip_keytable_async_stop(circl_hip_keytable *table);
int circl_hip_keytable_eventfd(const circl_hip_keytable *table, int replica);
int circl_hip_mlkem_encaps_table_submit(const circl_hip_keytable *table, const uint32_t *key_idx, const uint8_t *m, uint8_t *ct, uint8_t *ss,
                                        uint8_t *status, size_t n, uint64_t *ticket);
int circl_hip_mlkem_decaps_table_submit(const circl_hip_keytable *table, const uint32_t *key_idx, const uint8_t *ct, uint8_t *ss, uint8_t *status,
                                        size_t n, uint64_t *ticket);
int circl_hip_mldsa_verify_table_submit(const circl_hip_keytable *table, const uint32_t *key_idx, const uint8_t *sig, const uint8_t *msg_blob,
                                        const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off, uint8_t *ok, size_t n,
                                        uint64_t *ticket);
int circl_hip_hybrid_encaps_table_submit(const circl_hip_keytable *table, const uint32_t *key_idx, const uint8_t *eseed, uint8_t *ct, uint8_t *ss,
                                         uint8_t *status, size_t n, uint64_t *ticket);
int circl_hip_hybrid_decaps_table_submit(const circl_hip_keytable *table, const uint32_t *key_idx, const uint8_t *ct, uint8_t *ss, uint8_t *status,
                                         size_t n, uint64_t *ticket);
int circl_hip_poll(const circl_hip_keytable *table, const uint64_t *tickets, size_t n, int8_t *state);
int circl_hip_wait(const circl_hip_keytable *table, uint64_t ticket, int64_t timeout_us);
/* ---- the asynchronous form for keys that come WITH the call: circl_hip_queue ------------------------------------------------------
 * A TLS 1.3 server encapsulates once per handshake, to the CLIENT'S ephemeral key (kem/hybrid/hybrid.go:271-300 ->
 * kem/mlkem/mlkem768/kyber.go:359-370): there is no resident table to hang a queue on.  A circl_hip_queue is that queue by itself: one
 * operation, one parameter set, one device, the same dispatcher thread, tickets, poll / wait / eventfd as above.
 *   circl_hip_queue_open(op, param, device, max_items, want_eventfd, &q): op = CIRCL_HIP_QUEUE_MLKEM_ENCAPS / _MLKEM_DECAPS (param 512 / 768 /
 *       1024) or CIRCL_HIP_QUEUE_HYBRID_ENCAPS / _HYBRID_DECAPS (param = the CIRCL_HIP_HYBRID_* scheme); device >= 0.
 *   circl_hip_queue_submit(q, key, in, out0, ss, status, n, &ticket): n <= max_items / 4 items, each with its OWN key row.
 *       encapsulation: key = n encapsulation (public) keys, in = n seeds m (ML-KEM: 32 bytes; hybrids: their eseed), out0 = n ciphertexts;
 *       decapsulation: key = n decapsulation (private) keys, in = n ciphertexts, out0 unused (NULL).
 *       ss = n shared secrets, status = n per-item status bytes (may be NULL).  key and in are copied before the call returns; out0 / ss /
 *       status must stay valid until the ticket is done.  Returns as the *_table_submit calls do (CIRCL_HIP_EAGAIN: poll, submit again).
 *   circl_hip_queue_poll / _wait / _eventfd: as circl_hip_poll / circl_hip_wait / circl_hip_keytable_eventfd.
 *   circl_hip_queue_close: CIRCL_HIP_EBUSY while a call is inside the queue; otherwise finishes every submitted call, then frees.
 * Bytes are those of circl_hip_mlkem_encaps / _decaps / circl_hip_hybrid_encaps / _decaps on the same rows. */
#define CIRCL_HIP_QUEUE_MLKEM_ENCAPS 1
#define CIRCL_HIP_QUEUE_MLKEM_DECAPS 2
#define CIRCL_HIP_QUEUE_HYBRID_ENCAPS 3
#define CIRCL_HIP_QUEUE_HYBRID_DECAPS 4
typedef struct circl_hip_queue circl_hip_queue;
int circl_hip_queue_open(int op, int param, int device, size_t max_items, int want_eventfd, circl_hip_queue **out);
int circl_hip_queue_close(circl_hip_queue *q);
int circl_hip_queue_eventfd(const circl_hip_queue *q);
int circl_hip_queue_stats(const circl_hip_queue *q, uint64_t *calls, uint64_t *items, uint64_t *launches); /* as circl_hip_keytable_coalesce_stats */
int circl_hip_queue_submit(circl_hip_queue *q, const uint8_t *key, const uint8_t *in, uint8_t *out0, uint8_t *ss, uint8_t *status, size_t n,
                           uint64_t *ticket);
int circl_hip_queue_poll(const circl_hip_queue *q, const uint64_t *tickets, size_t n, int8_t *state);
int circl_hip_queue_wait(const circl_hip_queue *q, uint64_t ticket, int64_t timeout_us);
/* The same for the entry points that take their keys WITH the call -- a TLS 1.3 server encapsulates once per handshake, to the
 * client's ephemeral key share (kem/hybrid/hybrid.go:271-300 -> kem/mlkem/mlkem768/kyber.go:359-370): nothing resident to attach a
 * batch to.  circl_hip_set_coalesce(max_items, max_wait_us), process-wide, default off: small calls (<= max_items / 4 items) of
 * circl_hip_mlkem_encaps, circl_hip_mlkem_decaps, circl_hip_mldsa_verify / _verify_internal, circl_hip_hybrid_encaps and
 * circl_hip_hybrid_decaps from concurrent threads share launches per (entry point, parameter set, device); every item still brings
 * its own key, the bytes are those of the un-coalesced calls.  Call it once, before the threads start (later calls only affect
 * (entry point, parameter set, device) combinations that have not been used yet).  circl_hip_set_coalesce(0, 0) switches new joins off
 * AND DRAINS: it returns once no call is inside any process-wide batch (CIRCL_HIP_EBUSY if that takes more than five seconds). */
int circl_hip_set_coalesce(size_t max_items, uint32_t max_wait_us);
int circl_hip_keytable_coalesce_stats(const circl_hip_keytable *table, uint64_t *calls, uint64_t *items, uint64_t *launches);

/* ---- PrivateKey.Public() over a batch -----------------------------------------------------------
 * circl_hip_mlkem_public_from_private: kem/mlkem/mlkem768/kyber.go:323-328 -- the encapsulation key stored inside each
 *   decapsulation key (a strided copy on the host, no device work).
 * circl_hip_mldsa_public_from_private: sign/mldsa/mldsa65/internal/dilithium.go:473-484 -- t = A s1 + s2 recomputed from the
 *   packed private key (ExpandA, NTT, Power2Round), pk = rho || PackT1(t1).  The _dev form needs
 *   circl_hip_mldsa_workspace_size(param, n) bytes. */
int circl_hip_mlkem_public_from_private(int param, const uint8_t *dk, uint8_t *ek, size_t n);
int circl_hip_mldsa_public_from_private(int param, const uint8_t *sk, uint8_t *pk, size_t n, int device);
int circl_hip_mldsa_public_from_private_dev(int param, const uint8_t *d_sk, uint8_t *d_pk, size_t n, void *d_workspace, size_t workspace_bytes,
                                            void *stream);

/* ---- round-3 Kyber ------------------------------------------------------
 * kem/kyber/kyber{512,768,1024}: the pre-standard KEM the reference still ships ("Kyber512/768/1024" in
 * kem/schemes).  param 512 | 768 | 1024; key and ciphertext sizes are those of ML-KEM.  Differences from
 * ML-KEM (kem/kyber/kyber768/kyber.go):
 *   keygen  scheme.DeriveKeyPair(seed64): G(d) without the domain byte                          :60-82
 *   encaps  scheme.EncapsulateDeterministically(pk, seed32): m = H(seed); (K',r) = G(m || H(pk));
 *           K = KDF(K' || H(ct)); key coefficients >= q are reduced, never rejected   :105-154, :248-262
 *   decaps  scheme.Decapsulate(sk, ct): K = KDF((ct' == ct ? K'' : z) || H(ct)); the private key is
 *           not checked                                                               :156-197, :215-232
 * so none of the three has a per-item failure.  Workspace of the _dev variants:
 * circl_hip_mlkem_workspace_size(param, n). */
int circl_hip_kyber_keygen(int param, const uint8_t *seed64, uint8_t *ek, uint8_t *dk, size_t n,
                           int device);
int circl_hip_kyber_encaps(int param, const uint8_t *ek, const uint8_t *seed32, uint8_t *ct,
                           uint8_t *ss, size_t n, int device);
int circl_hip_kyber_decaps(int param, const uint8_t *dk, const uint8_t *ct, uint8_t *ss, size_t n,
                           int device);
int circl_hip_kyber_keygen_dev(int param, const uint8_t *d_seed64, uint8_t *d_ek, uint8_t *d_dk,
                               size_t n, void *d_workspace, size_t workspace_bytes, void *stream);
int circl_hip_kyber_encaps_dev(int param, const uint8_t *d_ek, const uint8_t *d_seed32, uint8_t *d_ct,
                               uint8_t *d_ss, size_t n, void *d_workspace, size_t workspace_bytes,
                               void *stream);
int circl_hip_kyber_decaps_dev(int param, const uint8_t *d_dk, const uint8_t *d_ct, uint8_t *d_ss,
                               size_t n, void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- ML-DSA (param 44 | 65 | 87) and round-3 Dilithium2/3/5 (param 2 | 3 | 5) ----
 * Every circl_hip_mldsa_* entry point below also accepts param 2, 3 or 5 = sign/dilithium/mode{2,3,5}: the same
 * lattice, 32-byte tr and c~ (sign/dilithium/mode3/internal/params.go:5-18), key seed hashed without the K, L domain
 * bytes (internal/dilithium.go:191-193), mu = CRH(tr || msg) without a context prefix (mode3/dilithium.go:54-75) and
 * deterministic signing (no rnd, internal/dilithium.go:360-362).  For these modes contexts must be empty / NULL
 * (the reference fails with sign.ErrContextNotSupported) and rnd is ignored: the host-buffer entry points return
 * CIRCL_HIP_EPARAM when any context of the batch is non-empty; the device-resident ones (which cannot inspect the
 * offsets without synchronising) give ok[i] = 0 / an all-zero signature for such an item.
 *
 * ---- ML-DSA verify ----------------------------------------------------------------------
 * scheme.UnmarshalBinaryPublicKey(pk_i) + scheme.Verify(pk_i, msg_i, sig_i, &SignatureOpts{Context: ctx_i})
 * (sign/mldsa/mldsa65/dilithium.go:305-327 -> :115-132 -> internal/dilithium.go:273-332).
 * Messages / contexts are blobs with n+1 offsets; ctx_blob may be NULL (all contexts empty).
 * ok[i] = 1 valid, 0 invalid (bad encoding, norm bound, hint, hash mismatch or ctx > 255 bytes).
 */
int circl_hip_mldsa_verify(int param, const uint8_t *pk, const uint8_t *sig, const uint8_t *msg_blob,
                           const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off,
                           uint8_t *ok, size_t n, int device);
/* ML-DSA.Verify_internal (message hashed without the context prefix): the reference's
 * unsafeVerifyInternal, sign/mldsa/mldsa65/dilithium.go:101-109, which its ACVP tests drive. */
int circl_hip_mldsa_verify_internal(int param, const uint8_t *pk, const uint8_t *sig,
                                    const uint8_t *msg_blob, const uint64_t *msg_off, uint8_t *ok,
                                    size_t n, int device);
size_t circl_hip_mldsa_workspace_size(int param, size_t n);
int circl_hip_mldsa_verify_dev(int param, const uint8_t *d_pk, const uint8_t *d_sig,
                               const uint8_t *d_msg_blob, const uint64_t *d_msg_off,
                               const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off, uint8_t *d_ok,
                               size_t n, void *d_workspace, size_t workspace_bytes, void *stream);

/* Shared-key verification: all n signatures are checked under the ONE public key at `pk` -- n times
 * scheme.Verify(pk, msg_i, sig_i, opts_i) on one parsed key, where the reference caches A and tr in the PublicKey
 * object (sign/mldsa/mldsa65/internal/dilithium.go:114-126).  Same results as circl_hip_mldsa_verify on n copies of pk. */
int circl_hip_mldsa_verify_shared(int param, const uint8_t *pk, const uint8_t *sig, const uint8_t *msg_blob,
                                  const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off,
                                  uint8_t *ok, size_t n, int device);
int circl_hip_mldsa_verify_shared_dev(int param, const uint8_t *d_pk, const uint8_t *d_sig,
                                      const uint8_t *d_msg_blob, const uint64_t *d_msg_off,
                                      const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off, uint8_t *d_ok,
                                      size_t n, void *d_workspace, size_t workspace_bytes, void *stream);

/* Key-table verification: signature i is checked under row key_idx[i] of a table of nkeys public keys -- a verifier that
 * sees a few CA keys across a large batch.  tr and ExpandA (what PublicKey.Unpack caches per key,
 * sign/mldsa/mldsa65/internal/dilithium.go:114-126; benchmark note internal/dilithium_test.go:37-39) run once per TABLE
 * ENTRY.  Same results as circl_hip_mldsa_verify on the gathered rows.  The _dev variant needs
 * circl_hip_mldsa_keyed_workspace_size(param, n, nkeys) bytes and bounds d_key_idx to the table (>= nkeys: the last entry). */
int circl_hip_mldsa_verify_keyed(int param, const uint8_t *pk_table, size_t nkeys, const uint32_t *key_idx,
                                 const uint8_t *sig, const uint8_t *msg_blob, const uint64_t *msg_off,
                                 const uint8_t *ctx_blob, const uint64_t *ctx_off, uint8_t *ok, size_t n, int device);
size_t circl_hip_mldsa_keyed_workspace_size(int param, size_t n, size_t nkeys);
int circl_hip_mldsa_verify_keyed_dev(int param, const uint8_t *d_pk_table, size_t nkeys, const uint32_t *d_key_idx,
                                     const uint8_t *d_sig, const uint8_t *d_msg_blob, const uint64_t *d_msg_off,
                                     const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off, uint8_t *d_ok, size_t n,
                                     void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- ML-DSA key generation ----------------------------------------------------------------
 * scheme.DeriveKey(seed_i), seed 32 bytes (sign/mldsa/mldsa65/dilithium.go:272-281 ->
 * internal/dilithium.go:181-267 NewKeyFromSeed); keys in MarshalBinary form.  The _dev variant
 * needs circl_hip_mldsa_workspace_size(param, n) bytes of workspace. */
int circl_hip_mldsa_keygen(int param, const uint8_t *seed32, uint8_t *pk, uint8_t *sk, size_t n,
                           int device);
int circl_hip_mldsa_keygen_dev(int param, const uint8_t *d_seed32, uint8_t *d_pk, uint8_t *d_sk,
                               size_t n, void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- ML-DSA signing ----------------------------------------------------
 * scheme.UnmarshalBinaryPrivateKey(sk_i) + scheme.Sign(sk_i, msg_i, &SignatureOpts{Context: ctx_i})
 * (sign/mldsa/mldsa65/dilithium.go:283-303 -> :56-88 SignTo -> internal/dilithium.go:340-470).
 * rnd[n][32] is the per-signature randomness the Go wrapper draws from crypto/rand when
 * `randomized` is set; rnd == NULL means deterministic signing (32 zero bytes).  A context longer
 * than 255 bytes makes the host-buffer call return CIRCL_HIP_EPARAM (sign.ErrContextTooLong).
 * circl_hip_mldsa_sign_internal is ML-DSA.Sign_internal (unsafeSignInternal, dilithium.go:90-99).
 * The _dev variant needs circl_hip_mldsa_sign_workspace_size(param, n) bytes (about 60 KB per item for
 * ML-DSA-65: expanded matrix and NTT-domain secrets per item); d_rnd must not be NULL.  An item whose context is
 * longer than 255 bytes gets an ALL-ZERO signature from the _dev variants (never one over a truncated length).
 * The workspace holds key-equivalent intermediates while the call runs; its secret regions are zeroed before the
 * call's work completes.  Like every other _dev call the signer is ASYNCHRONOUS: the rejection loop runs as rounds
 * over the list of unsigned items whose length lives in device memory; the host enqueues a fixed schedule of rounds
 * (sized so that the chance of an item surviving it is below 2^-40) followed by a persistent kernel that finishes
 * whatever is left, and never reads anything back, so nothing synchronises `stream`. */
int circl_hip_mldsa_sign(int param, const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off,
                         const uint8_t *ctx_blob, const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sig,
                         size_t n, int device);
int circl_hip_mldsa_sign_internal(int param, const uint8_t *sk, const uint8_t *msg_blob,
                                  const uint64_t *msg_off, const uint8_t *rnd, uint8_t *sig, size_t n,
                                  int device);
/* Shared-key signing: all n messages are signed with the ONE private key at `sk` -- n times scheme.Sign(sk, msg_i, opts_i)
 * on one parsed key, where the reference caches A and the NTT-domain secrets in the PrivateKey
 * (sign/mldsa/mldsa65/internal/dilithium.go:149-179).  Same signatures as circl_hip_mldsa_sign on n copies of sk. */
int circl_hip_mldsa_sign_shared(int param, const uint8_t *sk, const uint8_t *msg_blob, const uint64_t *msg_off,
                                const uint8_t *ctx_blob, const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sig,
                                size_t n, int device);
int circl_hip_mldsa_sign_shared_dev(int param, const uint8_t *d_sk, const uint8_t *d_msg_blob,
                                    const uint64_t *d_msg_off, const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off,
                                    const uint8_t *d_rnd, int internal, uint8_t *d_sig, size_t n, void *d_workspace,
                                    size_t workspace_bytes, void *stream);
size_t circl_hip_mldsa_sign_workspace_size(int param, size_t n);
int circl_hip_mldsa_sign_dev(int param, const uint8_t *d_sk, const uint8_t *d_msg_blob,
                             const uint64_t *d_msg_off, const uint8_t *d_ctx_blob, const uint64_t *d_ctx_off,
                             const uint8_t *d_rnd, int internal, uint8_t *d_sig, size_t n, void *d_workspace,
                             size_t workspace_bytes, void *stream);

/* ---- primitives (host buffers), mirroring the reference's unit-tested building blocks ----
 * circl_hip_keccak_f1600 : internal/sha3/keccakf.go:12 KeccakF1600 / simd/keccakf1600 StateX4.Permute
 *                          on n states of 25 little-endian uint64 words each; rounds = 24 or 12.
 * circl_hip_kyber_ntt    : pke/kyber/internal/common Poly.NTT (inverse=0) / Poly.InvNTT (1) on n
 *                          polynomials int16[256] in standard order.  Outputs are normalised to
 *                          [0,q) (the reference compares inverse transforms after Normalize,
 *                          ntt_test.go:64-81).
 * circl_hip_kyber_mulhat : Poly.MulHat, out[i] = a[i] (*) b[i], normalised.
 * circl_hip_dilithium_ntt: sign/internal/dilithium Poly.NTT / InvNTT on n polynomials uint32[256],
 *                          outputs normalised to [0,q).
 * circl_hip_shake        : n independent SHAKE128 (rate=168) / SHAKE256 (136) / SHA3-256 (136, ds 6)
 *                          / SHA3-512 (72, ds 6) computations over equal-length inputs
 *                          (internal/sha3 State.Write/Read).
 */
int circl_hip_keccak_f1600(uint64_t *states, size_t n, int rounds, int device);
/* the same 24-round permutation through the wave-cooperative form (one state per wavefront, lanes exchange through LDS)
 * that the ML-DSA kernels use on a rare serial path; exported so that the parity tests can pin it */
int circl_hip_keccak_f1600_coop(uint64_t *states, size_t n, int device);
/* the same through the two-lanes-per-state form (low halves on the even lane, high halves on the odd one, DPP exchange for
 * the rotations) that the small- and medium-batch hashing paths use; exported so that the parity tests can pin it */
int circl_hip_keccak_f1600_split(uint64_t *states, size_t n, int device);
/* sign/mldsa/mldsa65/internal/sample.go:299-339 PolyDeriveUniformBall: n challenge seeds c~ (32 / 48 / 64 bytes for
 * ML-DSA-44 / 65 / 87, 32 for the round-3 modes) -> n polynomials uint32[256] with tau coefficients in {1, q-1}.
 * sequential != 0 runs the reference-order byte scan (otherwise the fallback of the block-parallel form). */
int circl_hip_mldsa_sample_in_ball(int param, const uint8_t *ctilde, uint32_t *polys, size_t n, int sequential,
                                   int device);
int circl_hip_kyber_ntt(int16_t *polys, size_t n, int inverse, int device);
int circl_hip_kyber_mulhat(int16_t *out, const int16_t *a, const int16_t *b, size_t n, int device);

/* Lane-local arithmetic of the device code, elementwise over uint32 arrays (parity tests: the reference checks these
 * functions over their whole domains -- pke/kyber/internal/common/poly_test.go:351-378 compress = exact rounding,
 * sign/mldsa/mldsa65/internal/rounding_test.go:14-67 decompose / makeHint / useHint, field_test.go Montgomery / Barrett --
 * and the GPU instantiation uses different instructions than the host one).  out1 may be NULL; b may be NULL when unused. */
#define CIRCL_HIP_LANE_KYBER_COMPRESS 1    /* arg = d in {4,5,10,11}: out0 = round(a 2^d / q) mod 2^d for ANY representative a < 4q (poly.go:248-332) */
#define CIRCL_HIP_LANE_KYBER_DECOMPRESS 2  /* arg = d in {1,4,5,10,11}: out0 = round(a q / 2^d) (poly.go:170-243) */
#define CIRCL_HIP_LANE_KYBER_MSG_BIT 3     /* out0 = message bit of a in [0,q) (poly.go:150-165) */
#define CIRCL_HIP_LANE_KYBER_MULC 4        /* out0 = a b mod q in [0,q) for a < 2^32/q, via the R = 2^32 constant product (field.go:4-32 replaced) */
#define CIRCL_HIP_LANE_KYBER_REDUCE32 5    /* out0 = -a 2^-32 mod q in [0,q) for any 32-bit a */
#define CIRCL_HIP_LANE_KYBER_NORMALIZE 6   /* a = int16 in the low half: out0 = Normalize (poly.go:35-39), out1 = barrettReduce (field.go:45-64) */
#define CIRCL_HIP_LANE_KYBER_CBD2_WORD 7   /* out0 = the eight biased eta = 2 samples of the PRF word a, one per nibble (sample.go:67-95) */
#define CIRCL_HIP_LANE_KYBER_DOT2 8        /* out0 = lo16(a) lo16(b) + hi16(a) hi16(b) + arg (signed 16-bit halves): MulHat's multiply-accumulate */
#define CIRCL_HIP_LANE_DIL_DECOMPOSE 9     /* arg = gamma2 (95232 | 261888): out0 = a0 + q, out1 = a1 (rounding.go:13-43) */
#define CIRCL_HIP_LANE_DIL_USE_HINT 10     /* arg = gamma2, b = hint: out0 = useHint(a, b) (rounding.go:72-81) */
#define CIRCL_HIP_LANE_DIL_MAKE_HINT 11    /* arg = gamma2, a = z0, b = r1: out0 = makeHint (rounding.go:56-70) */
#define CIRCL_HIP_LANE_DIL_POWER2ROUND 12  /* out0 = a0 + q, out1 = a1 (sign/internal/dilithium/field.go:35-52) */
#define CIRCL_HIP_LANE_DIL_MONT32 13       /* out0 = a b 2^-32 mod q in (0, 2q) for a b < 2^32 q (field.go:20-24) */
#define CIRCL_HIP_LANE_DIL_MONT64 14       /* the same reduction of the 64-bit value b 2^32 + a < 2^32 q */
#define CIRCL_HIP_LANE_DIL_NORMALIZE 15    /* out0 = a mod q for any 32-bit a, out1 = the partial reduction fold(a) < 2^24 (field.go:5-13, :27-31) */
#define CIRCL_HIP_LANE_DIL_EXCEEDS 16      /* b = bound: out0 = 1 iff the centred |a| >= bound, a in [0,q) (poly.go:51-71) */
#define CIRCL_HIP_LANE_OP_COUNT 17
int circl_hip_lane_op(int op, int arg, const uint32_t *a, const uint32_t *b, uint32_t *out0, uint32_t *out1, size_t n, int device);

/* The samplers of the fused kernels with caller-chosen arguments (parity tests: the reference's fixed vectors use arguments the
 * fused kernels never do -- pke/kyber/internal/common/sample_test.go:23-138, sign/mldsa/mldsa65/internal/sample_test.go:12-63).
 * circl_hip_kyber_sample_uniform: Poly.DeriveUniform(seed_i, x_i, y_i) (sample.go:192-236), xy = n pairs of bytes ->
 *   polys[n][256] int16 in coefficient order, values in [0,q).
 * circl_hip_kyber_sample_cbd: Poly.DeriveNoise(seed_i, nonce, eta) for every nonce 0..63 (sample.go:17-95) ->
 *   polys[n][64][256] int16, centred values in [-eta, eta]; eta in {2, 3}.
 * circl_hip_mldsa_sample_uniform: PolyDeriveUniform(seed_i, nonce_i) (sign/mldsa/mldsa65/internal/sample.go:92-123) ->
 *   polys[n][256] uint32 in [0,q). */
int circl_hip_kyber_sample_uniform(const uint8_t *seed32, const uint8_t *xy, int16_t *polys, size_t n, int device);
int circl_hip_kyber_sample_cbd(int eta, const uint8_t *seed32, int16_t *polys, size_t n, int device);
int circl_hip_mldsa_sample_uniform(const uint8_t *seed32, const uint16_t *nonce, uint32_t *polys, size_t n, int device);
int circl_hip_dilithium_ntt(uint32_t *polys, size_t n, int inverse, int device);
int circl_hip_shake(int rate, int ds, const uint8_t *in, size_t inlen, uint8_t *out, size_t outlen,
                    size_t n, int device);
/* Batched XOF service (xof/xof.go, internal/sha3/shake.go:56-100): n independent
 * sponges over variable-length messages (blob + n+1 offsets), `rounds` = 24 (SHA3 / SHAKE) or 12
 * (TurboSHAKE128/256 with domain byte `ds` in 0x01..0x7f), every output `outlen` bytes. */
int circl_hip_xof(int rate, int ds, int rounds, const uint8_t *in_blob, const uint64_t *in_off,
                  uint8_t *out, size_t outlen, size_t n, int device);
/* KangarooTwelve draft -10, n independent computations: xof/k12/k12.go Draft10Sum(hash, msg_i, ctx_i)
 * (and xof.New(xof.K12D10) for an empty context, xof/xof.go:61-64).  Messages and contexts are blobs with
 * n+1 offsets; ctx_blob may be NULL (all contexts empty); every output is `outlen` bytes.  Leaves and final
 * nodes run as TurboSHAKE128 batches on the device. */
int circl_hip_k12(const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob,
                  const uint64_t *ctx_off, uint8_t *out, size_t outlen, size_t n, int device);

/* ---- X25519 (the Diffie-Hellman half of X25519MLKEM768 and X-Wing) -------
 * Replaces dh/x25519: Shared(shared, secret, public) (key.go:41-47) with point != NULL, KeyGen(public, secret)
 * (key.go:34-36) with point == NULL (base point u = 9).  scalar, point, out are n rows of 32 bytes; the scalar is
 * clamped and bit 255 of the point is ignored, as the reference does.  ok[i] = 0 where Shared returns false (the point,
 * reduced mod 2^255 - 19, is one of the five low-order u-coordinates of curve.go:71-96; out[i] is then all zero),
 * 1 otherwise; ok may be NULL.  One Montgomery ladder per lane; device pointers 4-byte aligned. */
int circl_hip_x25519(const uint8_t *scalar, const uint8_t *point, uint8_t *out, uint8_t *ok, size_t n, int device);
int circl_hip_x25519_dev(const uint8_t *d_scalar, const uint8_t *d_point, uint8_t *d_out, uint8_t *d_ok, size_t n, void *stream);

/* ---- hybrid KEMs around ML-KEM-768, composed on the device -------------------------------
 * scheme = CIRCL_HIP_HYBRID_XWING: kem/xwing/xwing.go -- DeriveKeyPairPacked (:98-144), EncapsulateTo (:223-265),
 *   DecapsulateTo (:270-299), combiner (:53-71).  seed 32, eseed 64 (seedm || ekx), pk 1216 (ek || pk_X), sk 32 (the
 *   seed), ct 1120 (ct_M || ct_X), ss 32.  status[i] = 1 (kem.ErrPubKey) when the ML-KEM half of pk fails the
 *   encapsulation-key check (xwing.go:301-311); a low-order X25519 point is not an error (xwing.go:251-254).
 * scheme = CIRCL_HIP_HYBRID_X25519MLKEM768: kem/hybrid/hybrid.go (:236-323) with X25519 as a KEM (xkem.go:112-196).
 *   seed 64, eseed 32, pk 1216 (ek || pk_X), sk 2432 (dk || sk_X), ct 1120, ss 64 (ss_M || ss_X).  status[i] = 1
 *   (kem.ErrPubKey) for a non-canonical ek or a low-order X25519 point, 2 (kem.ErrPrivKey) for a private key failing its
 *   hash check; the item's outputs are then zero (the reference returns nil, err).
 * scheme = CIRCL_HIP_HYBRID_KYBER768_X25519 / _KYBER512_X25519: the same hybrid.go scheme with X25519 as the FIRST component
 *   and round-3 Kyber (kem/kyber) as the second: pk = pk_X || ek (1216 / 832), sk = sk_X || dk (2432 / 1664), ct = ct_X ||
 *   ct_K (1120 / 800), ss = ss_X || ss_K (64); seed 64 (SHAKE256 -> 32 for X25519, then 64 for Kyber), eseed 32.  Round-3
 *   Kyber has no per-item failure, so status is 1 only for a low-order X25519 point.
 * The deterministic forms only (EncapsulateDeterministically / DeriveKeyPair): randomness stays with the caller.
 * The _dev forms zero the secret temporaries in the workspace (seeds, X25519 scalars, private keys, half secrets) before
 * they return control of the stream; the ML-KEM workspace behind them follows the rules of the ML-KEM entry points.
 * status may be NULL on the host forms.  The _dev forms need circl_hip_hybrid_workspace_size(scheme, n) bytes, 16-byte
 * aligned arrays, and a non-NULL d_status. */
#define CIRCL_HIP_HYBRID_XWING 1
#define CIRCL_HIP_HYBRID_X25519MLKEM768 2
#define CIRCL_HIP_HYBRID_KYBER768_X25519 3 /* hybrid.Kyber768X25519(): X25519 first, round-3 Kyber768 second (hybrid.go:77-81) */
#define CIRCL_HIP_HYBRID_KYBER512_X25519 4 /* hybrid.Kyber512X25519() (hybrid.go:71-75) */
size_t circl_hip_hybrid_seed_size(int scheme);
size_t circl_hip_hybrid_eseed_size(int scheme);
size_t circl_hip_hybrid_pk_size(int scheme);
size_t circl_hip_hybrid_sk_size(int scheme);
size_t circl_hip_hybrid_ct_size(int scheme);
size_t circl_hip_hybrid_ss_size(int scheme);
size_t circl_hip_hybrid_workspace_size(int scheme, size_t n);
int circl_hip_hybrid_keygen(int scheme, const uint8_t *seed, uint8_t *pk, uint8_t *sk, size_t n, int device);
int circl_hip_hybrid_encaps(int scheme, const uint8_t *pk, const uint8_t *eseed, uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, int device);
int circl_hip_hybrid_decaps(int scheme, const uint8_t *sk, const uint8_t *ct, uint8_t *ss, uint8_t *status, size_t n, int device);
int circl_hip_hybrid_keygen_dev(int scheme, const uint8_t *d_seed, uint8_t *d_pk, uint8_t *d_sk, size_t n, void *d_ws, size_t ws_bytes, void *stream);
int circl_hip_hybrid_encaps_dev(int scheme, const uint8_t *d_pk, const uint8_t *d_eseed, uint8_t *d_ct, uint8_t *d_ss, uint8_t *d_status, size_t n,
                                void *d_ws, size_t ws_bytes, void *stream);
int circl_hip_hybrid_decaps_dev(int scheme, const uint8_t *d_sk, const uint8_t *d_ct, uint8_t *d_ss, uint8_t *d_status, size_t n, void *d_ws,
                                size_t ws_bytes, void *stream);

/* Hybrid key tables that live across calls (scheme = CIRCL_HIP_HYBRID_XWING or _X25519MLKEM768): kem/xwing's PrivateKey keeps the
 * expanded ML-KEM-768 key, the X25519 scalar and its public point next to the 32-byte seed (xwing.go:20-25), its PublicKey the
 * parsed ML-KEM key; kem/hybrid's keys hold the component schemes' parsed keys (hybrid.go:101-114).  keys = nkeys packed public
 * (private_keys = 0) or private (1) keys of the scheme; an X-Wing private key (the seed) is expanded ONCE, on the device.  The table
 * holds an ML-KEM key table of the lattice halves (A^T, H(ek), hash verdicts -> key_status[nkeys] if not NULL) and the X25519
 * rows.  Item i uses entry key_idx[i] (NULL: entry 0).  Results are those of circl_hip_hybrid_encaps / _decaps on the gathered
 * keys.  The _dev forms need circl_hip_hybrid_workspace_size(scheme, n) bytes. */
int circl_hip_hybrid_keytable_new(int scheme, int private_keys, const uint8_t *keys, size_t nkeys, int device, uint8_t *key_status,
                                  circl_hip_keytable **out);
int circl_hip_hybrid_encaps_table(const circl_hip_keytable *table, const uint32_t *key_idx, const uint8_t *eseed, uint8_t *ct, uint8_t *ss,
                                  uint8_t *status, size_t n);
int circl_hip_hybrid_decaps_table(const circl_hip_keytable *table, const uint32_t *key_idx, const uint8_t *ct, uint8_t *ss, uint8_t *status,
                                  size_t n);
int circl_hip_hybrid_encaps_table_dev(const circl_hip_keytable *table, const uint32_t *d_key_idx, const uint8_t *d_eseed, uint8_t *d_ct,
                                      uint8_t *d_ss, uint8_t *d_status, size_t n, void *d_ws, size_t ws_bytes, void *stream);
int circl_hip_hybrid_decaps_table_dev(const circl_hip_keytable *table, const uint32_t *d_key_idx, const uint8_t *d_ct, uint8_t *d_ss,
                                      uint8_t *d_status, size_t n, void *d_ws, size_t ws_bytes, void *stream);

/* ---- kernel-level profiling (used by bench.py for the roofline figures) --------------------
 * While enabled, every *_dev call brackets each kernel it enqueues with HIP events recorded on
 * the caller's stream.  circl_hip_profile_read synchronises the pending events, returns the
 * accumulated device time and launch count of one kernel and resets that kernel's counters. */
#define CIRCL_HIP_KERNEL_MLKEM_HASH 0     /* H(ek), G(m||H(ek))                      */
#define CIRCL_HIP_KERNEL_MLKEM_ENCRYPT 1  /* matrix expansion + PRF + K-PKE.Encrypt  */
#define CIRCL_HIP_KERNEL_MLKEM_DECRYPT 2  /* K-PKE.Decrypt + G + J                   */
#define CIRCL_HIP_KERNEL_MLKEM_KEYGEN 3
#define CIRCL_HIP_KERNEL_MLKEM_FINISH 4   /* decapsulation compare / select          */
#define CIRCL_HIP_KERNEL_MLDSA_HASH 5
#define CIRCL_HIP_KERNEL_MLDSA_VERIFY 6
#define CIRCL_HIP_KERNEL_MLDSA_KEYGEN 7
#define CIRCL_HIP_KERNEL_MLDSA_SIGN 8
#define CIRCL_HIP_KERNEL_MLKEM_KEYTABLE 9  /* key tables: H(ek) + A^T per table entry */
#define CIRCL_HIP_KERNEL_MLDSA_KEYTABLE 10 /* key tables: tr + ExpandA per table entry */
#define CIRCL_HIP_KERNEL_X25519 11         /* X25519 ladder                            */
#define CIRCL_HIP_KERNEL_COUNT 12
int circl_hip_profile_enable(int on);
int circl_hip_profile_read(int kernel, double *total_ms, uint64_t *launches);
/* The VALU issue rates this chip sustains, measured live (bench.py prices the kernels' VALU time against them instead of against
 * figures of an earlier round): wave-instructions per second per SIMD of (a) the library's own Keccak-f[1600] round running on
 * register-resident states (keccak_dev.h: V_BITOP3 / V_ALIGNBIT, no memory traffic) and (b) two-operand integer VALU instructions
 * (the cheapest class), with `waves_per_simd` (1..8; the Keccak probe needs ~110 VGPRs, so at most 4 are resident) wavefronts
 * on every SIMD of `device`.  About 10 ms of GPU time.  Either output may be NULL. */
int circl_hip_profile_valu_probe(int device, int waves_per_simd, double *keccak_insts_per_s_per_simd, double *simple_insts_per_s_per_simd);
/* Where the microseconds of ONE blocking coalesced call go (profiles/r06_one_call.txt): enable != 0 switches the time stamps on, 0 off,
 * < 0 leaves the switch alone; out8 (may be NULL) receives the CLOCK_MONOTONIC nanoseconds of the calling thread's LAST blocking
 * coalesced call: [0] entered, [1] rows reserved, [2] own inputs copied in, and -- for the call that led its batch, else 0 -- [3] batch
 * closed (its turn, room on the device), [4] every caller's inputs in, [5] copies + kernels enqueued, [6] batch finished; [7] own results
 * copied out. */
int circl_hip_profile_call_stamps(int enable, uint64_t *out8);

/* What the host-buffer pipeline's staging pools hold right now, over all (logical) devices: slots created, page-locked host bytes,
 * device bytes.  The pools grow on demand (up to CIRCL_HIP_HOST_SLOTS slots per device) and are never shrunk: after a process's largest
 * calls this is the high-water mark a node has to budget for (profiles/r05_logical8.txt).  Any pointer may be NULL. */
int circl_hip_host_pool_stats(int *slots, uint64_t *pinned_bytes, uint64_t *device_bytes);

/* pinned host memory helpers for callers that want zero-copy-speed transfers */
void *circl_hip_alloc_host(size_t bytes);
void circl_hip_free_host(void *p);

#ifdef __cplusplus
}
#endif
#endif
