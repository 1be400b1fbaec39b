"""Host-buffer (numpy) wrappers over the C ABI -- what the cgo bridge would call."""
import ctypes as C

import numpy as np

from . import _native as nat

KEM_SIZES = {512: (800, 1632, 768), 768: (1184, 2400, 1088), 1024: (1568, 3168, 1568)}  # ek, dk, ct
# pk, sig; 2 / 3 / 5 = round-3 Dilithium2/3/5 (sign/dilithium/mode{2,3,5}): no context, deterministic, 32-byte tr and c~
DSA_SIZES = {44: (1312, 2420), 65: (1952, 3309), 87: (2592, 4627), 2: (1312, 2420), 3: (1952, 3293), 5: (2592, 4595)}
DSA_SK_SIZES = {44: 2560, 65: 4032, 87: 4896, 2: 2528, 3: 4000, 5: 4864}


def _u8(x, cols):
    a = np.ascontiguousarray(np.frombuffer(x, np.uint8) if isinstance(x, (bytes, bytearray)) else x, dtype=np.uint8)
    return a.reshape(-1, cols)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    return nat.lib().circl_hip_device_count()


def mlkem_encaps(param, ek, m, device=0):
    EK, _, CT = KEM_SIZES[param]
    ek, m = _u8(ek, EK), _u8(m, 32)
    n = len(ek)
    assert len(m) == n
    ct = np.empty((n, CT), np.uint8)
    ss = np.empty((n, 32), np.uint8)
    st = np.empty(n, np.uint8)
    nat.check(nat.lib().circl_hip_mlkem_encaps(param, _p(ek), _p(m), _p(ct), _p(ss), _p(st), n, device), "mlkem_encaps")
    return ct, ss, st


def mlkem_decaps(param, dk, ct, device=0):
    _, DK, CT = KEM_SIZES[param]
    dk, ct = _u8(dk, DK), _u8(ct, CT)
    n = len(dk)
    assert len(ct) == n
    ss = np.empty((n, 32), np.uint8)
    st = np.empty(n, np.uint8)
    nat.check(nat.lib().circl_hip_mlkem_decaps(param, _p(dk), _p(ct), _p(ss), _p(st), n, device), "mlkem_decaps")
    return ss, st


def mlkem_keygen(param, seeds, device=0):
    EK, DK, _ = KEM_SIZES[param]
    seeds = _u8(seeds, 64)
    n = len(seeds)
    ek = np.empty((n, EK), np.uint8)
    dk = np.empty((n, DK), np.uint8)
    nat.check(nat.lib().circl_hip_mlkem_keygen(param, _p(seeds), _p(ek), _p(dk), n, device), "mlkem_keygen")
    return ek, dk


def mlkem_encaps_shared(param, ek, m, device=0):
    """one key for the whole batch -> ct, ss, status"""
    EK, _, CT = KEM_SIZES[param]
    ek, m = _u8(ek, EK), _u8(m, 32)
    assert len(ek) == 1
    n = len(m)
    ct = np.empty((n, CT), np.uint8)
    ss = np.empty((n, 32), np.uint8)
    st = np.empty(n, np.uint8)
    nat.check(nat.lib().circl_hip_mlkem_encaps_shared(param, _p(ek), _p(m), _p(ct), _p(ss), _p(st), n, device), "mlkem_encaps_shared")
    return ct, ss, st


def mlkem_decaps_shared(param, dk, ct, device=0):
    """one private key for the whole batch -> ss, status"""
    _, DK, CT = KEM_SIZES[param]
    dk, ct = _u8(dk, DK), _u8(ct, CT)
    assert len(dk) == 1
    n = len(ct)
    ss = np.empty((n, 32), np.uint8)
    st = np.empty(n, np.uint8)
    nat.check(nat.lib().circl_hip_mlkem_decaps_shared(param, _p(dk), _p(ct), _p(ss), _p(st), n, device), "mlkem_decaps_shared")
    return ss, st


def _idx(key_idx, n):
    a = np.ascontiguousarray(key_idx, dtype=np.uint32).reshape(-1)
    assert len(a) == n
    return a


def mlkem_encaps_keyed(param, ek_table, key_idx, m, device=0):
    """item i encapsulates to row key_idx[i] of ek_table -> ct, ss, status"""
    EK, _, CT = KEM_SIZES[param]
    ek_table, m = _u8(ek_table, EK), _u8(m, 32)
    n = len(m)
    idx = _idx(key_idx, n)
    ct = np.empty((n, CT), np.uint8)
    ss = np.empty((n, 32), np.uint8)
    st = np.empty(n, np.uint8)
    nat.check(nat.lib().circl_hip_mlkem_encaps_keyed(param, _p(ek_table), len(ek_table), _p(idx), _p(m), _p(ct), _p(ss), _p(st), n, device),
              "mlkem_encaps_keyed")
    return ct, ss, st


def mlkem_decaps_keyed(param, dk_table, key_idx, ct, device=0):
    """item i is decapsulated with row key_idx[i] of dk_table -> ss, status"""
    _, DK, CT = KEM_SIZES[param]
    dk_table, ct = _u8(dk_table, DK), _u8(ct, CT)
    n = len(ct)
    idx = _idx(key_idx, n)
    ss = np.empty((n, 32), np.uint8)
    st = np.empty(n, np.uint8)
    nat.check(nat.lib().circl_hip_mlkem_decaps_keyed(param, _p(dk_table), len(dk_table), _p(idx), _p(ct), _p(ss), _p(st), n, device),
              "mlkem_decaps_keyed")
    return ss, st


# round-3 Kyber (kem/kyber/kyber{512,768,1024}): no per-item failures
def kyber_keygen(param, seeds, device=0):
    EK, DK, _ = KEM_SIZES[param]
    seeds = _u8(seeds, 64)
    n = len(seeds)
    ek = np.empty((n, EK), np.uint8)
    dk = np.empty((n, DK), np.uint8)
    nat.check(nat.lib().circl_hip_kyber_keygen(param, _p(seeds), _p(ek), _p(dk), n, device), "kyber_keygen")
    return ek, dk


def kyber_encaps(param, ek, seeds, device=0):
    EK, _, CT = KEM_SIZES[param]
    ek, seeds = _u8(ek, EK), _u8(seeds, 32)
    n = len(ek)
    assert len(seeds) == n
    ct = np.empty((n, CT), np.uint8)
    ss = np.empty((n, 32), np.uint8)
    nat.check(nat.lib().circl_hip_kyber_encaps(param, _p(ek), _p(seeds), _p(ct), _p(ss), n, device), "kyber_encaps")
    return ct, ss


def kyber_decaps(param, dk, ct, device=0):
    _, DK, CT = KEM_SIZES[param]
    dk, ct = _u8(dk, DK), _u8(ct, CT)
    n = len(dk)
    assert len(ct) == n
    ss = np.empty((n, 32), np.uint8)
    nat.check(nat.lib().circl_hip_kyber_decaps(param, _p(dk), _p(ct), _p(ss), n, device), "kyber_decaps")
    return ss


def _blob(items):
    off = np.zeros(len(items) + 1, np.uint64)
    if len(items):
        off[1:] = np.cumsum([len(x) for x in items])
    blob = np.frombuffer(b"".join(bytes(x) for x in items) + b"\0" * 16, dtype=np.uint8).copy()
    return blob, off


def mldsa_keygen(param, seeds, device=0):
    PK, _ = DSA_SIZES[param]
    SK = DSA_SK_SIZES[param]
    seeds = _u8(seeds, 32)
    n = len(seeds)
    pk = np.empty((n, PK), np.uint8)
    sk = np.empty((n, SK), np.uint8)
    nat.check(nat.lib().circl_hip_mldsa_keygen(param, _p(seeds), _p(pk), _p(sk), n, device), "mldsa_keygen")
    return pk, sk


def mldsa_sign(param, sk, msgs, ctxs=None, rnd=None, internal=False, device=0):
    """deterministic when rnd is None"""
    _, SIG = DSA_SIZES[param]
    SK = DSA_SK_SIZES[param]
    sk = _u8(sk, SK)
    n = len(sk)
    assert len(msgs) == n
    mb, mo = _blob(msgs)
    sig = np.empty((n, SIG), np.uint8)
    r = None if rnd is None else _p(_u8(rnd, 32))
    if internal:
        rc = nat.lib().circl_hip_mldsa_sign_internal(param, _p(sk), _p(mb), _p(mo), r, _p(sig), n, device)
    elif ctxs is None:
        rc = nat.lib().circl_hip_mldsa_sign(param, _p(sk), _p(mb), _p(mo), None, None, r, _p(sig), n, device)
    else:
        cb, co = _blob(ctxs)
        rc = nat.lib().circl_hip_mldsa_sign(param, _p(sk), _p(mb), _p(mo), _p(cb), _p(co), r, _p(sig), n, device)
    nat.check(rc, "mldsa_sign")
    return sig


def mldsa_sign_shared(param, sk, msgs, ctxs=None, rnd=None, device=0):
    """n messages signed with ONE private key -> (n, SIG)"""
    _, SIG = DSA_SIZES[param]
    sk = _u8(sk, DSA_SK_SIZES[param])
    assert len(sk) == 1
    n = len(msgs)
    mb, mo = _blob(msgs)
    sig = np.empty((n, SIG), np.uint8)
    r = None if rnd is None else _p(_u8(rnd, 32))
    if ctxs is None:
        rc = nat.lib().circl_hip_mldsa_sign_shared(param, _p(sk), _p(mb), _p(mo), None, None, r, _p(sig), n, device)
    else:
        cb, co = _blob(ctxs)
        rc = nat.lib().circl_hip_mldsa_sign_shared(param, _p(sk), _p(mb), _p(mo), _p(cb), _p(co), r, _p(sig), n, device)
    nat.check(rc, "mldsa_sign_shared")
    return sig


def mldsa_verify_internal(param, pk, sig, msgs, device=0):
    PK, SIG = DSA_SIZES[param]
    pk, sig = _u8(pk, PK), _u8(sig, SIG)
    n = len(pk)
    mb, mo = _blob(msgs)
    ok = np.empty(n, np.uint8)
    nat.check(nat.lib().circl_hip_mldsa_verify_internal(param, _p(pk), _p(sig), _p(mb), _p(mo), _p(ok), n, device), "mldsa_verify_internal")
    return ok


def mldsa_verify(param, pk, sig, msgs, ctxs=None, device=0):
    PK, SIG = DSA_SIZES[param]
    pk, sig = _u8(pk, PK), _u8(sig, SIG)
    n = len(pk)
    assert len(sig) == n and len(msgs) == n
    mb, mo = _blob(msgs)
    ok = np.empty(n, np.uint8)
    if ctxs is None:
        rc = nat.lib().circl_hip_mldsa_verify(param, _p(pk), _p(sig), _p(mb), _p(mo), None, None, _p(ok), n, device)
    else:
        cb, co = _blob(ctxs)
        rc = nat.lib().circl_hip_mldsa_verify(param, _p(pk), _p(sig), _p(mb), _p(mo), _p(cb), _p(co), _p(ok), n, device)
    nat.check(rc, "mldsa_verify")
    return ok


def mldsa_verify_shared(param, pk, sig, msgs, ctxs=None, device=0):
    """n signatures under ONE public key -> ok (n,)"""
    PK, SIG = DSA_SIZES[param]
    pk, sig = _u8(pk, PK), _u8(sig, SIG)
    assert len(pk) == 1
    n = len(sig)
    assert len(msgs) == n
    mb, mo = _blob(msgs)
    ok = np.empty(n, np.uint8)
    if ctxs is None:
        rc = nat.lib().circl_hip_mldsa_verify_shared(param, _p(pk), _p(sig), _p(mb), _p(mo), None, None, _p(ok), n, device)
    else:
        cb, co = _blob(ctxs)
        rc = nat.lib().circl_hip_mldsa_verify_shared(param, _p(pk), _p(sig), _p(mb), _p(mo), _p(cb), _p(co), _p(ok), n, device)
    nat.check(rc, "mldsa_verify_shared")
    return ok


def mldsa_verify_keyed(param, pk_table, key_idx, sig, msgs, ctxs=None, device=0):
    """signature i is checked under row key_idx[i] of pk_table -> ok (n,)"""
    PK, SIG = DSA_SIZES[param]
    pk_table, sig = _u8(pk_table, PK), _u8(sig, SIG)
    n = len(sig)
    assert len(msgs) == n
    idx = _idx(key_idx, n)
    mb, mo = _blob(msgs)
    ok = np.empty(n, np.uint8)
    if ctxs is None:
        rc = nat.lib().circl_hip_mldsa_verify_keyed(param, _p(pk_table), len(pk_table), _p(idx), _p(sig), _p(mb), _p(mo), None, None, _p(ok), n, device)
    else:
        cb, co = _blob(ctxs)
        rc = nat.lib().circl_hip_mldsa_verify_keyed(param, _p(pk_table), len(pk_table), _p(idx), _p(sig), _p(mb), _p(mo), _p(cb), _p(co), _p(ok), n, device)
    nat.check(rc, "mldsa_verify_keyed")
    return ok


class KeyTable:
    """A parsed-key cache that lives across calls (circl_hip_*_keytable_new): the counterpart of CIRCL's key objects, which keep
    A^T / H(ek) (ML-KEM) or A / tr (ML-DSA) after unmarshalling.  kind: "mlkem-public", "mlkem-private", "mldsa-public", "mldsa-private",
    "hybrid-public", "hybrid-private" (param = the hybrid scheme id).  device = -1: replicated on every device, calls shard the batch."""

    def __init__(self, kind, param, keys, device=0):
        import ctypes as C
        self.kind, self.param, self.device = kind, param, device
        self.handle = C.c_void_p()
        self.key_status = None
        L = nat.lib()
        if kind.startswith("mlkem"):
            EK, DK, _ = KEM_SIZES[param]
            priv = kind == "mlkem-private"
            keys = _u8(keys, DK if priv else EK)
            self.nkeys = len(keys)
            self.key_status = np.zeros(self.nkeys, np.uint8)
            nat.check(L.circl_hip_mlkem_keytable_new(param, 1 if priv else 0, _p(keys), self.nkeys, device, _p(self.key_status), C.byref(self.handle)),
                      "mlkem_keytable_new")
        elif kind == "mldsa-private":
            keys = _u8(keys, nat.lib().circl_hip_mldsa_sk_size(param))
            self.nkeys = len(keys)
            nat.check(L.circl_hip_mldsa_privkeys_new(param, _p(keys), self.nkeys, device, C.byref(self.handle)), "mldsa_privkeys_new")
        elif kind.startswith("hybrid"):  # param = the hybrid scheme id (XWING, X25519MLKEM768)
            priv = kind == "hybrid-private"
            keys = _u8(keys, HYBRID_SIZES[param]["sk" if priv else "pk"])
            self.nkeys = len(keys)
            self.key_status = np.zeros(self.nkeys, np.uint8)
            nat.check(L.circl_hip_hybrid_keytable_new(param, 1 if priv else 0, _p(keys), self.nkeys, device, _p(self.key_status), C.byref(self.handle)),
                      "hybrid_keytable_new")
        else:
            PK, _ = DSA_SIZES[param]
            keys = _u8(keys, PK)
            self.nkeys = len(keys)
            nat.check(L.circl_hip_mldsa_keytable_new(param, _p(keys), self.nkeys, device, C.byref(self.handle)), "mldsa_keytable_new")

    def close(self):
        if self.handle:
            nat.lib().circl_hip_keytable_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_coalesce(self, max_items, max_wait_us=0):
        """circl_hip_keytable_set_coalesce: small calls of concurrent callers through this table share launches (0 = off)"""
        nat.check(nat.lib().circl_hip_keytable_set_coalesce(self.handle, max_items, max_wait_us), "keytable_set_coalesce")

    # ---- the asynchronous form (circl_hip_keytable_async_start / *_table_submit / circl_hip_poll / circl_hip_wait) ----
    def async_start(self, max_items, max_wait_us=0, eventfd=False):
        nat.check(nat.lib().circl_hip_keytable_async_start(self.handle, max_items, max_wait_us, 1 if eventfd else 0), "keytable_async_start")

    def async_stop(self):
        """CIRCL_HIP_OK, or CIRCL_HIP_EBUSY (returned, not raised) while calls are inside the table"""
        return nat.lib().circl_hip_keytable_async_stop(self.handle)

    def eventfd(self, replica=0):
        return nat.lib().circl_hip_keytable_eventfd(self.handle, replica)

    def submit_encaps(self, m, ct, ss, st, key_idx=None):
        """Returns (rc, ticket); the output arrays (caller-owned, C-contiguous uint8) are filled once the ticket is done."""
        import ctypes as C
        m = _u8(m, 32)
        n = len(m)
        t = C.c_uint64()
        rc = nat.lib().circl_hip_mlkem_encaps_table_submit(self.handle, self._kidx(key_idx, n), _p(m), _p(ct), _p(ss), _p(st), n, C.byref(t))
        return rc, t.value

    def submit_decaps(self, ct, ss, st, key_idx=None):
        import ctypes as C
        _, _, CT = KEM_SIZES[self.param]
        ct = _u8(ct, CT)
        n = len(ct)
        t = C.c_uint64()
        rc = nat.lib().circl_hip_mlkem_decaps_table_submit(self.handle, self._kidx(key_idx, n), _p(ct), _p(ss), _p(st), n, C.byref(t))
        return rc, t.value

    def submit_hybrid_encaps(self, eseeds, ct, ss, st, key_idx=None):
        import ctypes as C
        es = _u8(eseeds, HYBRID_SIZES[self.param]["eseed"])
        n = len(es)
        t = C.c_uint64()
        rc = nat.lib().circl_hip_hybrid_encaps_table_submit(self.handle, self._kidx(key_idx, n), _p(es), _p(ct), _p(ss), _p(st), n, C.byref(t))
        return rc, t.value

    def submit_hybrid_decaps(self, ct_in, ss, st, key_idx=None):
        import ctypes as C
        c = _u8(ct_in, HYBRID_SIZES[self.param]["ct"])
        n = len(c)
        t = C.c_uint64()
        rc = nat.lib().circl_hip_hybrid_decaps_table_submit(self.handle, self._kidx(key_idx, n), _p(c), _p(ss), _p(st), n, C.byref(t))
        return rc, t.value

    def submit_verify(self, sigs, msgs, ok, ctxs=None, key_idx=None):
        import ctypes as C
        _, SIG = DSA_SIZES[self.param]
        sigs = _u8(sigs, SIG)
        n = len(sigs)
        mb, mo = _blob(msgs)
        cb, cofs = _blob(ctxs) if ctxs is not None else (None, None)
        t = C.c_uint64()
        rc = nat.lib().circl_hip_mldsa_verify_table_submit(self.handle, self._kidx(key_idx, n), _p(sigs), _p(mb), _p(mo), _p(cb) if cb is not None else None,
                                                           _p(cofs) if cofs is not None else None, _p(ok), n, C.byref(t))
        return rc, t.value

    def poll(self, tickets):
        """-> list of states (1 done, 0 pending, < 0 failed)"""
        t = np.asarray(tickets, np.uint64)
        st = np.zeros(len(t), np.int8)
        nat.lib().circl_hip_poll(self.handle, _p(t), len(t), _p(st))
        return [int(x) for x in st]

    def wait(self, ticket, timeout_us=-1):
        return nat.lib().circl_hip_wait(self.handle, int(ticket), int(timeout_us))

    def try_close(self):
        """circl_hip_keytable_close: 0 and the table is gone, or CIRCL_HIP_EBUSY"""
        rc = nat.lib().circl_hip_keytable_close(self.handle)
        if rc == 0:
            self.handle = None
        return rc

    def coalesce_stats(self):
        """(calls, items, launches) that went through the table's coalescer"""
        import ctypes as C
        c, i, l = C.c_uint64(), C.c_uint64(), C.c_uint64()
        nat.check(nat.lib().circl_hip_keytable_coalesce_stats(self.handle, C.byref(c), C.byref(i), C.byref(l)), "keytable_coalesce_stats")
        return c.value, i.value, l.value

    def _kidx(self, key_idx, n):
        return None if key_idx is None else _p(_idx(key_idx, n))

    def encaps(self, m, key_idx=None):
        """item i encapsulates to entry key_idx[i] (None: entry 0) -> ct, ss, status"""
        _, _, CT = KEM_SIZES[self.param]
        m = _u8(m, 32)
        n = len(m)
        ct, ss, st = np.empty((n, CT), np.uint8), np.empty((n, 32), np.uint8), np.empty(n, np.uint8)
        nat.check(nat.lib().circl_hip_mlkem_encaps_table(self.handle, self._kidx(key_idx, n), _p(m), _p(ct), _p(ss), _p(st), n), "mlkem_encaps_table")
        return ct, ss, st

    def decaps(self, ct, key_idx=None):
        _, _, CT = KEM_SIZES[self.param]
        ct = _u8(ct, CT)
        n = len(ct)
        ss, st = np.empty((n, 32), np.uint8), np.empty(n, np.uint8)
        nat.check(nat.lib().circl_hip_mlkem_decaps_table(self.handle, self._kidx(key_idx, n), _p(ct), _p(ss), _p(st), n), "mlkem_decaps_table")
        return ss, st

    def sign(self, msgs, ctxs=None, rnd=None, key_idx=None):
        """scheme.Sign with the prepared private key(s): message i with entry key_idx[i] (None: entry 0) -> (n, SIG)"""
        _, SIG = DSA_SIZES[self.param]
        n = len(msgs)
        mb, mo = _blob(msgs)
        sig = np.empty((n, SIG), np.uint8)
        cb, co = _blob(ctxs) if ctxs is not None else (None, None)
        r = None if rnd is None else _u8(rnd, 32)
        if key_idx is None:
            rc = nat.lib().circl_hip_mldsa_sign_table(self.handle, _p(mb), _p(mo), _p(cb) if ctxs is not None else None, _p(co) if ctxs is not None else None,
                                                      None if r is None else _p(r), _p(sig), n)
        else:
            rc = nat.lib().circl_hip_mldsa_sign_table_keyed(self.handle, self._kidx(key_idx, n), _p(mb), _p(mo), _p(cb) if ctxs is not None else None,
                                                            _p(co) if ctxs is not None else None, None if r is None else _p(r), _p(sig), n)
        nat.check(rc, "mldsa_sign_table")
        return sig

    def hybrid_encaps(self, eseeds, key_idx=None):
        S = HYBRID_SIZES[self.param]
        es = _u8(eseeds, S["eseed"])
        n = len(es)
        ct, ss, st = np.empty((n, S["ct"]), np.uint8), np.empty((n, S["ss"]), np.uint8), np.empty(n, np.uint8)
        nat.check(nat.lib().circl_hip_hybrid_encaps_table(self.handle, self._kidx(key_idx, n), _p(es), _p(ct), _p(ss), _p(st), n), "hybrid_encaps_table")
        return ct, ss, st

    def hybrid_decaps(self, ct, key_idx=None):
        S = HYBRID_SIZES[self.param]
        ct = _u8(ct, S["ct"])
        n = len(ct)
        ss, st = np.empty((n, S["ss"]), np.uint8), np.empty(n, np.uint8)
        nat.check(nat.lib().circl_hip_hybrid_decaps_table(self.handle, self._kidx(key_idx, n), _p(ct), _p(ss), _p(st), n), "hybrid_decaps_table")
        return ss, st

    def verify(self, sig, msgs, ctxs=None, key_idx=None):
        _, SIG = DSA_SIZES[self.param]
        sig = _u8(sig, SIG)
        n = len(sig)
        assert len(msgs) == n
        mb, mo = _blob(msgs)
        ok = np.empty(n, np.uint8)
        cb, co = _blob(ctxs) if ctxs is not None else (None, None)
        rc = nat.lib().circl_hip_mldsa_verify_table(self.handle, self._kidx(key_idx, n), _p(sig), _p(mb), _p(mo), _p(cb) if ctxs is not None else None,
                                                    _p(co) if ctxs is not None else None, _p(ok), n)
        nat.check(rc, "mldsa_verify_table")
        return ok


def mlkem_public_from_private(param, dk):
    """PrivateKey.Public() over a batch (kem/mlkem/mlkem768/kyber.go:323-328)"""
    EK, DK, _ = KEM_SIZES[param]
    dk = _u8(dk, DK)
    ek = np.empty((len(dk), EK), np.uint8)
    nat.check(nat.lib().circl_hip_mlkem_public_from_private(param, _p(dk), _p(ek), len(dk)), "mlkem_public_from_private")
    return ek


def mldsa_public_from_private(param, sk, device=0):
    """PrivateKey.Public() over a batch (sign/mldsa/mldsa65/internal/dilithium.go:473-484)"""
    sk = _u8(sk, nat.lib().circl_hip_mldsa_sk_size(param))
    pk = np.empty((len(sk), nat.lib().circl_hip_mldsa_pk_size(param)), np.uint8)
    nat.check(nat.lib().circl_hip_mldsa_public_from_private(param, _p(sk), _p(pk), len(sk), device), "mldsa_public_from_private")
    return pk


def keccak_f1600(states, rounds=24, device=0):
    a = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 25).copy()
    nat.check(nat.lib().circl_hip_keccak_f1600(_p(a), len(a), rounds, device), "keccak_f1600")
    return a


def keccak_f1600_coop(states, device=0):
    a = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 25).copy()
    nat.check(nat.lib().circl_hip_keccak_f1600_coop(_p(a), len(a), device), "keccak_f1600_coop")
    return a


def keccak_f1600_split(states, device=0):
    a = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 25).copy()
    nat.check(nat.lib().circl_hip_keccak_f1600_split(_p(a), len(a), device), "keccak_f1600_split")
    return a


def mldsa_sample_in_ball(param, ctilde, sequential=False, device=0):
    """c~ rows -> (n, 256) uint32 challenge polynomials (PolyDeriveUniformBall)"""
    ct = {44: 32, 65: 48, 87: 64, 2: 32, 3: 32, 5: 32}[param]
    c = _u8(ctilde, ct)
    out = np.empty((len(c), 256), np.uint32)
    nat.check(nat.lib().circl_hip_mldsa_sample_in_ball(param, _p(c), _p(out), len(c), int(sequential), device), "mldsa_sample_in_ball")
    return out


def kyber_ntt(polys, inverse=False, device=0):
    a = np.ascontiguousarray(polys, dtype=np.int16).reshape(-1, 256).copy()
    nat.check(nat.lib().circl_hip_kyber_ntt(_p(a), len(a), int(inverse), device), "kyber_ntt")
    return a


def kyber_mulhat(a, b, device=0):
    a = np.ascontiguousarray(a, dtype=np.int16).reshape(-1, 256)
    b = np.ascontiguousarray(b, dtype=np.int16).reshape(-1, 256)
    out = np.empty_like(a)
    nat.check(nat.lib().circl_hip_kyber_mulhat(_p(out), _p(a), _p(b), len(a), device), "kyber_mulhat")
    return out


def dilithium_ntt(polys, inverse=False, device=0):
    a = np.ascontiguousarray(polys, dtype=np.uint32).reshape(-1, 256).copy()
    nat.check(nat.lib().circl_hip_dilithium_ntt(_p(a), len(a), int(inverse), device), "dilithium_ntt")
    return a


LANE_OPS = dict(KYBER_COMPRESS=1, KYBER_DECOMPRESS=2, KYBER_MSG_BIT=3, KYBER_MULC=4, KYBER_REDUCE32=5, KYBER_NORMALIZE=6, KYBER_CBD2_WORD=7,
                KYBER_DOT2=8, DIL_DECOMPOSE=9, DIL_USE_HINT=10, DIL_MAKE_HINT=11, DIL_POWER2ROUND=12, DIL_MONT32=13, DIL_MONT64=14,
                DIL_NORMALIZE=15, DIL_EXCEEDS=16)


def lane_op(op, a, b=None, arg=0, two=False, device=0):
    """The DEVICE instantiation of a coefficient-level function, elementwise over uint32 arrays -> out0 (, out1 with two=True)"""
    a = np.ascontiguousarray(a, dtype=np.uint32).reshape(-1)
    n = len(a)
    bb = None if b is None else np.ascontiguousarray(np.broadcast_to(np.asarray(b, dtype=np.uint32), a.shape))
    o0 = np.empty(n, np.uint32)
    o1 = np.empty(n, np.uint32) if two else None
    nat.check(nat.lib().circl_hip_lane_op(LANE_OPS[op], int(arg), _p(a), None if bb is None else _p(bb), _p(o0), None if o1 is None else _p(o1), n, device),
              "lane_op " + op)
    return (o0, o1) if two else o0


def kyber_sample_uniform(seeds, xy, device=0):
    """Poly.DeriveUniform(seed_i, x_i, y_i) -> (n, 256) int16 in [0, q)"""
    seeds, xy = _u8(seeds, 32), _u8(xy, 2)
    out = np.empty((len(seeds), 256), np.int16)
    nat.check(nat.lib().circl_hip_kyber_sample_uniform(_p(seeds), _p(xy), _p(out), len(seeds), device), "kyber_sample_uniform")
    return out


def kyber_sample_cbd(eta, seeds, device=0):
    """Poly.DeriveNoise(seed_i, nonce, eta) for nonce 0..63 -> (n, 64, 256) int16 in [-eta, eta]"""
    seeds = _u8(seeds, 32)
    out = np.empty((len(seeds), 64, 256), np.int16)
    nat.check(nat.lib().circl_hip_kyber_sample_cbd(eta, _p(seeds), _p(out), len(seeds), device), "kyber_sample_cbd")
    return out


def mldsa_sample_uniform(seeds, nonces, device=0):
    """PolyDeriveUniform(seed_i, nonce_i) -> (n, 256) uint32 in [0, q)"""
    seeds = _u8(seeds, 32)
    nonces = np.ascontiguousarray(nonces, dtype=np.uint16).reshape(-1)
    assert len(nonces) == len(seeds)
    out = np.empty((len(seeds), 256), np.uint32)
    nat.check(nat.lib().circl_hip_mldsa_sample_uniform(_p(seeds), _p(nonces), _p(out), len(seeds), device), "mldsa_sample_uniform")
    return out


def shake(rate, ds, msgs, outlen, device=0):
    """msgs: (n, inlen) uint8 (equal lengths) -> (n, outlen)"""
    msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
    n, inlen = msgs.shape
    out = np.empty((n, outlen), np.uint8)
    nat.check(nat.lib().circl_hip_shake(rate, ds, _p(msgs), inlen, _p(out), outlen, n, device), "shake")
    return out


def xof(rate, ds, msgs, outlen, rounds=24, device=0):
    """variable-length messages -> (n, outlen); rounds=12 gives TurboSHAKE"""
    n = len(msgs)
    mb, mo = _blob(msgs)
    out = np.empty((n, outlen), np.uint8)
    nat.check(nat.lib().circl_hip_xof(rate, ds, rounds, _p(mb), _p(mo), _p(out), outlen, n, device), "xof")
    return out


def k12(msgs, outlen, ctxs=None, device=0):
    """KangarooTwelve draft -10 of every message (xof/k12 Draft10Sum) -> (n, outlen)"""
    n = len(msgs)
    mb, mo = _blob(msgs)
    out = np.empty((n, outlen), np.uint8)
    if ctxs is None:
        rc = nat.lib().circl_hip_k12(_p(mb), _p(mo), None, None, _p(out), outlen, n, device)
    else:
        cb, co = _blob(ctxs)
        rc = nat.lib().circl_hip_k12(_p(mb), _p(mo), _p(cb), _p(co), _p(out), outlen, n, device)
    nat.check(rc, "k12")
    return out


def x25519(scalar, point=None, device=0):
    """Batch X25519 (dh/x25519): Shared(scalar_i, point_i), or KeyGen(scalar_i) when point is None.
    Returns (out (n, 32), ok (n,)): ok = 0 where the reference's Shared reports a low-order public key."""
    scalar = _u8(scalar, 32)
    n = scalar.shape[0]
    out = np.empty((n, 32), np.uint8)
    ok = np.empty(n, np.uint8)
    pt = None if point is None else _u8(point, 32)
    nat.check(nat.lib().circl_hip_x25519(_p(scalar), None if pt is None else _p(pt), _p(out), _p(ok), n, device), "x25519")
    return out, ok


XWING, X25519MLKEM768, KYBER768_X25519, KYBER512_X25519 = 1, 2, 3, 4
HYBRID_SIZES = {XWING: dict(seed=32, eseed=64, pk=1216, sk=32, ct=1120, ss=32), X25519MLKEM768: dict(seed=64, eseed=32, pk=1216, sk=2432, ct=1120, ss=64),
                KYBER768_X25519: dict(seed=64, eseed=32, pk=1216, sk=2432, ct=1120, ss=64),
                KYBER512_X25519: dict(seed=64, eseed=32, pk=832, sk=1664, ct=800, ss=64)}


def hybrid_keygen(scheme, seeds, device=0):
    """X-Wing DeriveKeyPairPacked / X25519MLKEM768 DeriveKeyPair for every seed -> (pk, sk)"""
    S = HYBRID_SIZES[scheme]
    seeds = _u8(seeds, S["seed"])
    n = seeds.shape[0]
    pk, sk = np.empty((n, S["pk"]), np.uint8), np.empty((n, S["sk"]), np.uint8)
    nat.check(nat.lib().circl_hip_hybrid_keygen(scheme, _p(seeds), _p(pk), _p(sk), n, device), "hybrid keygen")
    return pk, sk


def hybrid_encaps(scheme, pk, eseeds, device=0):
    """deterministic encapsulation -> (ct, ss, status)"""
    S = HYBRID_SIZES[scheme]
    pk, eseeds = _u8(pk, S["pk"]), _u8(eseeds, S["eseed"])
    n = pk.shape[0]
    ct, ss, st = np.empty((n, S["ct"]), np.uint8), np.empty((n, S["ss"]), np.uint8), np.empty(n, np.uint8)
    nat.check(nat.lib().circl_hip_hybrid_encaps(scheme, _p(pk), _p(eseeds), _p(ct), _p(ss), _p(st), n, device), "hybrid encaps")
    return ct, ss, st


def hybrid_decaps(scheme, sk, ct, device=0):
    """-> (ss, status)"""
    S = HYBRID_SIZES[scheme]
    sk, ct = _u8(sk, S["sk"]), _u8(ct, S["ct"])
    n = sk.shape[0]
    ss, st = np.empty((n, S["ss"]), np.uint8), np.empty(n, np.uint8)
    nat.check(nat.lib().circl_hip_hybrid_decaps(scheme, _p(sk), _p(ct), _p(ss), _p(st), n, device), "hybrid decaps")
    return ss, st


class CallQueue:
    """circl_hip_queue: the asynchronous form of the entry points that take their keys WITH the call (a TLS 1.3 server encapsulates to the client's
    ephemeral key).  op: "mlkem-encaps", "mlkem-decaps" (param 512 / 768 / 1024), "hybrid-encaps", "hybrid-decaps" (param = the hybrid scheme id)."""
    OPS = {"mlkem-encaps": 1, "mlkem-decaps": 2, "hybrid-encaps": 3, "hybrid-decaps": 4}

    def __init__(self, op, param, max_items, device=0, eventfd=False):
        import ctypes as C
        self.op, self.param = op, param
        self.handle = C.c_void_p()
        nat.check(nat.lib().circl_hip_queue_open(self.OPS[op], param, device, max_items, 1 if eventfd else 0, C.byref(self.handle)), "queue_open")

    def submit(self, key, inp, out0, ss, st):
        """(rc, ticket); key / inp: C-contiguous uint8 rows (copied before the call returns), out0 (None for a decapsulation) / ss / st: filled when the ticket is done"""
        import ctypes as C
        key, inp = np.ascontiguousarray(key, np.uint8), np.ascontiguousarray(inp, np.uint8)
        n = len(inp)
        t = C.c_uint64()
        rc = nat.lib().circl_hip_queue_submit(self.handle, _p(key), _p(inp), None if out0 is None else _p(out0), _p(ss), None if st is None else _p(st), n, C.byref(t))
        return rc, t.value

    def poll(self, tickets):
        t = np.asarray(tickets, np.uint64)
        st = np.zeros(len(t), np.int8)
        nat.lib().circl_hip_queue_poll(self.handle, _p(t), len(t), _p(st))
        return [int(x) for x in st]

    def wait(self, ticket, timeout_us=-1):
        return nat.lib().circl_hip_queue_wait(self.handle, int(ticket), int(timeout_us))

    def eventfd(self):
        return nat.lib().circl_hip_queue_eventfd(self.handle)

    def stats(self):
        import ctypes as C
        c, i, l = C.c_uint64(), C.c_uint64(), C.c_uint64()
        nat.check(nat.lib().circl_hip_queue_stats(self.handle, C.byref(c), C.byref(i), C.byref(l)), "queue_stats")
        return c.value, i.value, l.value

    def close(self):
        """CIRCL_HIP_OK (the queue is gone) or CIRCL_HIP_EBUSY"""
        rc = nat.lib().circl_hip_queue_close(self.handle)
        if rc == 0:
            self.handle = None
        return rc
