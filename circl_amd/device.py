"""Device-resident entry points for callers that already hold torch CUDA tensors (bench.py, smoke).

torch is plumbing here: it owns the HBM allocations and the stream; the work is done by the
*_dev functions of the C ABI, launched on torch's current stream.
"""
import ctypes as C

import torch

from . import _native as nat

KEM_SIZES = {512: (800, 1632, 768), 768: (1184, 2400, 1088), 1024: (1568, 3168, 1568)}  # ek, dk, ct
DSA_SIZES = {44: (1312, 2560, 2420), 65: (1952, 4032, 3309), 87: (2592, 4896, 4627),  # pk, sk, sig
             2: (1312, 2528, 2420), 3: (1952, 4000, 3293), 5: (2592, 4864, 4595)}
KERNELS = {"mlkem_hash": 0, "mlkem_encrypt": 1, "mlkem_decrypt": 2, "mlkem_keygen": 3, "mlkem_finish": 4,
           "mldsa_hash": 5, "mldsa_verify": 6, "mldsa_keygen": 7, "mldsa_sign": 8, "mlkem_keytable": 9, "mldsa_keytable": 10, "x25519": 11}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, cols=None):
    assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous(), "need a contiguous uint8 CUDA tensor"
    if cols is not None:
        assert t.shape[-1] == cols, (t.shape, cols)
    return t.data_ptr()


class MLKEMDevice:
    """Holds the output / workspace tensors for one parameter set and batch size."""

    def __init__(self, param, n, device="cuda"):
        self.param, self.n = param, n
        self.EK, self.DK, self.CT = KEM_SIZES[param]
        self.L = nat.lib()
        self.wsb = self.L.circl_hip_mlkem_workspace_size(param, n)
        self.ws = torch.empty(max(self.wsb, 256), dtype=torch.uint8, device=device)
        self.ct = torch.empty((n, self.CT), dtype=torch.uint8, device=device)
        self.ss = torch.empty((n, 32), dtype=torch.uint8, device=device)
        self.status = torch.empty(n, dtype=torch.uint8, device=device)

    def encaps(self, ek, m, ct=None, ss=None, status=None):
        ct = self.ct if ct is None else ct
        ss = self.ss if ss is None else ss
        status = self.status if status is None else status
        rc = self.L.circl_hip_mlkem_encaps_dev(self.param, _chk(ek, self.EK), _chk(m, 32), _chk(ct, self.CT), _chk(ss, 32),
                                               _chk(status), self.n, self.ws.data_ptr(), self.wsb, _stream())
        nat.check(rc, "mlkem_encaps_dev")
        return ct, ss, status

    def encaps_shared(self, ek1, m, ct=None, ss=None, status=None):
        """every item encapsulates to the one key ek1 (a single row)"""
        ct = self.ct if ct is None else ct
        ss = self.ss if ss is None else ss
        status = self.status if status is None else status
        rc = self.L.circl_hip_mlkem_encaps_shared_dev(self.param, _chk(ek1, self.EK), _chk(m, 32), _chk(ct, self.CT), _chk(ss, 32),
                                                      _chk(status), self.n, self.ws.data_ptr(), self.wsb, _stream())
        nat.check(rc, "mlkem_encaps_shared_dev")
        return ct, ss, status

    def encaps_table(self, table, m, key_idx=None, ct=None, ss=None, status=None):
        """through a resident key table (hostapi.KeyTable of kind "mlkem-public"); key_idx None: entry 0 for every item"""
        ct = self.ct if ct is None else ct
        ss = self.ss if ss is None else ss
        status = self.status if status is None else status
        rc = self.L.circl_hip_mlkem_encaps_table_dev(table.handle, None if key_idx is None else key_idx.data_ptr(), _chk(m, 32), _chk(ct, self.CT),
                                                     _chk(ss, 32), _chk(status), self.n, self.ws.data_ptr(), self.wsb, _stream())
        nat.check(rc, "mlkem_encaps_table_dev")
        return ct, ss, status

    def decaps_table(self, table, ct, key_idx=None, ss=None, status=None):
        ss = self.ss if ss is None else ss
        status = self.status if status is None else status
        rc = self.L.circl_hip_mlkem_decaps_table_dev(table.handle, None if key_idx is None else key_idx.data_ptr(), _chk(ct, self.CT), _chk(ss, 32),
                                                     _chk(status), self.n, self.ws.data_ptr(), self.wsb, _stream())
        nat.check(rc, "mlkem_decaps_table_dev")
        return ss, status

    def decaps(self, dk, ct, ss=None, status=None):
        ss = self.ss if ss is None else ss
        status = self.status if status is None else status
        rc = self.L.circl_hip_mlkem_decaps_dev(self.param, _chk(dk, self.DK), _chk(ct, self.CT), _chk(ss, 32), _chk(status),
                                               self.n, self.ws.data_ptr(), self.wsb, _stream())
        nat.check(rc, "mlkem_decaps_dev")
        return ss, status

    def keygen(self, seeds, ek=None, dk=None):
        if ek is None:
            ek = torch.empty((self.n, self.EK), dtype=torch.uint8, device=seeds.device)
        if dk is None:
            dk = torch.empty((self.n, self.DK), dtype=torch.uint8, device=seeds.device)
        rc = self.L.circl_hip_mlkem_keygen_dev(self.param, _chk(seeds, 64), _chk(ek, self.EK), _chk(dk, self.DK), self.n,
                                               self.ws.data_ptr(), self.wsb, _stream())
        nat.check(rc, "mlkem_keygen_dev")
        return ek, dk


    def encaps_keyed(self, ek_table, key_idx, m, ct=None, ss=None, status=None):
        """item i encapsulates to row key_idx[i] (int32 / uint32 tensor) of ek_table"""
        ct = self.ct if ct is None else ct
        ss = self.ss if ss is None else ss
        status = self.status if status is None else status
        nkeys = ek_table.shape[0]
        wsb = self.L.circl_hip_mlkem_keyed_workspace_size(self.param, self.n, nkeys)
        if getattr(self, "_kws", None) is None or self._kws.numel() < wsb:
            self._kws = torch.empty(wsb, dtype=torch.uint8, device=ek_table.device)
        assert key_idx.is_cuda and key_idx.dtype in (torch.int32, torch.uint32) and key_idx.is_contiguous() and key_idx.numel() == self.n
        rc = self.L.circl_hip_mlkem_encaps_keyed_dev(self.param, _chk(ek_table, self.EK), nkeys, key_idx.data_ptr(), _chk(m, 32), _chk(ct, self.CT),
                                                     _chk(ss, 32), _chk(status), self.n, self._kws.data_ptr(), wsb, _stream())
        nat.check(rc, "mlkem_encaps_keyed_dev")
        return ct, ss, status

    def decaps_keyed(self, dk_table, key_idx, ct, ss=None, status=None):
        ss = self.ss if ss is None else ss
        status = self.status if status is None else status
        nkeys = dk_table.shape[0]
        wsb = self.L.circl_hip_mlkem_keyed_workspace_size(self.param, self.n, nkeys)
        if getattr(self, "_kws", None) is None or self._kws.numel() < wsb:
            self._kws = torch.empty(wsb, dtype=torch.uint8, device=dk_table.device)
        assert key_idx.is_cuda and key_idx.dtype in (torch.int32, torch.uint32) and key_idx.is_contiguous() and key_idx.numel() == self.n
        rc = self.L.circl_hip_mlkem_decaps_keyed_dev(self.param, _chk(dk_table, self.DK), nkeys, key_idx.data_ptr(), _chk(ct, self.CT), _chk(ss, 32),
                                                     _chk(status), self.n, self._kws.data_ptr(), wsb, _stream())
        nat.check(rc, "mlkem_decaps_keyed_dev")
        return ss, status

    def decaps_shared(self, dk1, ct, ss=None, status=None):
        ss = self.ss if ss is None else ss
        status = self.status if status is None else status
        rc = self.L.circl_hip_mlkem_decaps_shared_dev(self.param, _chk(dk1, self.DK), _chk(ct, self.CT), _chk(ss, 32), _chk(status), self.n,
                                                      self.ws.data_ptr(), self.wsb, _stream())
        nat.check(rc, "mlkem_decaps_shared_dev")
        return ss, status


class MLDSADevice:
    """Device-resident ML-DSA batch of n items with fixed-length messages (msg_len bytes each) and no contexts:
    the shape of BASELINE.json's ML-DSA configs.  Holds the workspaces and the offset array."""

    def __init__(self, param, n, device="cuda", msg_len=32, nkeys=0, sign=False):
        self.param, self.n, self.msg_len = param, n, msg_len
        self.PK, self.SK, self.SIG = DSA_SIZES[param]
        self.L = nat.lib()
        self.wsb = max(self.L.circl_hip_mldsa_workspace_size(param, n),
                       self.L.circl_hip_mldsa_keyed_workspace_size(param, n, nkeys) if nkeys else 0)
        self.ws = torch.empty(max(self.wsb, 256), dtype=torch.uint8, device=device)
        self.off = torch.arange(0, msg_len * (n + 1), msg_len, dtype=torch.int64, device=device)
        self.ok = torch.empty(n, dtype=torch.uint8, device=device)
        self.sws = None
        if sign:
            self.swsb = self.L.circl_hip_mldsa_sign_workspace_size(param, n)
            self.sws = torch.empty(self.swsb, dtype=torch.uint8, device=device)
            self.rnd0 = torch.zeros((n, 32), dtype=torch.uint8, device=device)

    def _msg(self, msg):
        assert msg.is_cuda and msg.dtype == torch.uint8 and msg.is_contiguous() and msg.numel() >= self.n * self.msg_len + 4, \
            "messages: contiguous uint8 CUDA tensor with >= 4 bytes of slack behind the last one"
        return msg.data_ptr()

    def keygen(self, seeds, pk=None, sk=None):
        pk = torch.empty((self.n, self.PK), dtype=torch.uint8, device=seeds.device) if pk is None else pk
        sk = torch.empty((self.n, self.SK), dtype=torch.uint8, device=seeds.device) if sk is None else sk
        rc = self.L.circl_hip_mldsa_keygen_dev(self.param, _chk(seeds, 32), _chk(pk, self.PK), _chk(sk, self.SK), self.n, self.ws.data_ptr(), self.wsb,
                                               _stream())
        nat.check(rc, "mldsa_keygen_dev")
        return pk, sk

    def sign(self, sk, msg, sig=None, rnd=None, shared=False):
        """deterministic unless rnd (n, 32) is given; asynchronous on the current stream"""
        assert self.sws is not None, "construct with sign=True"
        sig = torch.empty(self.n * self.SIG + 16, dtype=torch.uint8, device=sk.device)[:self.n * self.SIG].view(self.n, self.SIG) if sig is None else sig
        rnd = self.rnd0 if rnd is None else rnd
        fn = self.L.circl_hip_mldsa_sign_shared_dev if shared else self.L.circl_hip_mldsa_sign_dev
        rc = fn(self.param, _chk(sk, self.SK), self._msg(msg), self.off.data_ptr(), None, None, _chk(rnd, 32), 0, _chk(sig, self.SIG), self.n,
                self.sws.data_ptr(), self.swsb, _stream())
        nat.check(rc, "mldsa_sign_dev")
        return sig

    def sign_table(self, table, msg, sig=None, rnd=None):
        """with a private key prepared once (hostapi.KeyTable of kind "mldsa-private")"""
        assert self.sws is not None, "construct with sign=True"
        sig = torch.empty(self.n * self.SIG + 16, dtype=torch.uint8, device=msg.device)[:self.n * self.SIG].view(self.n, self.SIG) if sig is None else sig
        rnd = self.rnd0 if rnd is None else rnd
        rc = self.L.circl_hip_mldsa_sign_table_dev(table.handle, self._msg(msg), self.off.data_ptr(), None, None, _chk(rnd, 32), 0, _chk(sig, self.SIG), self.n,
                                                   self.sws.data_ptr(), self.swsb, _stream())
        nat.check(rc, "mldsa_sign_table_dev")
        return sig

    def verify(self, pk, sig, msg, ok=None):
        ok = self.ok if ok is None else ok
        rc = self.L.circl_hip_mldsa_verify_dev(self.param, _chk(pk, self.PK), _chk(sig, self.SIG), self._msg(msg), self.off.data_ptr(), None, None,
                                               _chk(ok), self.n, self.ws.data_ptr(), self.wsb, _stream())
        nat.check(rc, "mldsa_verify_dev")
        return ok

    def verify_table(self, table, sig, msg, key_idx=None, ok=None):
        """through a resident key table (hostapi.KeyTable of kind "mldsa-public"); key_idx None: entry 0 for every item"""
        ok = self.ok if ok is None else ok
        rc = self.L.circl_hip_mldsa_verify_table_dev(table.handle, None if key_idx is None else key_idx.data_ptr(), _chk(sig, self.SIG), self._msg(msg),
                                                     self.off.data_ptr(), None, None, _chk(ok), self.n, self.ws.data_ptr(), self.wsb, _stream())
        nat.check(rc, "mldsa_verify_table_dev")
        return ok

    def verify_shared(self, pk1, sig, msg, ok=None):
        ok = self.ok if ok is None else ok
        rc = self.L.circl_hip_mldsa_verify_shared_dev(self.param, _chk(pk1, self.PK), _chk(sig, self.SIG), self._msg(msg), self.off.data_ptr(), None, None,
                                                      _chk(ok), self.n, self.ws.data_ptr(), self.wsb, _stream())
        nat.check(rc, "mldsa_verify_shared_dev")
        return ok

    def verify_keyed(self, pk_table, key_idx, sig, msg, ok=None):
        ok = self.ok if ok is None else ok
        nkeys = pk_table.shape[0]
        assert self.L.circl_hip_mldsa_keyed_workspace_size(self.param, self.n, nkeys) <= self.wsb, "construct with nkeys="
        assert key_idx.is_cuda and key_idx.dtype in (torch.int32, torch.uint32) and key_idx.is_contiguous() and key_idx.numel() == self.n
        rc = self.L.circl_hip_mldsa_verify_keyed_dev(self.param, _chk(pk_table, self.PK), nkeys, key_idx.data_ptr(), _chk(sig, self.SIG), self._msg(msg),
                                                     self.off.data_ptr(), None, None, _chk(ok), self.n, self.ws.data_ptr(), self.wsb, _stream())
        nat.check(rc, "mldsa_verify_keyed_dev")
        return ok


def profile_enable(on=True):
    nat.check(nat.lib().circl_hip_profile_enable(int(on)), "profile_enable")


def profile_read(kernel):
    """-> (total_ms, launches) since the last read, for one kernel name of KERNELS."""
    ms, cnt = C.c_double(0), C.c_uint64(0)
    nat.check(nat.lib().circl_hip_profile_read(KERNELS[kernel], C.byref(ms), C.byref(cnt)), "profile_read")
    return ms.value, cnt.value


def valu_probe(device=0, waves_per_simd=4):
    """circl_hip_profile_valu_probe -> (Keccak-round, two-operand integer) wave-instructions per second per SIMD, measured now"""
    k, s = C.c_double(0), C.c_double(0)
    nat.check(nat.lib().circl_hip_profile_valu_probe(device, waves_per_simd, C.byref(k), C.byref(s)), "profile_valu_probe")
    return k.value, s.value


XWING, X25519MLKEM768, KYBER768_X25519, KYBER512_X25519 = 1, 2, 3, 4


class HybridDevice:
    """X-Wing / X25519MLKEM768 on resident tensors (circl_hip_hybrid_*_dev); x25519() is the bare ladder batch."""

    def __init__(self, scheme, n, device="cuda"):
        self.scheme, self.n = scheme, n
        self.L = nat.lib()
        self.S = {k: getattr(self.L, "circl_hip_hybrid_%s_size" % k)(scheme) for k in ("seed", "eseed", "pk", "sk", "ct", "ss")}
        self.wsb = self.L.circl_hip_hybrid_workspace_size(scheme, n)
        self.ws = torch.empty(self.wsb, dtype=torch.uint8, device=device)
        self.pk = torch.empty((n, self.S["pk"]), dtype=torch.uint8, device=device)
        self.sk = torch.empty((n, self.S["sk"]), dtype=torch.uint8, device=device)
        self.ct = torch.empty((n, self.S["ct"]), dtype=torch.uint8, device=device)
        self.ss = torch.empty((n, self.S["ss"]), dtype=torch.uint8, device=device)
        self.ss2 = torch.empty((n, self.S["ss"]), dtype=torch.uint8, device=device)
        self.status = torch.empty(n, dtype=torch.uint8, device=device)

    def keygen(self, seeds):
        nat.check(self.L.circl_hip_hybrid_keygen_dev(self.scheme, _chk(seeds, self.S["seed"]), _chk(self.pk), _chk(self.sk), self.n,
                                                     self.ws.data_ptr(), self.wsb, _stream()), "hybrid_keygen_dev")
        return self.pk, self.sk

    def encaps(self, pk, eseeds):
        nat.check(self.L.circl_hip_hybrid_encaps_dev(self.scheme, _chk(pk, self.S["pk"]), _chk(eseeds, self.S["eseed"]), _chk(self.ct), _chk(self.ss),
                                                     _chk(self.status), self.n, self.ws.data_ptr(), self.wsb, _stream()), "hybrid_encaps_dev")
        return self.ct, self.ss, self.status

    def decaps(self, sk, ct):
        nat.check(self.L.circl_hip_hybrid_decaps_dev(self.scheme, _chk(sk, self.S["sk"]), _chk(ct, self.S["ct"]), _chk(self.ss2), _chk(self.status),
                                                     self.n, self.ws.data_ptr(), self.wsb, _stream()), "hybrid_decaps_dev")
        return self.ss2, self.status


def x25519(scalar, point=None, out=None, ok=None):
    n = scalar.shape[0]
    out = torch.empty_like(scalar) if out is None else out
    ok = torch.empty(n, dtype=torch.uint8, device=scalar.device) if ok is None else ok
    nat.check(nat.lib().circl_hip_x25519_dev(_chk(scalar, 32), None if point is None else _chk(point, 32), _chk(out, 32), _chk(ok), n, _stream()),
              "x25519_dev")
    return out, ok
