"""Device-resident entry points for callers that already hold torch CUDA tensors (bench.py, smoke).

torch is plumbing here: it owns the HBM allocations and the stream; the work is done by the
*_dev functions of the C ABI, launched on torch's current stream.
"""
import ctypes as C

import torch

from . import _native as nat

KEM_SIZES = {512: (800, 1632, 768), 768: (1184, 2400, 1088), 1024: (1568, 3168, 1568)}  # ek, dk, ct
KERNELS = {"mlkem_hash": 0, "mlkem_encrypt": 1, "mlkem_decrypt": 2, "mlkem_keygen": 3, "mlkem_finish": 4,
           "mldsa_hash": 5, "mldsa_verify": 6}


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


def profile_enable(on=True):
    nat.check(nat.lib().circl_hip_profile_enable(int(on)), "profile_enable")


def profile_read(kernel):
    """-> (total_ms, launches) since the last read, for one kernel name of KERNELS."""
    ms, cnt = C.c_double(0), C.c_uint64(0)
    nat.check(nat.lib().circl_hip_profile_read(KERNELS[kernel], C.byref(ms), C.byref(cnt)), "profile_read")
    return ms.value, cnt.value
