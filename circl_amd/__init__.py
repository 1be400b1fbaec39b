"""circl_amd -- MI355X-native batch ML-KEM / ML-DSA engine behind cloudflare/circl's
kem.Scheme / sign.Scheme interfaces.

The product is libcirclhip.so (hand-written HIP kernels for gfx950 + the C ABI of
include/circl_hip.h).  This Python package is plumbing for tests and bench.py: a ctypes binding
(`_native`), a host-side mirror of the reference's scheme interface (`kem`, `sign`, `schemes`)
and device-resident helpers that take torch tensors (`device`).
"""
from . import _native  # noqa: F401

__all__ = ["_native"]
