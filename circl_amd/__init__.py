"""circl_amd -- MI355X-native batch ML-KEM / ML-DSA engine behind cloudflare/circl's
kem.Scheme / sign.Scheme interfaces.

The product is libcirclhip.so (hand-written HIP kernels for gfx950 + the C ABI of
include/circl_hip.h); the host-side mirrors of the reference's scheme interfaces are the C++
headers include/circl/{kem,sign,xwing,hybrid}.hpp and the Go packages under go/.  This Python
package is plumbing for tests and bench.py only:

    build     compiles csrc/ into libcirclhip.so for gfx950
    _native   ctypes binding of every symbol of include/circl_hip.h (fails loudly without the library)
    hostapi   numpy wrappers over the host-buffer entry points (what cgo would call)
    device    torch-tensor wrappers over the device-resident (*_dev) entry points
    parallel  one process per GPU: batch split, timing barrier and reductions (no data-path collective)
"""
from . import _native  # noqa: F401

__all__ = ["_native"]
