"""ctypes binding of libcirclhip.so -- the C ABI of include/circl_hip.h.

Loading fails loudly when the library is missing: there is no fallback path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcirclhip.so")

# every symbol include/circl_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "circl_hip_init", "circl_hip_device_count", "circl_hip_physical_device", "circl_hip_last_error", "circl_hip_version", "circl_hip_device_info",
    "circl_hip_mlkem_encaps_keyed", "circl_hip_mlkem_decaps_keyed", "circl_hip_mlkem_keyed_workspace_size",
    "circl_hip_mlkem_encaps_keyed_dev", "circl_hip_mlkem_decaps_keyed_dev",
    "circl_hip_mldsa_verify_keyed", "circl_hip_mldsa_keyed_workspace_size", "circl_hip_mldsa_verify_keyed_dev",
    "circl_hip_mlkem_keytable_new", "circl_hip_mldsa_keytable_new", "circl_hip_keytable_free", "circl_hip_keytable_set_coalesce", "circl_hip_set_coalesce", "circl_hip_keytable_coalesce_stats", "circl_hip_profile_valu_probe", "circl_hip_host_pool_stats", "circl_hip_mlkem_encaps_table", "circl_hip_mlkem_decaps_table",
    "circl_hip_mldsa_verify_table", "circl_hip_mldsa_privkey_new", "circl_hip_mldsa_sign_table", "circl_hip_mldsa_sign_table_dev", "circl_hip_mlkem_encaps_table_dev", "circl_hip_mlkem_decaps_table_dev", "circl_hip_mldsa_verify_table_dev",
    "circl_hip_mlkem_ek_size", "circl_hip_mlkem_dk_size", "circl_hip_mlkem_ct_size",
    "circl_hip_mldsa_pk_size", "circl_hip_mldsa_sig_size", "circl_hip_mldsa_sk_size",
    "circl_hip_mldsa_keygen", "circl_hip_mldsa_keygen_dev",
    "circl_hip_mldsa_sign", "circl_hip_mldsa_sign_shared", "circl_hip_mldsa_sign_shared_dev", "circl_hip_mldsa_sign_internal", "circl_hip_mldsa_sign_workspace_size", "circl_hip_mldsa_sign_dev",
    "circl_hip_mlkem_encaps", "circl_hip_mlkem_decaps", "circl_hip_mlkem_keygen",
    "circl_hip_mlkem_workspace_size", "circl_hip_mlkem_encaps_dev", "circl_hip_mlkem_decaps_dev",
    "circl_hip_mlkem_keygen_dev",
    "circl_hip_mlkem_encaps_shared", "circl_hip_mlkem_encaps_shared_dev", "circl_hip_mlkem_decaps_shared", "circl_hip_mlkem_decaps_shared_dev",
    "circl_hip_kyber_keygen", "circl_hip_kyber_encaps", "circl_hip_kyber_decaps",
    "circl_hip_kyber_keygen_dev", "circl_hip_kyber_encaps_dev", "circl_hip_kyber_decaps_dev",
    "circl_hip_mldsa_verify", "circl_hip_mldsa_verify_shared", "circl_hip_mldsa_verify_shared_dev", "circl_hip_mldsa_verify_internal", "circl_hip_mldsa_workspace_size", "circl_hip_mldsa_verify_dev",
    "circl_hip_keccak_f1600", "circl_hip_keccak_f1600_coop", "circl_hip_keccak_f1600_split", "circl_hip_mldsa_sample_in_ball", "circl_hip_kyber_ntt", "circl_hip_kyber_mulhat", "circl_hip_lane_op", "circl_hip_kyber_sample_uniform", "circl_hip_kyber_sample_cbd", "circl_hip_mldsa_sample_uniform", "circl_hip_dilithium_ntt",
    "circl_hip_shake", "circl_hip_xof", "circl_hip_k12", "circl_hip_x25519", "circl_hip_x25519_dev",
    "circl_hip_hybrid_seed_size", "circl_hip_hybrid_eseed_size", "circl_hip_hybrid_pk_size", "circl_hip_hybrid_sk_size", "circl_hip_hybrid_ct_size",
    "circl_hip_hybrid_ss_size", "circl_hip_hybrid_workspace_size", "circl_hip_hybrid_keygen", "circl_hip_hybrid_encaps", "circl_hip_hybrid_decaps",
    "circl_hip_hybrid_keygen_dev", "circl_hip_hybrid_encaps_dev", "circl_hip_hybrid_decaps_dev", "circl_hip_alloc_host", "circl_hip_free_host",
    "circl_hip_profile_enable", "circl_hip_profile_read",
    "circl_hip_keytable_device", "circl_hip_keytable_nkeys", "circl_hip_keytable_on_device",
    "circl_hip_mldsa_privkeys_new", "circl_hip_mldsa_sign_table_keyed", "circl_hip_mldsa_sign_table_keyed_dev",
    "circl_hip_mlkem_public_from_private", "circl_hip_mldsa_public_from_private", "circl_hip_mldsa_public_from_private_dev",
    "circl_hip_hybrid_keytable_new", "circl_hip_hybrid_encaps_table", "circl_hip_hybrid_decaps_table",
    "circl_hip_hybrid_encaps_table_dev", "circl_hip_hybrid_decaps_table_dev",
    "circl_hip_keytable_close", "circl_hip_keytable_async_start", "circl_hip_keytable_async_stop", "circl_hip_keytable_eventfd",
    "circl_hip_mlkem_encaps_table_submit", "circl_hip_mlkem_decaps_table_submit", "circl_hip_mldsa_verify_table_submit",
    "circl_hip_hybrid_encaps_table_submit", "circl_hip_hybrid_decaps_table_submit",
    "circl_hip_queue_open", "circl_hip_queue_close", "circl_hip_queue_eventfd", "circl_hip_queue_stats", "circl_hip_queue_submit", "circl_hip_queue_poll",
    "circl_hip_queue_wait",
    "circl_hip_poll", "circl_hip_wait", "circl_hip_profile_call_stamps",
]

OK, EPARAM, ENODEV, EHIP, ENOMEM, EWORKSPACE, EBUSY, EAGAIN = 0, -1, -2, -3, -4, -5, -6, -7
ALL_DEVICES = -1


class CirclHipError(RuntimeError):
    def __init__(self, code, where, detail=""):
        names = {EPARAM: "EPARAM", ENODEV: "ENODEV", EHIP: "EHIP", ENOMEM: "ENOMEM", EWORKSPACE: "EWORKSPACE", EBUSY: "EBUSY", EAGAIN: "EAGAIN"}
        super().__init__(f"{where}: {names.get(code, code)} {detail}".strip())
        self.code = code


_lib = None


def _preload_hip_runtime():
    """libcirclhip.so needs libamdhip64.so.7.  When PyTorch-ROCm is installed it bundles its own copy
    of that runtime (same soname, requested by torch under the un-versioned file name), and two HIP
    runtimes in one process do not work.  Loading torch's copy first -- by path, without importing
    torch -- makes both libcirclhip.so and a later `import torch` share ONE runtime, whatever the
    import order.  Without torch the system ROCm runtime is used."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Returns the loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m circl_amd.build` "
                "(circl_amd has no CPU fallback)")
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for s in ("circl_hip_mlkem_ek_size", "circl_hip_mlkem_dk_size", "circl_hip_mlkem_ct_size",
                  "circl_hip_mldsa_pk_size", "circl_hip_mldsa_sig_size", "circl_hip_mldsa_sk_size"):
            getattr(L, s).restype = C.c_size_t
            getattr(L, s).argtypes = [C.c_int]
        for s in ("circl_hip_mlkem_workspace_size", "circl_hip_mldsa_workspace_size", "circl_hip_mldsa_sign_workspace_size"):
            getattr(L, s).restype = C.c_size_t
            getattr(L, s).argtypes = [C.c_int, C.c_size_t]
        for s in ("circl_hip_mlkem_keyed_workspace_size", "circl_hip_mldsa_keyed_workspace_size"):
            getattr(L, s).restype = C.c_size_t
            getattr(L, s).argtypes = [C.c_int, C.c_size_t, C.c_size_t]
        L.circl_hip_last_error.restype = C.c_char_p
        L.circl_hip_version.restype = C.c_char_p
        L.circl_hip_alloc_host.restype = C.c_void_p
        L.circl_hip_alloc_host.argtypes = [C.c_size_t]
        L.circl_hip_free_host.argtypes = [C.c_void_p]
        vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
        L.circl_hip_mlkem_encaps.argtypes = [i, vp, vp, vp, vp, vp, sz, i]
        L.circl_hip_mlkem_decaps.argtypes = [i, vp, vp, vp, vp, sz, i]
        L.circl_hip_mlkem_keygen.argtypes = [i, vp, vp, vp, sz, i]
        L.circl_hip_mlkem_encaps_dev.argtypes = [i, vp, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_mlkem_decaps_dev.argtypes = [i, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_mlkem_keygen_dev.argtypes = [i, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_mlkem_encaps_shared.argtypes = [i, vp, vp, vp, vp, vp, sz, i]
        L.circl_hip_mlkem_encaps_shared_dev.argtypes = [i, vp, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_mlkem_decaps_shared.argtypes = [i, vp, vp, vp, vp, sz, i]
        L.circl_hip_mlkem_decaps_shared_dev.argtypes = [i, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_device_info.argtypes = [i, vp, vp]
        L.circl_hip_mlkem_encaps_keyed.argtypes = [i, vp, sz, vp, vp, vp, vp, vp, sz, i]
        L.circl_hip_mlkem_decaps_keyed.argtypes = [i, vp, sz, vp, vp, vp, vp, sz, i]
        L.circl_hip_mlkem_encaps_keyed_dev.argtypes = [i, vp, sz, vp, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_mlkem_decaps_keyed_dev.argtypes = [i, vp, sz, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_mldsa_verify_keyed.argtypes = [i, vp, sz, vp, vp, vp, vp, vp, vp, vp, sz, i]
        L.circl_hip_mldsa_verify_keyed_dev.argtypes = [i, vp, sz, vp, vp, vp, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_mlkem_keytable_new.argtypes = [i, i, vp, sz, i, vp, C.POINTER(vp)]
        L.circl_hip_mldsa_keytable_new.argtypes = [i, vp, sz, i, C.POINTER(vp)]
        L.circl_hip_keytable_free.argtypes = [vp]
        L.circl_hip_keytable_free.restype = None
        L.circl_hip_keytable_set_coalesce.argtypes = [vp, sz, C.c_uint32]
        L.circl_hip_set_coalesce.argtypes = [sz, C.c_uint32]
        L.circl_hip_keytable_coalesce_stats.argtypes = [vp, vp, vp, vp]
        L.circl_hip_keytable_close.argtypes = [vp]
        L.circl_hip_keytable_async_start.argtypes = [vp, sz, C.c_uint32, i]
        L.circl_hip_keytable_async_stop.argtypes = [vp]
        L.circl_hip_keytable_eventfd.argtypes = [vp, i]
        L.circl_hip_mlkem_encaps_table_submit.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp]
        L.circl_hip_mlkem_decaps_table_submit.argtypes = [vp, vp, vp, vp, vp, sz, vp]
        L.circl_hip_mldsa_verify_table_submit.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        L.circl_hip_hybrid_encaps_table_submit.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp]
        L.circl_hip_hybrid_decaps_table_submit.argtypes = [vp, vp, vp, vp, vp, sz, vp]
        L.circl_hip_queue_open.argtypes = [i, i, i, sz, i, vp]
        L.circl_hip_queue_close.argtypes = [vp]
        L.circl_hip_queue_eventfd.argtypes = [vp]
        L.circl_hip_queue_stats.argtypes = [vp, vp, vp, vp]
        L.circl_hip_queue_submit.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp]
        L.circl_hip_queue_poll.argtypes = [vp, vp, sz, vp]
        L.circl_hip_queue_wait.argtypes = [vp, C.c_uint64, C.c_int64]
        L.circl_hip_poll.argtypes = [vp, vp, sz, vp]
        L.circl_hip_wait.argtypes = [vp, C.c_uint64, C.c_int64]
        L.circl_hip_profile_call_stamps.argtypes = [i, vp]
        L.circl_hip_mlkem_encaps_table.argtypes = [vp, vp, vp, vp, vp, vp, sz]
        L.circl_hip_mlkem_decaps_table.argtypes = [vp, vp, vp, vp, vp, sz]
        L.circl_hip_mldsa_verify_table.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz]
        L.circl_hip_mlkem_encaps_table_dev.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_mlkem_decaps_table_dev.argtypes = [vp, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_mldsa_verify_table_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_mldsa_privkey_new.argtypes = [i, vp, i, C.POINTER(vp)]
        L.circl_hip_mldsa_sign_table.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz]
        L.circl_hip_mldsa_sign_table_dev.argtypes = [vp, vp, vp, vp, vp, vp, i, vp, sz, vp, sz, vp]
        L.circl_hip_kyber_keygen.argtypes = [i, vp, vp, vp, sz, i]
        L.circl_hip_kyber_encaps.argtypes = [i, vp, vp, vp, vp, sz, i]
        L.circl_hip_kyber_decaps.argtypes = [i, vp, vp, vp, sz, i]
        L.circl_hip_kyber_keygen_dev.argtypes = [i, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_kyber_encaps_dev.argtypes = [i, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_kyber_decaps_dev.argtypes = [i, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_mldsa_verify.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, sz, i]
        L.circl_hip_mldsa_verify_shared.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, sz, i]
        L.circl_hip_mldsa_verify_shared_dev.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_mldsa_sign.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, sz, i]
        L.circl_hip_mldsa_sign_shared.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, sz, i]
        L.circl_hip_mldsa_sign_shared_dev.argtypes = [i, vp, vp, vp, vp, vp, vp, i, vp, sz, vp, sz, vp]
        L.circl_hip_mldsa_sign_internal.argtypes = [i, vp, vp, vp, vp, vp, sz, i]
        L.circl_hip_mldsa_sign_dev.argtypes = [i, vp, vp, vp, vp, vp, vp, i, vp, sz, vp, sz, vp]
        L.circl_hip_mldsa_keygen.argtypes = [i, vp, vp, vp, sz, i]
        L.circl_hip_mldsa_keygen_dev.argtypes = [i, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_mldsa_verify_internal.argtypes = [i, vp, vp, vp, vp, vp, sz, i]
        L.circl_hip_mldsa_verify_dev.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_keccak_f1600.argtypes = [vp, sz, i, i]
        L.circl_hip_keccak_f1600_coop.argtypes = [vp, sz, i]
        L.circl_hip_keccak_f1600_split.argtypes = [vp, sz, i]
        L.circl_hip_mldsa_sample_in_ball.argtypes = [i, vp, vp, sz, i, i]
        L.circl_hip_kyber_ntt.argtypes = [vp, sz, i, i]
        L.circl_hip_kyber_mulhat.argtypes = [vp, vp, vp, sz, i]
        L.circl_hip_dilithium_ntt.argtypes = [vp, sz, i, i]
        L.circl_hip_lane_op.argtypes = [i, i, vp, vp, vp, vp, sz, i]
        L.circl_hip_kyber_sample_uniform.argtypes = [vp, vp, vp, sz, i]
        L.circl_hip_kyber_sample_cbd.argtypes = [i, vp, vp, sz, i]
        L.circl_hip_mldsa_sample_uniform.argtypes = [vp, vp, vp, sz, i]
        L.circl_hip_shake.argtypes = [i, i, vp, sz, vp, sz, sz, i]
        L.circl_hip_xof.argtypes = [i, i, i, vp, vp, vp, sz, sz, i]
        L.circl_hip_k12.argtypes = [vp, vp, vp, vp, vp, sz, sz, i]
        L.circl_hip_x25519.argtypes = [vp, vp, vp, vp, sz, i]
        L.circl_hip_x25519_dev.argtypes = [vp, vp, vp, vp, sz, vp]
        for f in ("seed", "eseed", "pk", "sk", "ct", "ss"):
            fn = getattr(L, "circl_hip_hybrid_%s_size" % f)
            fn.restype, fn.argtypes = sz, [i]
        L.circl_hip_hybrid_workspace_size.restype = sz
        L.circl_hip_hybrid_workspace_size.argtypes = [i, sz]
        L.circl_hip_hybrid_keygen.argtypes = [i, vp, vp, vp, sz, i]
        L.circl_hip_hybrid_encaps.argtypes = [i, vp, vp, vp, vp, vp, sz, i]
        L.circl_hip_hybrid_decaps.argtypes = [i, vp, vp, vp, vp, sz, i]
        L.circl_hip_hybrid_keygen_dev.argtypes = [i, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_hybrid_encaps_dev.argtypes = [i, vp, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_hybrid_decaps_dev.argtypes = [i, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_keytable_device.argtypes = [vp]
        L.circl_hip_keytable_nkeys.argtypes = [vp]
        L.circl_hip_keytable_nkeys.restype = sz
        L.circl_hip_keytable_on_device.argtypes = [vp, i]
        L.circl_hip_keytable_on_device.restype = vp
        L.circl_hip_mldsa_privkeys_new.argtypes = [i, vp, sz, i, C.POINTER(vp)]
        L.circl_hip_mldsa_sign_table_keyed.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz]
        L.circl_hip_mldsa_sign_table_keyed_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, vp, sz, vp, sz, vp]
        L.circl_hip_mlkem_public_from_private.argtypes = [i, vp, vp, sz]
        L.circl_hip_mldsa_public_from_private.argtypes = [i, vp, vp, sz, i]
        L.circl_hip_mldsa_public_from_private_dev.argtypes = [i, vp, vp, sz, vp, sz, vp]
        L.circl_hip_hybrid_keytable_new.argtypes = [i, i, vp, sz, i, vp, C.POINTER(vp)]
        L.circl_hip_hybrid_encaps_table.argtypes = [vp, vp, vp, vp, vp, vp, sz]
        L.circl_hip_hybrid_decaps_table.argtypes = [vp, vp, vp, vp, vp, sz]
        L.circl_hip_hybrid_encaps_table_dev.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_hybrid_decaps_table_dev.argtypes = [vp, vp, vp, vp, vp, sz, vp, sz, vp]
        L.circl_hip_profile_enable.argtypes = [i]
        L.circl_hip_profile_read.argtypes = [i, vp, vp]
        L.circl_hip_profile_valu_probe.argtypes = [i, i, vp, vp]
        L.circl_hip_host_pool_stats.argtypes = [vp, vp, vp]
        _lib = L
    return _lib


def check(rc, where):
    if rc != 0:
        raise CirclHipError(rc, where, (lib().circl_hip_last_error() or b"").decode())
