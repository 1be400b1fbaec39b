"""NIST PQCgenKAT AES-256-CTR DRBG (reference: internal/nist/drbg.go:9-64), used only to replay
the KAT transcripts of kem/kyber/kat_test.go.  AES-256 single-block encryption comes from the
system libcrypto (OpenSSL 3) through ctypes."""
import ctypes as C
import ctypes.util

_crypto = None


def _lib():
    global _crypto
    if _crypto is None:
        name = ctypes.util.find_library("crypto") or "libcrypto.so.3"
        _crypto = C.CDLL(name)
        _crypto.EVP_CIPHER_CTX_new.restype = C.c_void_p
        _crypto.EVP_aes_256_ecb.restype = C.c_void_p
    return _crypto


def aes256_block(key, block):
    L = _lib()
    ctx = C.c_void_p(L.EVP_CIPHER_CTX_new())
    try:
        assert L.EVP_EncryptInit_ex(ctx, C.c_void_p(L.EVP_aes_256_ecb()), None, key, None) == 1
        L.EVP_CIPHER_CTX_set_padding(ctx, 0)
        out = C.create_string_buffer(32)
        n = C.c_int(0)
        assert L.EVP_EncryptUpdate(ctx, out, C.byref(n), block, 16) == 1
        return out.raw[:16]
    finally:
        L.EVP_CIPHER_CTX_free(ctx)


class DRBG:
    def __init__(self, seed48):
        self.key = bytes(32)
        self.v = bytes(16)
        self._update(seed48)

    def _inc(self):
        self.v = ((int.from_bytes(self.v, "big") + 1) % (1 << 128)).to_bytes(16, "big")

    def _update(self, pd):
        buf = b""
        for _ in range(3):
            self._inc()
            buf += aes256_block(self.key, self.v)
        if pd is not None:
            buf = bytes(a ^ b for a, b in zip(buf, pd))
        self.key, self.v = buf[:32], buf[32:]

    def fill(self, n):
        out = b""
        while len(out) < n:
            self._inc()
            out += aes256_block(self.key, self.v)
        self._update(None)
        return out[:n]
