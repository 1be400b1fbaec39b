"""Worker of tests/test_gpu_round3.py::test_sign_round_modes: batch signing against the oracle in its own process, so that
the environment can force the round modes at small batch sizes (CIRCL_HIP_SIGN_SPEC = list entries per CU below which rounds
speculate widely; above it the rounds run lazy pairs, or single attempts with CIRCL_HIP_SIGN_PAIR=0).
    python tests/sign_worker.py <param> <n> [shared]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from circl_amd import hostapi  # noqa: E402
from oracle import orc  # noqa: E402

param, n = int(sys.argv[1]), int(sys.argv[2])
shared = len(sys.argv) > 3 and sys.argv[3] == "shared"
rng = np.random.default_rng(param * 1000003 + n)
pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (1 if shared else n, 32), dtype=np.uint8))
msgs = [bytes(rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8)) for _ in range(n)]
r3 = param in (2, 3, 5)  # round-3 Dilithium: no context, deterministic (sign/dilithium/mode3)
ctxs = None if r3 else [bytes(rng.integers(0, 256, int(rng.integers(0, 30)), dtype=np.uint8)) for _ in range(n)]
rnd = None if r3 else rng.integers(0, 256, (n, 32), dtype=np.uint8)
if rnd is not None:
    rnd[::3] = 0
if shared:
    sig = hostapi.mldsa_sign_shared(param, sk, msgs, ctxs=ctxs, rnd=rnd)
    want = orc.mldsa_sign(param, np.tile(sk, (n, 1)), msgs, ctxs=ctxs, rnd=rnd)
    assert hostapi.mldsa_verify_shared(param, pk, sig, msgs, ctxs=ctxs).all()
else:
    sig = hostapi.mldsa_sign(param, sk, msgs, ctxs=ctxs, rnd=rnd)
    want = orc.mldsa_sign(param, sk, msgs, ctxs=ctxs, rnd=rnd)
    assert hostapi.mldsa_verify(param, pk, sig, msgs, ctxs=ctxs).all()
bad = np.nonzero((sig != want).any(axis=1))[0]
assert len(bad) == 0, ("signatures differ from the oracle's", bad[:10].tolist(), len(bad))
print("sign worker ok", param, n, "shared" if shared else "")
