import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import hostapi
from oracle import orc
param, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(5)
pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (n, 32), dtype=np.uint8))
msgs = [bytes(rng.integers(0, 256, 40, dtype=np.uint8)) for _ in range(n)]
sig = hostapi.mldsa_sign(param, sk, msgs)
want = orc.mldsa_sign(param, sk, msgs)
ok = hostapi.mldsa_verify(param, pk, sig, msgs)
bad = np.nonzero((sig != want).any(axis=1))[0]
CT = {44: 32, 65: 48, 87: 64}[param]
print("pair", os.environ.get("CIRCL_HIP_SIGN_PAIR"), "spec", os.environ.get("CIRCL_HIP_SIGN_SPEC"), "param", param, "n", n, "mismatch", len(bad), "invalid", int((ok == 0).sum()),
      "ctilde differs", int((sig[bad][:, :CT] != want[bad][:, :CT]).any(axis=1).sum()) if len(bad) else 0, "first bad", bad[:8].tolist())
