"""Worker of tests/test_gpu_round3.py::test_kem_batch_routes: ML-KEM encapsulation / decapsulation (per-item keys and one key)
against the oracle in its own process, so that the environment can force each batch route at small sizes: the hashing
wavefronts two sponges per wavefront (CIRCL_HIP_KEM_COOP), a sponge per lane pair (CIRCL_HIP_KEM_SPLIT), a sponge per lane;
the small-batch routes on or off (CIRCL_HIP_KEM_SMALL, CIRCL_HIP_KEM_SMALL_SHARED, CIRCL_HIP_KEM_SMALL_SHARED_DECAPS); the one-launch
form for up to 2^CIRCL_HIP_KEM_CHAIN_ITEM items (mlkem_*_chain_kernel<K, false>: two / four wavefronts per item; 512 by default).
    python tests/kem_routes_worker.py <param>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from circl_amd import hostapi  # noqa: E402
from oracle import orc  # noqa: E402

p = int(sys.argv[1])
# lane pairs: 32 items per wavefront; cooperative: 2; groups of G = 16 / 7 / 4 items; with CIRCL_HIP_KEM_SMALL_WGS=1 (256 ring-phase
# groups) 511 / 1000 / 1500 items are groups of 2 / 4 / 6: PRF streams on lane pairs while a group's streams fit 32 pairs, on lanes beyond
for n in (1, 2, 31, 33, 64, 65, 511, 512, 513, 1000, 1500, 2049, 2500):
    rng = np.random.default_rng(p * 131 + n)
    seeds = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    ek, dk = orc.mlkem_keygen(p, seeds)
    ek_g, dk_g = hostapi.mlkem_keygen(p, seeds)   # (one launch with two wavefronts per key up to 2^CIRCL_HIP_KEM_CHAIN_ITEM keys, three launches beyond)
    assert (ek_g == ek).all() and (dk_g == dk).all(), ("keygen", p, n)
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ct0, ss0, _ = orc.mlkem_encaps(p, ek, m)
    ct, ss, st = hostapi.mlkem_encaps(p, ek, m)
    assert (st == 0).all() and (ct == ct0).all() and (ss == ss0).all(), ("encaps", p, n)
    if n >= 31:  # a key with a coefficient >= q among them: kem.ErrPubKey for that item only (cpapke.go:45-55)
        ek_nc = ek.copy()
        ek_nc[n // 2, 0] = 0xff
        ek_nc[n // 2, 1] |= 0x0f
        ctn, ssn, stn = hostapi.mlkem_encaps(p, ek_nc, m)
        keep = np.arange(n) != n // 2
        assert stn[n // 2] == 1 and not ctn[n // 2].any() and not ssn[n // 2].any() and not stn[keep].any() and (ctn[keep] == ct0[keep]).all(), ("encaps, bad key", p, n)
    bad_ct = ct.copy()
    bad_ct[::3, 9] ^= 2  # implicit rejection for every third item
    bad_dk = dk.copy()
    bad_dk[1::4, -40] ^= 1  # stored H(ek) no longer matches: status 2, zero secret
    got, st = hostapi.mlkem_decaps(p, bad_dk, bad_ct)
    want, st0 = orc.mlkem_decaps(p, bad_dk, bad_ct)
    assert (st == st0).all() and (got == want).all(), ("decaps", p, n)
    assert (st[1::4] == 2).all() and not got[1::4].any() and (st[0::4] == 0).all()
    # one key for the batch
    ct1, ss1, st1 = hostapi.mlkem_encaps_shared(p, ek[:1], m)
    ct10, ss10, _ = orc.mlkem_encaps(p, np.tile(ek[:1], (n, 1)), m)
    assert (st1 == 0).all() and (ct1 == ct10).all() and (ss1 == ss10).all(), ("encaps, one key", p, n)
    ct1[::3, 9] ^= 2
    got, st = hostapi.mlkem_decaps_shared(p, dk[:1], ct1)
    want, _ = orc.mlkem_decaps(p, np.tile(dk[:1], (n, 1)), ct1)
    assert (st == 0).all() and (got == want).all(), ("decaps, one key", p, n)
    bk = dk[:1].copy()
    bk[0, -40] ^= 1
    got, st = hostapi.mlkem_decaps_shared(p, bk, ct1)
    assert (st == 2).all() and not got.any()
print("kem routes ok", p)
