"""Worker of tests/test_gpu_round3.py: runs in its OWN process with CIRCL_HIP_LOGICAL_DEVICES=L, so that a one-GPU box
executes the host-side code of an L-GPU node: shard()'s threaded all-devices branch, L staging pools, L x (H2D, D2H,
2 compute) streams, L byte-mover pools.  SURVEY.md 8(e); shape of the work: kem/schemes/schemes_test.go:28-51.

    python tests/logical_worker.py parity | concurrent | config3 [log2_n]

Every sub-command prints one JSON line and exits non-zero on any mismatch.
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from circl_amd import _native as nat  # noqa: E402
from circl_amd import hostapi  # noqa: E402
from oracle import orc  # noqa: E402
from oracle import hybrid as ohyb  # noqa: E402

ALL = nat.ALL_DEVICES


def eq(a, b):
    if isinstance(a, tuple):
        return all(eq(x, y) for x, y in zip(a, b))
    return a.shape == b.shape and bool((a == b).all())


def expect_devices():
    L = nat.lib()
    want = int(os.environ["CIRCL_HIP_LOGICAL_DEVICES"])
    nd = L.circl_hip_device_count()
    assert nd == want, (nd, want)
    import ctypes as C
    phys = [L.circl_hip_physical_device(d) for d in range(nd)]
    nphys = max(phys) + 1
    assert phys == [d % nphys for d in range(nd)], phys
    assert L.circl_hip_physical_device(nd) == nat.ENODEV
    for d in range(nd):
        cus, numa = C.c_int(0), C.c_int(-2)
        assert L.circl_hip_device_info(d, C.byref(cus), C.byref(numa)) == 0 and cus.value >= 64
    return nd, nphys


def ragged(rng, n, lo=0, hi=300):
    return [bytes(rng.integers(0, 256, int(rng.integers(lo, hi)), dtype=np.uint8)) for _ in range(n)]


def parity():
    nd, nphys = expect_devices()
    rng = np.random.default_rng(20260924)
    report = {"logical_devices": nd, "hip_devices": nphys, "checks": 0}

    def both(fn, *a, **kw):
        r0 = fn(*a, device=0, **kw)
        r1 = fn(*a, device=ALL, **kw)
        assert eq(r0, r1), fn.__name__
        r2 = fn(*a, device=nd - 1, **kw)   # the last logical device alone (its own pool, streams, movers)
        assert eq(r0, r2), fn.__name__ + " on the last logical device"
        report["checks"] += 2
        return r0

    # ---- ML-KEM: every parameter set, n below / at / above the device count and ragged sizes ----
    for param in (512, 768, 1024):
        for n in ((1, 3, 7, 8, 9, 1000, 40011) if param == 768 else (5, 2049)):
            seeds = rng.integers(0, 256, (n, 64), dtype=np.uint8)
            ek, dk = both(hostapi.mlkem_keygen, param, seeds)
            m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            if n >= 8:
                ek[n // 2, 0:2] = 0xff  # one non-canonical key: its status byte must land in the right shard
            ct, ss, st = both(hostapi.mlkem_encaps, param, ek, m)
            assert st.sum() == (1 if n >= 8 else 0)
            ss2, st2 = both(hostapi.mlkem_decaps, param, dk, ct)
            good = st == 0
            assert (ss2[good] == ss[good]).all()
            k = min(n, 600)
            ek0, dk0 = orc.mlkem_keygen(param, seeds[:k])
            ekc = ek[:k].copy()
            ct0, ss0, st0 = orc.mlkem_encaps(param, ekc, m[:k])
            assert (dk[:k] == dk0).all() and (ct[:k] == ct0).all() and (ss[:k] == ss0).all() and (st[:k] == st0).all()
            # one key for the batch, and a key table
            both(hostapi.mlkem_encaps_shared, param, ek[:1], m)
            both(hostapi.mlkem_decaps_shared, param, dk[:1], ct)
            nk = min(n, 37)
            idx = rng.integers(0, nk, n).astype(np.uint32)
            both(hostapi.mlkem_encaps_keyed, param, ek[:nk], idx, m)
            both(hostapi.mlkem_decaps_keyed, param, dk[:nk], idx, ct)
    # round-3 Kyber
    seeds = rng.integers(0, 256, (333, 64), dtype=np.uint8)
    ek, dk = both(hostapi.kyber_keygen, 768, seeds)
    ct, ss = both(hostapi.kyber_encaps, 768, ek, seeds[:, :32])
    assert eq(both(hostapi.kyber_decaps, 768, dk, ct), ss)

    # ---- ML-DSA: keygen, sign (shards below and above the 16-item switch to the round-based signer), verify ----
    for param, n in ((65, 5), (65, 700), (44, 130), (87, 67)):
        seeds = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        pk, sk = both(hostapi.mldsa_keygen, param, seeds)
        msgs = ragged(rng, n)
        ctxs = ragged(rng, n, 0, 40)
        rnd = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        sig = both(hostapi.mldsa_sign, param, sk, msgs, ctxs=ctxs, rnd=rnd)
        sig_det = both(hostapi.mldsa_sign, param, sk, msgs)
        k = min(n, 48)
        assert (sig[:k] == orc.mldsa_sign(param, sk[:k], msgs[:k], ctxs=ctxs[:k], rnd=rnd[:k])).all()
        assert (sig_det[:k] == orc.mldsa_sign(param, sk[:k], msgs[:k])).all()
        bad = sig.copy()
        bad[::7, 40] ^= 4
        ok = both(hostapi.mldsa_verify, param, pk, bad, msgs, ctxs=ctxs)
        want = np.ones(n, np.uint8)
        want[::7] = 0
        assert (ok == want).all()
        assert both(hostapi.mldsa_verify, param, pk, sig_det, msgs).all()
        # one key / key table
        sg1 = both(hostapi.mldsa_sign_shared, param, sk[:1], msgs)
        assert both(hostapi.mldsa_verify_shared, param, pk[:1], sg1, msgs).all()
        nk = min(n, 11)
        idx = rng.integers(0, nk, n).astype(np.uint32)
        okk = both(hostapi.mldsa_verify_keyed, param, pk[:nk], idx, sig_det, msgs)
        assert (okk == (idx == np.arange(n)).astype(np.uint8)).all()

    # ---- hybrids and X25519 ----
    for scheme in (hostapi.XWING, hostapi.X25519MLKEM768):
        S = hostapi.HYBRID_SIZES[scheme]
        n = 1501
        seeds = rng.integers(0, 256, (n, S["seed"]), dtype=np.uint8)
        es = rng.integers(0, 256, (n, S["eseed"]), dtype=np.uint8)
        pk, sk = both(hostapi.hybrid_keygen, scheme, seeds)
        ct, ss, st = both(hostapi.hybrid_encaps, scheme, pk, es)
        ss2, st2 = both(hostapi.hybrid_decaps, scheme, sk, ct)
        assert not st.any() and not st2.any() and (ss == ss2).all()
        k = 64
        ct0, ss0, _ = (ohyb.xwing_encaps if scheme == hostapi.XWING else ohyb.hybrid_encaps)(pk[:k], es[:k])
        assert (ct[:k] == ct0).all() and (ss[:k] == ss0).all()
    sc = rng.integers(0, 256, (999, 32), dtype=np.uint8)
    both(hostapi.x25519, sc)
    # ---- key tables that live across calls: on device 0, on the last device, REPLICATED over all (device = -1: every call shards) ----
    def tables(kind, param, keys, call):
        res = []
        for dev in (0, nd - 1, ALL):
            t = hostapi.KeyTable(kind, param, keys, device=dev)
            assert L.circl_hip_keytable_device(t.handle) == dev and L.circl_hip_keytable_nkeys(t.handle) == len(keys)
            for d in range(nd):  # circl_hip_keytable_on_device: the replica / the table itself / NULL
                assert bool(L.circl_hip_keytable_on_device(t.handle, d)) == (dev == ALL or d == dev)
            res.append(call(t))
            t.close()
        assert eq(res[0], res[1]) and eq(res[0], res[2]), kind
        report["checks"] += 2
        return res[0]
    L = nat.lib()
    for n in (5, 4099):
        nk = 13
        ek, dk = hostapi.mlkem_keygen(768, rng.integers(0, 256, (nk, 64), dtype=np.uint8))
        idx = rng.integers(0, nk, n).astype(np.uint32)
        m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        ct, ss, st = tables("mlkem-public", 768, ek, lambda t: t.encaps(m, idx))
        assert eq((ct, ss, st), hostapi.mlkem_encaps(768, ek[idx], m))
        assert eq(tables("mlkem-private", 768, dk, lambda t: t.decaps(ct, idx)), hostapi.mlkem_decaps(768, dk[idx], ct))
        pk, sk = hostapi.mldsa_keygen(65, rng.integers(0, 256, (nk, 32), dtype=np.uint8))
        msgs = ragged(rng, n, 0, 120)
        sig = tables("mldsa-private", 65, sk, lambda t: t.sign(msgs, key_idx=idx))
        k = min(n, 40)
        assert (sig[:k] == orc.mldsa_sign(65, sk[idx[:k]], msgs[:k])).all()
        assert tables("mldsa-public", 65, pk, lambda t: t.verify(sig, msgs, key_idx=idx)).all()
        for scheme in (hostapi.XWING, hostapi.X25519MLKEM768):
            S = hostapi.HYBRID_SIZES[scheme]
            hpk, hsk = hostapi.hybrid_keygen(scheme, rng.integers(0, 256, (nk, S["seed"]), dtype=np.uint8))
            es = rng.integers(0, 256, (n, S["eseed"]), dtype=np.uint8)
            hct, hss, hst = tables("hybrid-public", scheme, hpk, lambda t: t.hybrid_encaps(es, idx))
            assert eq((hct, hss, hst), hostapi.hybrid_encaps(scheme, hpk[idx], es))
            got = tables("hybrid-private", scheme, hsk, lambda t: t.hybrid_decaps(hct, idx))
            assert (got[0] == hss).all() and not got[1].any()
    assert eq(both(hostapi.mldsa_public_from_private, 65, sk), pk)
    # ---- primitives and the sponge service ----
    both(hostapi.keccak_f1600, rng.integers(0, 1 << 63, (777, 25), dtype=np.uint64))
    both(hostapi.kyber_ntt, rng.integers(0, 3329, (515, 256)).astype(np.int16))
    both(hostapi.dilithium_ntt, rng.integers(0, 8380417, (515, 256)).astype(np.uint32))
    both(hostapi.xof, 168, 0x1f, ragged(rng, 300, 0, 900), 96)
    print(json.dumps(report))


def concurrent():
    """Three callers at once, each splitting its own batch over all logical devices (and a fourth on one device)."""
    nd, _ = expect_devices()
    rng = np.random.default_rng(7)
    n = 30011
    seeds = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    ek, dk = hostapi.mlkem_keygen(768, seeds, device=0)
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ref = hostapi.mlkem_encaps(768, ek, m, device=0)
    ss_ref, _ = hostapi.mlkem_decaps(768, dk, ref[0], device=0)
    dseeds = rng.integers(0, 256, (2500, 32), dtype=np.uint8)
    pk, sk = hostapi.mldsa_keygen(65, dseeds, device=0)
    msgs = ragged(rng, 2500)
    sig_ref = hostapi.mldsa_sign(65, sk, msgs, device=0)
    errs = []

    def kem_caller(dev):
        try:
            for _ in range(3):
                assert eq(hostapi.mlkem_encaps(768, ek, m, device=dev), ref)
                assert eq(hostapi.mlkem_decaps(768, dk, ref[0], device=dev)[0], ss_ref)
        except BaseException as e:  # noqa: BLE001
            errs.append(repr(e))

    def dsa_caller(dev):
        try:
            for _ in range(2):
                assert eq(hostapi.mldsa_sign(65, sk, msgs, device=dev), sig_ref)
                assert hostapi.mldsa_verify(65, pk, sig_ref, msgs, device=dev).all()
        except BaseException as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=kem_caller, args=(ALL,)), threading.Thread(target=kem_caller, args=(ALL,)),
          threading.Thread(target=dsa_caller, args=(ALL,)), threading.Thread(target=kem_caller, args=(nd - 1,))]
    t = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    print(json.dumps({"logical_devices": nd, "callers": len(th), "seconds": time.perf_counter() - t}))


def config3(lg):
    """BASELINE configs[2] in its stated shape: ML-KEM-768 Encaps + Decaps of n = 2^lg items (2^23 by default) as contiguous
    shards over the L logical devices through ONE call each, distinct GPU-generated keys, every ss_dec == ss_enc, a 2^16
    sample re-done by the oracle."""
    nd, nphys = expect_devices()
    n = 1 << lg
    t0 = time.perf_counter()
    seeds = np.random.default_rng(3).integers(0, 256, (n, 64), dtype=np.uint8)
    m = np.random.default_rng(4).integers(0, 256, (n, 32), dtype=np.uint8)
    # outputs allocated and touched once, calls timed on their second run: the figures below are the host path itself, not the
    # kernel's first-touch page faults of 40 GB of fresh numpy memory
    L = nat.lib()
    P = hostapi._p
    ek, dk = np.zeros((n, 1184), np.uint8), np.zeros((n, 2400), np.uint8)
    ct, ss, ss2 = np.zeros((n, 1088), np.uint8), np.zeros((n, 32), np.uint8), np.zeros((n, 32), np.uint8)
    st, st2 = np.ones(n, np.uint8), np.ones(n, np.uint8)
    t_kg = t_enc = t_dec = 0.0
    for rep in range(2):
        t = time.perf_counter()
        nat.check(L.circl_hip_mlkem_keygen(768, P(seeds), P(ek), P(dk), n, ALL), "keygen")
        t_kg = time.perf_counter() - t
        t = time.perf_counter()
        nat.check(L.circl_hip_mlkem_encaps(768, P(ek), P(m), P(ct), P(ss), P(st), n, ALL), "encaps")
        t_enc = time.perf_counter() - t
        t = time.perf_counter()
        nat.check(L.circl_hip_mlkem_decaps(768, P(dk), P(ct), P(ss2), P(st2), n, ALL), "decaps")
        t_dec = time.perf_counter() - t
    assert not st.any() and not st2.any()
    all_equal = bool((ss == ss2).all())
    assert all_equal
    # the shard boundaries and a uniform sample against the oracle
    rng = np.random.default_rng(5)
    edges = np.unique(np.clip(np.concatenate([[n * d // nd + o for o in (-1, 0, 1)] for d in range(nd + 1)]), 0, n - 1))
    idx = np.unique(np.concatenate([edges, rng.choice(n, size=min(n, 1 << 16), replace=False)]))
    ct0, ss0, st0 = orc.mlkem_encaps(768, ek[idx], m[idx])
    ssd0, _ = orc.mlkem_decaps(768, dk[idx], ct[idx])
    exact = bool((ct[idx] == ct0).all() and (ss[idx] == ss0).all() and (ss2[idx] == ssd0).all() and not st0.any())
    assert exact
    print(json.dumps({"logical_devices": nd, "hip_devices": nphys, "n": n, "shard_items": n // nd, "all_items_ss_dec_equals_ss_enc": all_equal,
                      "oracle_sample": int(len(idx)), "bit_exact_vs_oracle": exact, "keygen_s": t_kg, "encaps_s": t_enc, "decaps_s": t_dec,
                      "encaps_per_s_host_path": n / t_enc, "decaps_per_s_host_path": n / t_dec, "pairs_per_s_host_path": n / (t_enc + t_dec),
                      "wall_s": time.perf_counter() - t0}))


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "parity":
        parity()
    elif cmd == "concurrent":
        concurrent()
    elif cmd == "config3":
        config3(int(sys.argv[2]) if len(sys.argv) > 2 else 23)
    else:
        raise SystemExit("unknown sub-command")
