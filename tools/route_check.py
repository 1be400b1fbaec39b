#!/usr/bin/env python3
"""tools/route_check.py -- are the library's route cut-overs on the right side ON THIS BOX?  

The library picks between a one-launch ("chain") route and the multi-launch batch route -- and between zero-copy and copied staging --
by thresholds that were tuned by hand on two box classes in rounds 4-5 (CIRCL_HIP_KEM_CHAIN, _KEM_CHAIN_ENCAPS, _KEM_CHAIN_ITEM,
_DSA_CHAIN, _DSA_CHAIN_ITEM, _SIGN_CHAIN_LOG2, _DSA_KEYGEN_CHAIN, _ZEROCOPY_KB: INTEGRATION.md).  For every cut-over this script
times BOTH routes -- the knob forced high (the one-launch / zero-copy route always) and at 0 (never) -- at half the threshold, at the
threshold, at twice and at four times its size, each (route, size) in two processes of its own (the faster median of REPS calls through
the C ABI: a process's placement moves a median by +-10 %), and requires the route the DEFAULT takes at that size to be within TOL (15 %)
of the better one.

Exit status 1 if a default loses by more than TOL anywhere; the table goes to stdout (profiles/r06_routes.txt).

    python tools/route_check.py            # every cut-over
    python tools/route_check.py --wide     # calibration: one more size per cut-over
    python tools/route_check.py kem_chain  # one of them
    python tools/route_check.py --worker <case> <n>   # (internal) one timing, prints microseconds
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TOL = 0.15
REPS = 15

# case -> (what is timed, the knob, its default as log2 items (or KB), value forcing route A (chain / zero-copy), value forcing route B)
CASES = {
    "kem_chain":        ("ML-KEM-768 decapsulation, resident keys (circl_hip_mlkem_decaps_table)", "CIRCL_HIP_KEM_CHAIN", 11, "20", "0"),
    "kem_chain_encaps": ("ML-KEM-768 encapsulation, resident keys (circl_hip_mlkem_encaps_table)", "CIRCL_HIP_KEM_CHAIN_ENCAPS", 10, "20", "0"),
    "kem_chain_item":   ("ML-KEM-768 encapsulation, keys with the call (circl_hip_mlkem_encaps)", "CIRCL_HIP_KEM_CHAIN_ITEM", 9, "20", "0"),
    "dsa_chain":        ("ML-DSA-65 verification, resident keys (circl_hip_mldsa_verify_table)", "CIRCL_HIP_DSA_CHAIN", 11, "16", "0"),
    "dsa_chain_item":   ("ML-DSA-65 verification, keys with the call (circl_hip_mldsa_verify)", "CIRCL_HIP_DSA_CHAIN_ITEM", 9, "16", "0"),
    "dsa_chain_item87": ("ML-DSA-87 verification, keys with the call (circl_hip_mldsa_verify)", "CIRCL_HIP_DSA_CHAIN_ITEM", 8, "16", "0"),
    "sign_chain":       ("ML-DSA-65 signing, prepared keys (circl_hip_mldsa_sign_table)", "CIRCL_HIP_SIGN_CHAIN_LOG2", 8, "16", "0"),
    "dsa_keygen_chain": ("ML-DSA-65 key generation (circl_hip_mldsa_keygen)", "CIRCL_HIP_DSA_KEYGEN_CHAIN", 9, "12", "0"),
    # zero-copy: the threshold is BYTES moved (2048 KB); an ML-KEM-768 resident-key encapsulation moves 1157 B per item
    "zerocopy":         ("ML-KEM-768 encapsulation, resident keys, by bytes moved (CIRCL_HIP_ZEROCOPY_KB)", "CIRCL_HIP_ZEROCOPY_KB", 2048, "1048576", "0"),
    "zerocopy_decaps":  ("ML-KEM-768 decapsulation, resident keys, by bytes moved (CIRCL_HIP_ZEROCOPY_KB; the one-launch route off)", "CIRCL_HIP_ZEROCOPY_KB", 2048, "1048576", "0"),
}


def worker(case, n):
    from circl_amd import hostapi
    rng = np.random.default_rng(7)
    if case in ("kem_chain", "kem_chain_encaps", "kem_chain_item", "zerocopy", "zerocopy_decaps"):
        nk = 8
        ek, dk = hostapi.mlkem_keygen(768, rng.integers(0, 256, (nk, 64), dtype=np.uint8))
        idx = (np.arange(n) % nk).astype(np.uint32)
        m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        if case == "kem_chain_item":
            eks = np.ascontiguousarray(ek[idx])
            call = lambda: hostapi.mlkem_encaps(768, eks, m)
        elif case in ("kem_chain", "zerocopy_decaps"):
            pub = hostapi.KeyTable("mlkem-public", 768, ek)
            ct, _, _ = pub.encaps(m, idx)
            prv = hostapi.KeyTable("mlkem-private", 768, dk)
            call = lambda: prv.decaps(ct, idx)
        else:
            pub = hostapi.KeyTable("mlkem-public", 768, ek)
            call = lambda: pub.encaps(m, idx)
    else:
        param = 87 if case.endswith("87") else 65
        nk = 4
        pk, sk = hostapi.mldsa_keygen(param, rng.integers(0, 256, (nk, 32), dtype=np.uint8))
        idx = (np.arange(n) % nk).astype(np.uint32)
        msgs = [bytes(rng.integers(0, 256, 64, dtype=np.uint8)) for _ in range(n)]
        if case == "dsa_keygen_chain":
            seeds = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            call = lambda: hostapi.mldsa_keygen(param, seeds)
        elif case == "sign_chain":
            signer = hostapi.KeyTable("mldsa-private", param, sk)
            call = lambda: signer.sign(msgs, key_idx=idx)
        else:
            signer = hostapi.KeyTable("mldsa-private", param, sk)
            sig = signer.sign(msgs, key_idx=idx)
            if case == "dsa_chain":
                ver = hostapi.KeyTable("mldsa-public", param, pk)
                call = lambda: ver.verify(sig, msgs, key_idx=idx)
            else:
                pks = np.ascontiguousarray(pk[idx])
                call = lambda: hostapi.mldsa_verify(param, pks, sig, msgs)
    for _ in range(3):
        call()
    ts = []
    for _ in range(REPS):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("%.2f" % (ts[len(ts) // 2] * 1e6))


def timed(case, n, knob, value):
    env = dict(os.environ)
    env[knob] = value
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", case, str(n)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("worker %s n=%d %s=%s failed: %s" % (case, n, knob, value, r.stderr[-400:]))
    return float(r.stdout.strip().splitlines()[-1])


def best_of(case, n, knob, value, runs=2):
    """the faster of `runs` processes' medians: a process's placement (NUMA node of its staging, clock state) moves a median by +-10 %"""
    return min(timed(case, n, knob, value) for _ in range(runs))


def main(argv):
    if len(argv) >= 3 and argv[0] == "--worker":
        worker(argv[1], int(argv[2]))
        return 0
    wide = "--wide" in argv  # calibration: one more size per cut-over
    cases = [a for a in argv if a in CASES] or list(CASES)
    wrong = 0
    print("route A = the one-launch / zero-copy route (knob forced high), route B = the batch / copied route (knob 0); every (route, size) in two\n"
          "processes of its own, the faster median of %d calls each, microseconds; the default takes A up to its threshold, B beyond; it must be\n"
          "within %.0f %% of the better route at every size" % (REPS, TOL * 100))
    for case in cases:
        what, knob, dflt, force_a, force_b = CASES[case]
        zc = case.startswith("zerocopy")
        per = 32 + 1088 + 32 + 1 + 4
        ns = [(dflt * 1024 * (2 ** k) // 2) // per - (k == 1) for k in range(0, 4 + wide)] if zc else [1 << (dflt + k) for k in range(-1, 3 + wide)]
        limit = dflt * 1024 // per if zc else 1 << dflt
        print("\n== %s\n   %s, default %s%s" % (what, knob, dflt, " KB" if zc else " (log2 items)"))
        print("   %8s %12s %12s   %s" % ("n", "route A", "route B", "the default's route"))
        last_a_win = None
        for n in ns:
            if case == "sign_chain":  # (a per-ROUND threshold: 'route A' = the default, 'route B' = never; forcing it high changes other rounds too)
                ta, tb = best_of(case, n, knob, str(dflt)), best_of(case, n, knob, "0")
                takes_a = True
            else:
                ta, tb = best_of(case, n, knob, force_a), best_of(case, n, knob, force_b)
                takes_a = n <= limit
            t_def, t_alt = (ta, tb) if takes_a else (tb, ta)
            loss = t_def / t_alt - 1
            bad = loss > TOL
            wrong += bad
            if ta <= tb:
                last_a_win = n
            print("   %8d %12.1f %12.1f   %s: %s" % (n, ta, tb, "A" if takes_a else "B",
                                                    "ok" if not bad else "WRONG SIDE (loses %.0f %%)" % (loss * 100)) + ("" if loss <= 0 or bad else "  (%.0f %% behind)" % (loss * 100)))
        print("   route A wins up to n = %s" % last_a_win)
    print("\n%s" % ("every default is within %.0f %% of the better route at every size" % (TOL * 100) if not wrong else "%d point(s) on the wrong side" % wrong))
    return 1 if wrong else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
