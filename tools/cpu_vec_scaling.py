#!/usr/bin/env python3
"""Thread scaling of the two CPU baselines of bench.py on this host (no GPU needed): the scalar oracle (oracle/kyber.c) and the
batch-vectorised port (oracle/vec), ML-KEM-768 distinct-key encapsulation, AVX2 and AVX-512.  -> profiles/r05_cpu_vec.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "?"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
    rng = np.random.default_rng(5)
    ek, _ = orc.mlkem_keygen(768, rng.integers(0, 256, (n, 64), dtype=np.uint8))
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    cap = orc.ncpu()
    print(f"{cpu_model()}; CPUs this process may use: {cap} (affinity {len(os.sched_getaffinity(0))}); ML-KEM-768 encapsulation, {n} distinct keys")
    ns = min(n, 1 << 13)
    ct0, ss0, st0 = orc.mlkem_encaps(768, ek[:ns], m[:ns], threads=cap)
    for isa in (1, 2):
        if orc.vec_isa(isa) != isa:
            continue
        ct, ss, st = orc.mlkem_encaps_vec(768, ek[:ns], m[:ns], threads=cap, isa=isa)
        print(f"  isa {isa}: first {ns} items equal the scalar oracle: {bool((ct == ct0).all() and (ss == ss0).all() and (st == st0).all())}")
    threads = sorted({t for t in (1, 2, 4, 8, 16, 32, 64) if t <= 2 * cap})
    for label, fn, reps in [("scalar oracle (oracle/kyber.c)", lambda t: orc.mlkem_encaps(768, ek[:ns * 2], m[:ns * 2], threads=t), 1)] + \
                           [(f"oracle/vec {'AVX2' if isa == 1 else 'AVX-512'}", (lambda t, isa=isa: orc.mlkem_encaps_vec(768, ek, m, threads=t, isa=isa)), 3)
                            for isa in (1, 2) if orc.vec_isa(isa) == isa]:
        items = ns * 2 if label.startswith("scalar") else n
        row = []
        for t in threads:
            best = 1e9
            for _ in range(reps):
                t0 = time.perf_counter()
                fn(t)
                best = min(best, time.perf_counter() - t0)
            row.append(f"T={t}: {items / best:.3e}/s ({best / items * t * 1e6:.2f} us x thread per item)")
        print(f"{label}\n    " + "\n    ".join(row))


if __name__ == "__main__":
    main()
