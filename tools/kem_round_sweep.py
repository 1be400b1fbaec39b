"""Not a test: how the big-batch ML-KEM encapsulation kernels' times move with the batch size around whole "rounds".

The encrypt kernel runs R resident single-wave workgroups (an occupancy query; 4 096 on MI355X at 4 waves per SIMD) which pull
groups of G = 64 / K^2 items from a ticket counter, so one "round" is R * G items (28 672 for ML-KEM-768).  This sweep times the
hash and the encrypt kernel (the library's own HIP-event brackets, circl_hip_profile_read) for batches of x rounds, x whole and
fractional, to separate what a mid-size batch pays for the partial last round from what it pays once per launch.

    python tools/kem_round_sweep.py [param] > gpurun_out/kem_round_sweep.txt     (on the GPU box)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from circl_amd import device as cdev  # noqa: E402


def main():
    param = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    K = {512: 2, 768: 3, 1024: 4}[param]
    G = 64 // (K * K)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    R = cus * 16
    rnd = R * G
    print(f"ML-KEM-{param}: {cus} CUs, assumed {R} resident workgroups, group {G} items, round {rnd} items")
    xs = [0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 2.0, 2.29, 2.5, 3.0, 3.5, 4.0, 4.57, 5.0, 6.0, 8.0, 9.0, 9.14, 9.5, 10.0, 12.0, 16.0, 18.29, 24.0, 32.0, 36.57]
    nmax = int(max(xs) * rnd) + 64
    rng = np.random.default_rng(7)
    big = cdev.MLKEMDevice(param, nmax)
    seeds = torch.from_numpy(rng.integers(0, 256, (nmax, 64), dtype=np.uint8)).cuda()
    ek, _dk = big.keygen(seeds)
    m = torch.from_numpy(rng.integers(0, 256, (nmax, 32), dtype=np.uint8)).cuda()
    del big
    print(f"{'rounds':>7} {'n':>9} {'call us':>9} {'hash us':>9} {'encrypt us':>10} {'enc us/round':>12} {'items/s':>10}")
    for x in xs:
        n = int(round(x * rnd))
        eng = cdev.MLKEMDevice(param, n)
        e, mm = ek[:n], m[:n]
        for _ in range(3):
            eng.encaps(e, mm)
        torch.cuda.synchronize()
        for k in ("mlkem_hash", "mlkem_encrypt"):
            cdev.profile_read(k)
        cdev.profile_enable(True)
        iters = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            eng.encaps(e, mm)
        e1.record()
        torch.cuda.synchronize()
        cdev.profile_enable(False)
        call = e0.elapsed_time(e1) / iters * 1e3
        h = cdev.profile_read("mlkem_hash")[0] / iters * 1e3
        c = cdev.profile_read("mlkem_encrypt")[0] / iters * 1e3
        print(f"{x:7.2f} {n:9d} {call:9.1f} {h:9.1f} {c:10.1f} {c / x:12.1f} {n / call * 1e6:10.3e}")
        assert int(eng.status.sum()) == 0
        del eng


if __name__ == "__main__":
    os.environ.setdefault("CIRCL_HIP_KEM_SMALL", "0")  # big-batch routes for every size (the sweep is about that kernel)
    main()
