"""Not a test: what this GPU's HBM delivers to plain streaming kernels (the ceiling for the matrix reads of batch signing).   python tools/hbm_probe.py"""
import time

import torch

n = 1 << 30  # 4 GiB of int32
x = torch.ones(n, dtype=torch.int32, device="cuda")
y = torch.empty_like(x)


def t(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - s)
    return best


b = 4 * n
print(f"read  (sum)      : {b / t(lambda: x.sum()) / 1e12:.2f} TB/s")
print(f"write (fill)     : {b / t(lambda: y.fill_(3)) / 1e12:.2f} TB/s")
print(f"copy  (read+write): {2 * b / t(lambda: y.copy_(x)) / 1e12:.2f} TB/s")
print(f"add   (2r + 1w)  : {3 * b / t(lambda: torch.add(x, y, out=y)) / 1e12:.2f} TB/s")
