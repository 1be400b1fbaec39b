"""Not a test: device-resident timing of the ML-KEM kernels (run on the GPU box via gpurun)."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from circl_amd import _native as nat  # noqa: E402
from oracle import orc  # noqa: E402


def main(n=1 << 18, param=768, pool=1 << 12, iters=5):
    L = nat.lib()
    EK, _, CT = {512: (800, 1632, 768), 768: (1184, 2400, 1088), 1024: (1568, 3168, 1568)}[param]
    rng = np.random.default_rng(1)
    ekp, _ = orc.mlkem_keygen(param, rng.integers(0, 256, (pool, 64), dtype=np.uint8))
    ek = torch.from_numpy(np.tile(ekp, (n // pool, 1))).cuda()
    m = torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).cuda()
    ct = torch.empty((n, CT), dtype=torch.uint8, device="cuda")
    ss = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    st = torch.empty(n, dtype=torch.uint8, device="cuda")
    wsb = L.circl_hip_mlkem_workspace_size(param, n)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        rc = L.circl_hip_mlkem_encaps_dev(param, ek.data_ptr(), m.data_ptr(), ct.data_ptr(), ss.data_ptr(), st.data_ptr(), n,
                                          ws.data_ptr(), wsb, C.c_void_p(stream))
        assert rc == 0, rc

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"ML-KEM-{param} n={n}: {ms:.3f} ms/batch -> {n / ms * 1e3:.3e} encaps/s")
    # spot parity
    idx = rng.integers(0, n, 256)
    ct0, ss0, _ = orc.mlkem_encaps(param, ek[idx].cpu().numpy(), m[idx].cpu().numpy())
    print("parity:", bool((ct[idx].cpu().numpy() == ct0).all() and (ss[idx].cpu().numpy() == ss0).all()), "status", int(st.sum()))


if __name__ == "__main__":
    for p in (768, 512, 1024):
        main(param=p)
