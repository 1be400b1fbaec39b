"""Worker of tests/test_gpu_round3.py::test_kem_route_boundaries: digests of ML-KEM outputs at the batch sizes where the routes
switch (2^11: two-per-wavefront / lane pairs, 2^14: ring-phase group count, 2^15: small / big batch, 2^17: one-key routes), so that
the same inputs can be taken through different routes in different processes (environment) and compared.
    python tests/kem_boundary_worker.py <param>"""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from circl_amd import device as cdev  # noqa: E402

p = int(sys.argv[1])
nmax = (1 << 17) + 1
g = torch.Generator(device="cuda").manual_seed(p)
seeds = torch.randint(0, 256, (nmax, 64), dtype=torch.uint8, device="cuda", generator=g)
m_all = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device="cuda", generator=g)
kg = cdev.MLKEMDevice(p, nmax)
ek_all, dk_all = kg.keygen(seeds)
torch.cuda.synchronize()
h = hashlib.sha256()
for n in (2047, 2048, 2049, 16384, 16385, 32767, 32768, 32769, 131072, 131073):
    eng = cdev.MLKEMDevice(p, n)
    ek, dk, m = ek_all[:n].contiguous(), dk_all[:n].contiguous(), m_all[:n].contiguous()
    ct, ss, st = eng.encaps(ek, m)
    ct = ct.clone()
    h.update(ct.cpu().numpy().tobytes() + ss.cpu().numpy().tobytes() + st.cpu().numpy().tobytes())
    ct[::5, 17] ^= 4  # implicit rejection for every fifth item
    ss2 = torch.empty_like(ss)
    st2 = torch.empty_like(st)
    eng.decaps(dk, ct, ss2, st2)
    h.update(ss2.cpu().numpy().tobytes() + st2.cpu().numpy().tobytes())
    ct1, ss1, st1 = eng.encaps_shared(ek[:1], m)
    ct1 = ct1.clone()
    h.update(ct1.cpu().numpy().tobytes() + ss1.cpu().numpy().tobytes() + st1.cpu().numpy().tobytes())
    ct1[::5, 17] ^= 4
    eng.decaps_shared(dk[:1], ct1, ss2, st2)
    h.update(ss2.cpu().numpy().tobytes() + st2.cpu().numpy().tobytes())
print("kem boundary digest", p, h.hexdigest())
