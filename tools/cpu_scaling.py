import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from oracle import orc
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: pass
rng=np.random.default_rng(1)
n=20000
ek,dk=orc.mlkem_keygen(768, rng.integers(0,256,(n,64),dtype=np.uint8))
m=rng.integers(0,256,(n,32),dtype=np.uint8)
for th in (1, 8, 32, 64, 128, 256):
    k = n if th > 1 else 2000
    t=time.perf_counter(); orc.mlkem_encaps(768, ek[:k], m[:k], threads=th); dt=time.perf_counter()-t
    print(th, "threads: %.3e encaps/s"%(k/dt))
