#!/bin/bash
# per-kernel time of one ML-DSA-65 batch signing call (2 calls of n = 2^16) under rocprofv3 --kernel-trace --stats
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
cat > /tmp/sg.py <<PY
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0,'$ROOT')
from circl_amd import _native as nat
from oracle import orc
L=nat.lib(); param=65; PK,SK,SIG=orc.DSA_SIZES[param]; n=1<<16
rng=np.random.default_rng(1); pool=1<<10
pk,sk=orc.mldsa_keygen(param, rng.integers(0,256,(pool,32),dtype=np.uint8))
d_sk=torch.from_numpy(np.tile(sk,(n//pool,1))).cuda()
d_msg=torch.from_numpy(rng.integers(0,256,32*n+16,dtype=np.uint8)).cuda()
d_off=torch.arange(0,32*(n+1),32,dtype=torch.int64).cuda()
d_rnd=torch.zeros((n,32),dtype=torch.uint8,device='cuda')
sig=torch.empty((n,SIG),dtype=torch.uint8,device='cuda')
wsb=L.circl_hip_mldsa_sign_workspace_size(param,n); ws=torch.empty(wsb,dtype=torch.uint8,device='cuda')
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(2):
    rc=L.circl_hip_mldsa_sign_dev(param,d_sk.data_ptr(),d_msg.data_ptr(),d_off.data_ptr(),None,None,d_rnd.data_ptr(),0,sig.data_ptr(),n,ws.data_ptr(),wsb,st); assert rc==0
    torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/sgt -o sgt -- python /tmp/sg.py > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$ROOT/gpurun_out/sgt/sgt_kernel_stats.csv")))
tot=0
for r in rows:
    if "mldsa" in r["Name"] or "sign_" in r["Name"]:
        ms=float(r["TotalDurationNs"])/1e6/2; tot+=ms
        print(f"{r['Name'].split('(')[0][:60]:60s} calls/2 {int(r['Calls'])//2:4d}  {ms:8.3f} ms per signing call  avg {float(r['AverageNs'])/1e3:8.1f} us")
print("sum of kernel time per call: %.3f ms" % tot)
PY
