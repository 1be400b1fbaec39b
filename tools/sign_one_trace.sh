#!/bin/bash
# Not a test: the timeline of ONE ML-DSA-65 signature with a prepared key (circl_hip_mldsa_sign_table_dev, n = 1) under rocprofv3 --kernel-trace:
# every kernel of the last of ten calls with its duration and the gap to its predecessor.   tools/sign_one_trace.sh [param]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
PARAM=${1:-65}
cat > /tmp/s1.py <<PY
import sys, time, numpy as np, torch
sys.path.insert(0,'$ROOT')
from circl_amd import device as cdev, hostapi
g=torch.Generator(device='cuda').manual_seed(1)
d=cdev.MLDSADevice($PARAM,1,'cuda',sign=True)
pk,sk=d.keygen(torch.randint(0,256,(1,32),dtype=torch.uint8,device='cuda',generator=g))
st=hostapi.KeyTable("mldsa-private",$PARAM,sk.cpu().numpy())
msg=torch.randint(0,256,(48,),dtype=torch.uint8,device='cuda',generator=g)
sig=d.sign_table(st,msg); torch.cuda.synchronize()
for _ in range(10):
    d.sign_table(st,msg,sig); torch.cuda.synchronize(); time.sleep(0.002)
PY
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/s1t -o s1t -- python /tmp/s1.py > /dev/null 2>&1
python - <<PY
import csv, glob
f=glob.glob("$ROOT/gpurun_out/s1t/**/*kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
# the last call: find the last kernel whose name holds 'sign_front'
idx=[i for i,r in enumerate(rows) if 'sign_front' in r["Kernel_Name"] or 'long_scan' in r["Kernel_Name"]]
i0=idx[-1]
t0=int(rows[i0]["Start_Timestamp"]); prev=None
for r in rows[i0:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    name=r["Kernel_Name"].replace("circl::mldsa::","").split("(")[0][:60]
    print(f"{(s-t0)/1e3:8.1f} us  +{((s-prev)/1e3 if prev else 0):6.1f} gap  {(e-s)/1e3:7.1f} us  {name}")
    prev=e
print(f"first kernel start to last kernel end: {(prev-t0)/1e3:.1f} us")
PY
