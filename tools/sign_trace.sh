#!/bin/bash
# per-kernel and per-round time of one ML-DSA batch signing call under rocprofv3 --kernel-trace --stats
#   tools/sign_trace.sh [param] [log2 n]
# The trace runs with CIRCL_HIP_SIGN_NOSPLIT=1 (the whole batch on one stream) so that the kernels of a round follow each other;
# the rate of the default path (two halves side by side) is printed first.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
PARAM=${1:-65}; LOGN=${2:-16}
cat > /tmp/sg.py <<PY
import sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0,'$ROOT')
from circl_amd import device as cdev
param=$PARAM; n=1<<$LOGN
g=torch.Generator(device='cuda').manual_seed(1)
eng=cdev.MLDSADevice(param,n,'cuda',sign=True)
seeds=torch.randint(0,256,(n,32),dtype=torch.uint8,device='cuda',generator=g)
msg=torch.randint(0,256,(n*32+16,),dtype=torch.uint8,device='cuda',generator=g)
pk,sk=eng.keygen(seeds)
sig=eng.sign(sk,msg); torch.cuda.synchronize()
for _ in range(2):
    t=time.perf_counter(); eng.sign(sk,msg,sig); te=time.perf_counter()-t; torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(f"ML-DSA-{param} sign n={n}: enqueue {te*1e3:.2f} ms, complete {dt*1e3:.2f} ms -> {n/dt:.3e}/s")
PY
echo "default path (two halves on two streams), not profiled:"; python /tmp/sg.py 2>&1 | grep "ML-DSA"
echo "one stream (CIRCL_HIP_SIGN_NOSPLIT=1), under rocprofv3 --kernel-trace:"
CIRCL_HIP_SIGN_NOSPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/sgt -o sgt -- python /tmp/sg.py 2>&1 | grep "ML-DSA"
python - <<PY
import csv
rows=list(csv.DictReader(open("$ROOT/gpurun_out/sgt/sgt_kernel_stats.csv")))
tot=0
for r in rows:
    if any(k in r["Name"] for k in ("mldsa","sign_","fillBuffer")):
        ms=float(r["TotalDurationNs"])/1e6/3; tot+=ms
        print(f"{r['Name'].split('(')[0][:60]:60s} calls/3 {int(r['Calls'])//3:4d}  {ms:8.3f} ms per signing call  avg {float(r['AverageNs'])/1e3:8.1f} us")
print("sum of kernel time per call: %.3f ms" % tot)
tr=sorted(csv.DictReader(open("$ROOT/gpurun_out/sgt/sgt_kernel_trace.csv")), key=lambda r:int(r["Start_Timestamp"]))
# the last signing call: from the last sign_prep kernel on
idx=[i for i,r in enumerate(tr) if "mldsa_sign_prep_kernel" in r["Kernel_Name"]][-1]
call=tr[idx:]
t0=int(call[0]["Start_Timestamp"])
rnd=0
print("last call, kernel by kernel (start offset us, duration us, grid):")
for r in call:
    nm=r["Kernel_Name"].split("(")[0].replace("circl::mldsa::","").replace("void ","")[:34]
    if "sign_mask" in nm: rnd+=1
    print(f"  r{rnd:02d} {nm:34s} +{(int(r['Start_Timestamp'])-t0)/1e3:9.1f}  {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  grid {r.get('Grid_Size','?')}")
print("call span: %.3f ms" % ((int(call[-1]["End_Timestamp"])-t0)/1e6))
PY
