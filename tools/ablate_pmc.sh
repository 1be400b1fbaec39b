set -u
ROOT=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $ROOT/gpurun_out/abl -o abl -- $ROOT/build/ablate > /dev/null 2>&1
python - <<PY
import csv,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$ROOT/gpurun_out/abl/abl_counter_collection.csv")):
    k=r["Kernel_Name"].split("(")[0]
    d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in d.items():
    print(k[:90], {c.replace("SQ_INSTS_",""): "%.3e"%sorted(x)[len(x)//2] for c,x in v.items()}, len(list(v.values())[0]))
PY
