set -u
ROOT=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
for v in new old; do
  [ $v = old ] && cp $ROOT/build/libcirclhip_old.so $ROOT/circl_amd/libcirclhip.so
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/cmp/$v -o $v -- $CMD 2>&1 | tail -1 | cut -c1-200
  python - <<PY
import csv,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open("$ROOT/gpurun_out/cmp/$v/${v}_counter_collection.csv")):
    if "encrypt" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$v", {k: sorted(v)[len(v)//2] for k,v in d.items()})
PY
done
