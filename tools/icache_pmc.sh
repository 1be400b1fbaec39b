ROOT=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQC?_[A-Z_0-9]*(ICACHE|IFETCH|INST_PREFETCH|DCACHE)[A-Z_0-9]*" | sort -u | tr '\n' ' '; echo
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $ROOT/gpurun_out/ic -o ic -- $CMD > $ROOT/gpurun_out/ic.log 2>&1
python - <<PY
import csv,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open("$ROOT/gpurun_out/ic/ic_counter_collection.csv")):
    if "encrypt" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: "%.3e" % sorted(v)[len(v)//2] for k,v in d.items()})
PY
done
tail -3 $ROOT/gpurun_out/ic.log | cut -c1-200
