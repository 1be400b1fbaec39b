#!/bin/bash
# round 5, the last GPU minutes: pure KERNEL durations of the one-item routes on the final library (rocprofv3 --kernel-trace of
# tools/table_latency.py at n = 1: every call of the table is one launch), next to the per-call costs the same run prints; then the
# timeline of one prepared-key ML-DSA-65 signature.   -> profiles/r05_one_item_kernels.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r05trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1
CIRCL_LATENCY_LOGNS=0 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt1 -o t -- python $ROOT/tools/table_latency.py > $OUT/calls.txt 2>&1
grep "n=2" $OUT/calls.txt
python - > $OUT/kernels.txt <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("/tmp/kt1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0].replace("circl::mlkem::", "").replace("circl::mldsa::", "").replace("void ", "")[:84]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-84s %6s %9s %9s %9s" % ("kernel (n = 1 calls of tools/table_latency.py)", "calls", "median us", "min us", "max us"))
for k, v in sorted(d.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 20 and ("mlkem" in k or "mldsa" in k or "sign" in k or "keccak" in k):
        v = sorted(v); print("%-84s %6d %9.1f %9.1f %9.1f" % (k, len(v), v[len(v) // 2], v[0], v[-1]))
PY
cat $OUT/kernels.txt
cd $ROOT && timeout 90 bash tools/sign_one_trace.sh 65 > $OUT/sign_timeline.txt 2>&1; tail -25 $OUT/sign_timeline.txt
rm -rf $ROOT/gpurun_out/s1t
