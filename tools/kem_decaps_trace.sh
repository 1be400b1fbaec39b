#!/bin/bash
# Not a test: kernel durations of the small-batch ML-KEM decapsulation routes (rocprofv3 --kernel-trace), one batch size per run.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for logn in ${LOGNS:-12 14 15}; do
  rm -rf /tmp/kst
  CIRCL_LATENCY_ALL=1 CIRCL_LATENCY_LOGNS=$logn rocprofv3 --kernel-trace --output-format csv -d /tmp/kst -o t -- python $ROOT/tests/gpu_microbench.py 18 latency > /tmp/kst.log 2>&1
  echo "== 2^$logn"; grep "decaps " /tmp/kst.log
  python - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("/tmp/kst/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[(r["Kernel_Name"].split("(")[0][:80], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "mlkem" in k and len(v) >= 10: v = sorted(v); print(f"   {k:80s} grid {g:>9s} x{len(v):3d} median {v[len(v)//2]:8.1f} us")
PY
done
