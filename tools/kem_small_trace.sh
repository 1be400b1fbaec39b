#!/bin/bash
# Not a test: kernel durations of the small-batch ML-KEM routes (rocprofv3 --kernel-trace --stats), one batch size per run.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for logn in ${LOGNS:-12 13 14 15}; do
  rm -rf /tmp/kst; 
  CIRCL_LATENCY_LOGNS=$logn rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o t -- python $ROOT/tests/gpu_microbench.py 18 latency > /tmp/kst.log 2>&1
  echo "== 2^$logn"; grep "encaps  n=2" /tmp/kst.log
  python - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("/tmp/kst/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "mlkem" in k and len(v) > 2: v = sorted(v)[-20:]; print(f"   {k:90s} top-20 launches: median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")  # (the batch-of-one tail of the curve launches the same kernels 50 x)
PY
done
