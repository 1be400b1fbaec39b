#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   pass 1: --kernel-trace --stats   (per-kernel durations; must agree with bench.py's HIP events)
#   pass 2: --pmc FETCH_SIZE         (HBM read bytes;  own pass, kernel-trace only)
#   pass 3: --pmc WRITE_SIZE         (HBM write bytes; own pass)
# Raw output goes to gpurun_out/prof (scratch); profiles/summarize.py condenses it into the files
# that are committed under profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
TAG=${1:-r01}
STEPS=${2:-5}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- $CMD > "$OUT/kt.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $CMD > "$OUT/write.log" 2>&1
cd "$ROOT" && python profiles/summarize.py "$OUT" "$TAG" > "$OUT/summary_$TAG.log" 2>&1
tail -40 "$OUT/summary_$TAG.log"
