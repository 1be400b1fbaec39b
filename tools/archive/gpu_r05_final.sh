#!/bin/bash
# round 5, the last GPU call of the round (14 GPU-minutes were left): the WHOLE GPU suite on the final library (the cooperative permutation
# of 647de99 / 5c11870 had only met the tests of the routes that run on it), then -- only if it is green -- the bench line as the driver runs
# it and the small-call latency tables again.  Ordered by priority: the box-time limit may cut the tail.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05final; mkdir -p $OUT
T0=$SECONDS
sha256sum circl_amd/libcirclhip.so > $OUT/lib.sha256
timeout 470 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=6 --durations=12 > $OUT/tests.log 2>&1
rc=$?
echo "tests rc=$rc after $((SECONDS - T0)) s"; tail -30 $OUT/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi   # keep what is left of the budget for the fix
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 240 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$? after $((SECONDS - T0)) s"; tail -3 $OUT/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05final/bench.json").read().strip().splitlines()[-1])
    print("value %.4e ms %.3f frac %.4f mix %s wall %.0f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"],
          (d["roofline"].get("valu") or {}).get("frac_of_mix_ceiling"), d["bench_wall_s"]))
    print("parity", d.get("parity")); print("small", json.dumps(d["configs"].get("small_batches"))[:1500])
except Exception as e:
    print("no bench line:", e)
PY
timeout 60 python tools/table_latency.py 2>&1 | grep "n=2" > $OUT/table_latency.txt; cat $OUT/table_latency.txt
{ for p in 44 65 87; do CIRCL_LATENCY_LOGNS=0,4,6,8,10 timeout 60 python tools/dsa_latency.py $p; done; timeout 60 python tools/dsa_sign_small.py 65; } 2>&1 | grep "ML-DSA" > $OUT/dsa_latency.txt
cat $OUT/dsa_latency.txt
CIRCL_LATENCY_ALL=1 CIRCL_LATENCY_LOGNS=0,6,10,12,14 timeout 90 python tests/gpu_microbench.py 0 latency 2>&1 | grep "n=2\|decaps" > $OUT/latency.txt; cat $OUT/latency.txt
timeout 90 python tools/host_small.py 2>&1 | grep -v amdgpu.ids | tail -25 > $OUT/host_small.txt; tail -12 $OUT/host_small.txt
echo "done after $((SECONDS - T0)) s"
