#!/bin/bash
# round 5, final run: the whole GPU suite on the final library, then the small-call latency tables and the bench line again (the cooperative
# permutation changed under every one of them)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05z; mkdir -p $OUT
timeout 800 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -20 $OUT/tests.log
python tools/table_latency.py 2>&1 | grep "n=2" > $OUT/table_latency.txt
{ for p in 44 65 87; do CIRCL_LATENCY_LOGNS=0,4,6,8,10 python tools/dsa_latency.py $p; done; python tools/dsa_sign_small.py 65; } 2>&1 | grep "ML-DSA" > $OUT/dsa_latency.txt
CIRCL_LATENCY_ALL=1 CIRCL_LATENCY_LOGNS=0,6,10,12,14 python tests/gpu_microbench.py 0 latency 2>&1 | grep "n=2\|decaps" > $OUT/latency.txt
python tools/host_small.py 2>&1 | grep -v amdgpu.ids | tail -25 > $OUT/host_small.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/table_latency.txt $OUT/dsa_latency.txt $OUT/latency.txt; tail -12 $OUT/host_small.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05z/bench.json").read().strip().splitlines()[-1])
print("value %.4e ms %.3f frac %.4f mix %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], (d["roofline"].get("valu") or {}).get("frac_of_mix_ceiling")))
print("parity", d.get("parity")); print("small", json.dumps(d.get("small_batches"))[:1500])
PY
