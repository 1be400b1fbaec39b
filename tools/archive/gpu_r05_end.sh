#!/bin/bash
# round 5, closing call: thread scaling of the two CPU baselines on the GPU box's host, the bench line the way the driver asks for it,
# and the bench-driving GPU tests on the final tree
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05end; mkdir -p $OUT
T0=$SECONDS
timeout 120 python tools/cpu_vec_scaling.py 65536 > $OUT/cpu_vec.txt 2>&1; cat $OUT/cpu_vec.txt
echo "cpu done after $((SECONDS - T0)) s"
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$? after $((SECONDS - T0)) s"; tail -2 $OUT/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05end/bench.json").read().strip().splitlines()[-1])
    c = d["cpu_baseline"]
    print("value %.4e ms %.3f frac %.4f mix %s | cpu %.3e (%s) scalar %.3e | wall %.0f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"],
          (d["roofline"].get("valu") or {}).get("frac_of_mix_ceiling"), c["value"], c["vectorized"]["sample"][:60], c["scalar_oracle"]["value"], d["bench_wall_s"]))
    print("parity", d["parity"])
except Exception as e:
    print("no bench line:", e)
PY
timeout 150 python -m pytest tests/test_gpu_round2.py -k "bench_py" -x -q -p no:cacheprovider 2>&1 | tail -3
echo "done after $((SECONDS - T0)) s"
