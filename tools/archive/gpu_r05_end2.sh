#!/bin/bash
# round 5: the bench line once more after cpu_baseline learnt the one-key shape (oracle/vec shared)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05end2; mkdir -p $OUT
timeout 60 python -m pytest tests/test_oracle_vec.py -q -p no:cacheprovider 2>&1 | tail -1
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -2 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05end2/bench.json").read().strip().splitlines()[-1])
c = d["cpu_baseline"]
print("value %.4e ms %.3f frac %.4f mix %s wall %.0f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], (d["roofline"].get("valu") or {}).get("frac_of_mix_ceiling"), d["bench_wall_s"]))
print("cpu %.3e shared %.3e | scalar %.3e shared %.3e" % (c["value"], c["shared_key"]["value"], c["scalar_oracle"]["value"], c["scalar_oracle"]["shared_key"]["value"]))
print(c["vectorized"]["shared_key"]); print("gpu shared", d["configs"]["shared_key"]["value"])
PY
