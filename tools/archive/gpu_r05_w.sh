#!/bin/bash
# round 5, run w: the bounded index vector (KeyIdx) -- its own tests, every key-table test, the sanitizer drivers, a short bench line
mkdir -p gpurun_out/r05w
python -m pytest tests/test_gpu_keyidx.py tests/test_gpu_keytable.py tests/test_gpu_round2.py tests/test_gpu_coalesce.py tests/test_gpu_sanitizers.py -x -q > gpurun_out/r05w/tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r05w/tests.log
tail -5 gpurun_out/r05w/tests.log
python bench.py --steps 200 --warmup 20 > gpurun_out/r05w/bench.json 2> gpurun_out/r05w/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05w/bench.json").read().strip().splitlines()[-1])
print("value %.4e ms %.3f frac %.4f mix %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], (d["roofline"].get("valu") or {}).get("frac_of_mix_ceiling")))
print("parity", d.get("parity"))
PY
