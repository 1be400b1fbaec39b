#!/bin/bash
# Round 5, GPU call A: the new host paths (coalescer, zero-copy small calls) -- correctness first, then what they buy.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05a; mkdir -p $OUT
nproc > $OUT/box.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/box.txt 2>/dev/null; lscpu | grep "Model name" >> $OUT/box.txt
timeout 900 python -m pytest tests/test_gpu_coalesce.py tests/test_gpu_keytable.py tests/test_gpu_host_mirror.py -x -q > $OUT/new_tests.log 2>&1; tail -5 $OUT/new_tests.log
B=tools/bin/concurrent_bench
{
  echo "== no coalescing: every call is its own launch (round-4 behaviour)"
  for op in encaps decaps verify; do timeout 120 $B $op 0 0 1 2 1 8 64 256; done
  echo "== coalescing, max_items 256, no linger"
  for op in encaps decaps verify; do timeout 120 $B $op 256 0 1 2 1 8 64 256 1024; done
  echo "== coalescing, max_items 1024, no linger, 8 items per call (a front-end that already groups a few handshakes)"
  timeout 120 $B encaps 1024 0 8 2 1 8 64 256
  echo "== coalescing, max_items 256, linger 50 us"
  timeout 120 $B encaps 256 50 1 2 1 8 64 256
  echo "== CIRCL_HIP_COALESCE_INFLIGHT=1 / 4 (encaps, max_items 256)"
  CIRCL_HIP_COALESCE_INFLIGHT=1 timeout 120 $B encaps 256 0 1 2 8 64 256
  CIRCL_HIP_COALESCE_INFLIGHT=4 timeout 120 $B encaps 256 0 1 2 8 64 256
  echo "== CIRCL_HIP_COALESCE_SPIN=2000 (encaps, max_items 256)"
  CIRCL_HIP_COALESCE_SPIN=2000 timeout 120 $B encaps 256 0 1 2 1 8 64
  echo "== CIRCL_HIP_ZEROCOPY_KB=0 (coalesced batches always copied), encaps + decaps"
  CIRCL_HIP_ZEROCOPY_KB=0 timeout 120 $B encaps 256 0 1 2 1 8 64 256
  CIRCL_HIP_ZEROCOPY_KB=0 timeout 120 $B decaps 256 0 1 2 1 64
  echo "== CIRCL_HIP_ZEROCOPY_KB=1024"
  CIRCL_HIP_ZEROCOPY_KB=1024 timeout 120 $B encaps 256 0 1 2 64 256 1024
  CIRCL_HIP_ZEROCOPY_KB=1024 timeout 120 $B decaps 256 0 1 2 64 256 1024
} > $OUT/concurrent.txt 2>&1
{
  for kb in 0 64 1024; do for rep in 1 2 3; do echo "CIRCL_HIP_ZEROCOPY_KB=$kb run $rep"; CIRCL_HIP_ZEROCOPY_KB=$kb python tools/host_small.py; done; done
} 2>&1 | grep -v amdgpu.ids > $OUT/host_small.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -5 $OUT/bench.err
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench.json"))
    print("bench", d["value"], d["ms_per_step"], d["config"]["key_pool"], d["parity"], d["roofline"]["valu"], d["configs"].get("pooled"), d["configs"]["config4"]["parity"], d["bench_wall_s"])
except Exception as e:
    print("bench parse failed", e)
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/full.log 2>&1; tail -3 $OUT/full.log
cat $OUT/concurrent.txt; cat $OUT/host_small.txt | cut -c1-260
