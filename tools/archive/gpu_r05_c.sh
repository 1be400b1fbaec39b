#!/bin/bash
# Round 5, GPU call C: process-wide coalescing of calls that bring their own keys, the coalescer after its last changes, the new defaults
# (zero-copy up to 1 MB, 4 byte movers per device), sanitizers over all of it.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_coalesce.py tests/test_gpu_sanitizers.py tests/test_gpu_host_mirror.py tests/test_gpu_hybrid.py tests/test_gpu_round3.py -x -q > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
B=tools/bin/concurrent_bench
{
  echo "== encaps to a key that comes WITH the call (circl_hip_mlkem_encaps, 4 096 distinct keys), no coalescing"
  timeout 100 $B encaps_item 0 0 1 2 1 8 64
  echo "== the same, circl_hip_set_coalesce(256, 0)"
  timeout 150 $B encaps_item 256 0 1 2 1 8 32 64 128
  echo "== resident keys, coalescing 256 (last version: statistics per batch, the last writer's wake only when the leader waits)"
  timeout 150 $B encaps 256 0 1 2 1 8 32 64 128
  timeout 100 $B decaps 256 0 1 2 32 64
  timeout 100 $B sign 256 0 1 2 32 64 128
} > $OUT/concurrent.txt 2>&1
python tools/host_small.py 5 2>&1 | grep -v amdgpu.ids > $OUT/host_small.txt
{ python tools/logical8.py 22; CIRCL_HIP_LOGICAL_DEVICES=8 python tools/logical8.py 23; } 2>&1 | grep -v amdgpu.ids > $OUT/logical8.txt
cat $OUT/concurrent.txt | cut -c1-300; cat $OUT/host_small.txt; cat $OUT/logical8.txt
