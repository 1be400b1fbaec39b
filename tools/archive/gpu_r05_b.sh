#!/bin/bash
# Round 5, GPU call B: coalescer after the sleeping lock (CPU cost per call, leader sync modes, signing), zero-copy cut-off, signing A/B
# (w kernel without the paired path in the long rounds; one launch less in front of / behind a prepared-key signature), host-side cost of
# an 8-GPU node.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05b; mkdir -p $OUT
{ nproc; cat /sys/fs/cgroup/cpu.max; lscpu | grep "Model name" | head -1; free -g | head -2; } > $OUT/box.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_coalesce.py tests/test_gpu_mldsa.py tests/test_gpu_keytable.py tests/test_gpu_round4.py tests/test_gpu_sanitizers.py -x -q > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
B=tools/bin/concurrent_bench
{
  echo "== encaps, coalescing 256 (sleeping reservation lock), CPU per call from cpu.stat"
  timeout 150 $B encaps 256 0 1 2 1 8 32 64 128 256 1024
  echo "== the same, CIRCL_HIP_COALESCE_BLOCKING=1 (leader sleeps on an interrupt-driven event)"
  CIRCL_HIP_COALESCE_BLOCKING=1 timeout 150 $B encaps 256 0 1 2 1 8 64 256 1024
  echo "== no coalescing (CPU per call)"
  timeout 100 $B encaps 0 0 1 2 1 8 64
  echo "== decaps / verify / sign, coalescing 256"
  timeout 100 $B decaps 256 0 1 2 1 64 256
  timeout 100 $B verify 256 0 1 2 1 64 256
  timeout 150 $B sign 256 0 1 2 1 8 64 256
  echo "== sign, no coalescing"
  timeout 100 $B sign 0 0 1 2 1 8 64
  echo "== encaps, 8 items per call, coalescing 1024"
  timeout 100 $B encaps 1024 0 8 2 8 64 256
  echo "== encaps, coalescing 256, CIRCL_HIP_ZEROCOPY_KB=4096"
  CIRCL_HIP_ZEROCOPY_KB=4096 timeout 100 $B encaps 256 0 1 2 64 256
} > $OUT/concurrent.txt 2>&1
{
  for kb in 64 1024 4096; do echo "CIRCL_HIP_ZEROCOPY_KB=$kb"; CIRCL_HIP_ZEROCOPY_KB=$kb python tools/host_small.py 5; done
} 2>&1 | grep -v amdgpu.ids > $OUT/host_small.txt
{
  for rep in 1 2; do for v in 0 1; do for p in 87 65; do CIRCL_HIP_SIGN_W_SINGLES=$v python tools/sign_rate.py $p 18 4 2>&1 | grep ML-DSA; done; done; done
  CIRCL_HIP_SIGN_W_SINGLES=0 python tools/sign_rate.py 44 18 4 2>&1 | grep ML-DSA; CIRCL_HIP_SIGN_W_SINGLES=1 python tools/sign_rate.py 44 18 4 2>&1 | grep ML-DSA
} > $OUT/sign_ab.txt 2>&1
{ python tools/table_latency.py; python tools/dsa_sign_small.py 65; } 2>&1 | grep -v amdgpu.ids > $OUT/table_latency.txt
{ CIRCL_HIP_LOGICAL_DEVICES=8 python tools/logical8.py 23; CIRCL_HIP_LOGICAL_DEVICES=8 CIRCL_HIP_HOST_THREADS=4 python tools/logical8.py 23; python tools/logical8.py 22; } 2>&1 | grep -v amdgpu.ids > $OUT/logical8.txt
cat $OUT/concurrent.txt; cat $OUT/host_small.txt; cat $OUT/sign_ab.txt; cut -c1-330 $OUT/table_latency.txt; cat $OUT/logical8.txt
