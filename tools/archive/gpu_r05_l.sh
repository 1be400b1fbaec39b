#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05l; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_coalesce.py tests/test_gpu_sanitizers.py tests/test_gpu_host_mirror.py -x -q > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
B=tools/bin/concurrent_bench
{ timeout 150 $B encaps 256 0 1 2 1 8 32 64 96 128 256; timeout 100 $B decaps 256 0 1 2 64 128; timeout 100 $B verify 256 0 1 2 64 128; timeout 100 $B sign 256 0 1 2 64 128 256; timeout 100 $B encaps_item 256 0 1 2 64 128; timeout 60 $B encaps 1024 0 8 2 64 128; } > $OUT/concurrent.txt 2>&1
cut -c1-420 $OUT/concurrent.txt
