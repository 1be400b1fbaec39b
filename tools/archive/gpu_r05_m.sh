#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_coalesce.py tests/test_gpu_sanitizers.py tests/test_gpu_host_mirror.py -x -q > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
for i in 1 2 3; do CIRCL_HIP_LOGICAL_DEVICES=4 timeout 300 tests/_san/race_driver_tsan 6 3 3000 100 2>&1 | tail -2; done > $OUT/tsan_more.log 2>&1; tail -6 $OUT/tsan_more.log
