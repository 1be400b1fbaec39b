#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05s; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -k keygen_routes > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
{ for lg in 6 0; do for p in 44 65 87; do CIRCL_HIP_DSA_KEYGEN_CHAIN=$lg python tools/dsa_latency.py $p 2>&1 | head -4; done; done; } 2>&1 | grep -v amdgpu.ids | cut -c1-140 > $OUT/latency.txt
cat $OUT/latency.txt
