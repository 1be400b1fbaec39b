#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05q; mkdir -p $OUT
{ for lg in 8 9 10; do for p in 65 87; do CIRCL_HIP_SIGN_CHAIN_LOG2=$lg python tools/dsa_sign_small.py $p; done; done; python tools/table_latency.py | cut -c1-30,170-330 | head -5; } 2>&1 | grep -v amdgpu.ids > $OUT/sweep.txt
B=tools/bin/concurrent_bench
{ timeout 100 $B sign 256 0 1 2 1 64 128 256; } > $OUT/concurrent_sign.txt 2>&1
cat $OUT/sweep.txt; cut -c1-300 $OUT/concurrent_sign.txt
