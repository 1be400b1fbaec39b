#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05h; mkdir -p $OUT
{ python tools/table_latency.py; for p in 65 44 87; do python tools/dsa_sign_small.py $p; done; } 2>&1 | grep -v amdgpu.ids > $OUT/table_latency.txt
B=tools/bin/concurrent_bench
{ timeout 100 $B sign 256 0 1 2 1 8 64 128; timeout 60 $B sign 0 0 1 2 1; } > $OUT/concurrent_sign.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/full.log 2>&1; tail -3 $OUT/full.log
cut -c1-330 $OUT/table_latency.txt; cut -c1-300 $OUT/concurrent_sign.txt
