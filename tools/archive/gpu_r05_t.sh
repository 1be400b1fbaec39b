#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05t; mkdir -p $OUT
{ for lg in 12 0; do for p in 44 65 87; do echo "CIRCL_HIP_DSA_KEYGEN_CHAIN=$lg"; CIRCL_HIP_DSA_KEYGEN_CHAIN=$lg python tools/dsa_latency.py $p 2>&1 | sed -n 3,6p; done; done; } 2>&1 | grep -v amdgpu.ids | sed 's/verify.*sign/sign/' | cut -c1-120 > $OUT/latency.txt
cat $OUT/latency.txt
