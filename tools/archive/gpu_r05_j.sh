#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05j; mkdir -p $OUT
{ for rep in 1 2; do for v in 0 1; do for p in 65 87; do CIRCL_HIP_SIGN_STAGGER=$v python tools/sign_rate.py $p 18 4 2>&1 | grep ML-DSA; done; done; done; } > $OUT/stagger.txt 2>&1
cat $OUT/stagger.txt
