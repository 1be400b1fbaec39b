#!/bin/bash
# the one-launch signing round: parity on every parameter set and size class, then what it buys
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05f; mkdir -p $OUT
{
for e in "" "CIRCL_HIP_SIGN_CHAIN_LOG2=16" "CIRCL_HIP_SIGN_CHAIN_LOG2=0"; do
  for a in "65 1" "87 5" "44 3 shared" "65 1500" "44 777" "87 600" "3 640" "65 900 shared"; do
    echo "== $e $a: $(env $e timeout 300 python tests/sign_worker.py $a 2>&1 | tail -2 | tr '\n' ' ')"
  done
done
} > $OUT/parity.txt 2>&1
{
for lg in 9 0 6 7 8 10 11; do echo "CIRCL_HIP_SIGN_CHAIN_LOG2=$lg"; CIRCL_HIP_SIGN_CHAIN_LOG2=$lg python tools/dsa_sign_small.py 65; CIRCL_HIP_SIGN_CHAIN_LOG2=$lg python tools/table_latency.py 2>&1 | cut -c1-30,170-330 | head -5; done
for p in 44 87; do CIRCL_HIP_SIGN_CHAIN_LOG2=9 python tools/dsa_sign_small.py $p; CIRCL_HIP_SIGN_CHAIN_LOG2=0 python tools/dsa_sign_small.py $p; done
} 2>&1 | grep -v amdgpu.ids > $OUT/latency.txt
cat $OUT/parity.txt | cut -c1-250; cat $OUT/latency.txt
