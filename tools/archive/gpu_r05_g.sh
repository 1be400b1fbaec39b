#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05g; mkdir -p $OUT
{
for e in "" "CIRCL_HIP_SIGN_FRONT=0"; do
  for a in "65 1" "87 5" "44 3 shared" "65 1500" "44 777" "87 600" "3 640" "65 900 shared" "65 40" "2 7" "5 33"; do
    echo "== $e $a: $(env $e timeout 300 python tests/sign_worker.py $a 2>&1 | tail -1 | tr '\n' ' ')"
  done
done
} > $OUT/parity.txt 2>&1
{
for lg in 1 0; do echo "CIRCL_HIP_SIGN_FRONT=$lg"; for p in 65 44 87; do CIRCL_HIP_SIGN_FRONT=$lg python tools/dsa_sign_small.py $p; done; CIRCL_HIP_SIGN_FRONT=$lg python tools/table_latency.py 2>&1 | sed -n 1,3p | sed 's/.*ML-DSA-65 verify/ML-DSA-65 verify/'; done
} 2>&1 | grep -v amdgpu.ids > $OUT/latency.txt
timeout 600 python -m pytest tests/test_gpu_mldsa.py tests/test_gpu_keytable.py -x -q 2>&1 | tail -2 >> $OUT/parity.txt
cat $OUT/parity.txt | cut -c1-200; cat $OUT/latency.txt
