#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/r3d_sign_sweep.txt; : > $out
for pair in 0 1; do for w in 4 5 6; do for f in 4 5 6; do
  CIRCL_HIP_SIGN_PAIR=$pair CIRCL_HIP_SIGN_W_WAVES=$w CIRCL_HIP_SIGN_F_WAVES=$f python tools/sign_rate.py 65 18 4 2>&1 | grep "ML-DSA" >> $out
done; done; done
for p in 44 87; do for pair in 0 1; do CIRCL_HIP_SIGN_PAIR=$pair python tools/sign_rate.py $p 18 4 2>&1 | grep "ML-DSA" >> $out; done; done
cat $out
