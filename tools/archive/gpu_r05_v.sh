#!/bin/bash
# soak: the four coalesced operations side by side in four processes, 15 s each, 48 callers each; then the randomised oracle soak for 2 minutes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05v; mkdir -p $OUT
B=tools/bin/concurrent_bench
for op in encaps decaps verify sign; do ( timeout 120 $B $op 256 0 1 15 48 > $OUT/soak_$op.txt 2>&1; echo "$op rc=$?" >> $OUT/soak_$op.txt ) & done; wait
cat $OUT/soak_*.txt | grep "T=\|rc=" | cut -c1-200
timeout 400 python tools/stress.py 5 120 2>&1 | tail -3
