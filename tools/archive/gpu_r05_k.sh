#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05k; mkdir -p $OUT
B=tools/bin/concurrent_bench
{ timeout 100 $B encaps 256 0 1 2 8 32 64 96 128; echo "== uncoalesced"; timeout 60 $B encaps 0 0 1 2 8 64; echo "== BLOCKING"; CIRCL_HIP_COALESCE_BLOCKING=1 timeout 60 $B encaps 256 0 1 2 64 128; which perf; } > $OUT/cpu_split.txt 2>&1
cut -c1-420 $OUT/cpu_split.txt
