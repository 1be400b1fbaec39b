#!/bin/bash
# byte movers per device: rate and host CPU for aligned and byte-misaligned pageable arrays (what a Go caller's sub-slices are)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05e; mkdir -p $OUT
for t in 2 4 6 8 12 16; do
  CIRCL_HIP_HOST_THREADS=$t python tools/logical8.py 20 misaligned 2>&1 | grep "device = 0\|arrays" | sed "s/^/movers=$t /"
  CIRCL_HIP_HOST_THREADS=$t python tools/logical8.py 20 2>&1 | grep "device = 0" | sed "s/^/movers=$t aligned /"
done > $OUT/movers.txt 2>&1
cat $OUT/movers.txt | cut -c1-260
