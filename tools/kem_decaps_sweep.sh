#!/bin/bash
# Not a test: where the small-batch decapsulation routes should end (CIRCL_HIP_KEM_SMALL = log2), and lane pairs against lanes.
export CIRCL_LATENCY_LOGNS=${LOGNS:-10,11,12,13,14,15,16} CIRCL_LATENCY_ALL=1
for cfg in "12 1" "13 1" "14 1" "15 1" "15 0"; do
  set -- $cfg
  echo "== CIRCL_HIP_KEM_SMALL=$1 CIRCL_HIP_KEM_SPLIT=$2"
  CIRCL_HIP_KEM_SMALL=$1 CIRCL_HIP_KEM_SPLIT=$2 timeout 200 python tests/gpu_microbench.py 18 latency 2>&1 | grep "n=2\|decaps " | grep -v "host-buffer"
done
