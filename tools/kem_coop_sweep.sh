#!/bin/bash
# Not a test: up to which batch the two-items-per-wavefront hashing beats the lane-pair form (CIRCL_HIP_KEM_COOP = log2).
export CIRCL_LATENCY_LOGNS=${LOGNS:-6,8,9,10,11,12}
for c in ${COOPS:-1 8 9 10 11}; do
  echo "== CIRCL_HIP_KEM_COOP=$c"
  CIRCL_HIP_KEM_COOP=$c timeout 120 python tests/gpu_microbench.py 18 latency 2>&1 | grep "encaps  n=2"
done
