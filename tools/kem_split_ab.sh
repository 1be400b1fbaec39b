#!/bin/bash
# Not a test: the lane-pair hashing of the small-batch encapsulation against the lane-per-item form; kernel durations too.
export CIRCL_LATENCY_LOGNS=${LOGNS:-12,13,14,15}
for f in 0 1; do
  echo "== CIRCL_HIP_KEM_SPLIT=$f"
  CIRCL_HIP_KEM_SPLIT=$f timeout 120 python tests/gpu_microbench.py 18 latency 2>&1 | grep "encaps  n=2"
done
for c in 11 10; do
  echo "== CIRCL_HIP_KEM_COOP=$c (lane pairs above)"
  CIRCL_HIP_KEM_COOP=$c timeout 120 python tests/gpu_microbench.py 18 latency 2>&1 | grep "encaps  n=2"
done
