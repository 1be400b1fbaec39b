#!/bin/bash
# Not a test: the one-launch small-batch encapsulation against the two-launch form.
export CIRCL_LATENCY_LOGNS=${LOGNS:-0,8,10,11,12,13,14,15}
for f in 0 1; do
  echo "== CIRCL_HIP_KEM_FUSED=$f"
  CIRCL_HIP_KEM_FUSED=$f timeout 120 python tests/gpu_microbench.py 18 latency 2>&1 | grep "encaps  n="
done
