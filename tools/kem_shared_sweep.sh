#!/bin/bash
# Not a test: how far the one-key small-batch routes should reach (CIRCL_HIP_KEM_SMALL_SHARED = log2).
export CIRCL_LATENCY_LOGNS=${LOGNS:-15,16,17,18} CIRCL_LATENCY_ALL=1
for s in 14 16 17 18; do
  echo "== CIRCL_HIP_KEM_SMALL_SHARED=$s"
  CIRCL_HIP_KEM_SMALL_SHARED=$s timeout 200 python tests/gpu_microbench.py 18 latency 2>&1 | grep "decaps " | sed 's/.*| encaps, one key/   encaps, one key/'
done
