#!/bin/bash
# Not a test: the small-batch routes (tuned on ML-KEM-768) against the big-batch routes for the other parameter sets.
export CIRCL_LATENCY_LOGNS=${LOGNS:-0,10,12,13,14,15} CIRCL_LATENCY_ALL=1
for p in 512 1024; do
  for e in "" "CIRCL_HIP_KEM_SMALL=0 CIRCL_HIP_KEM_SMALL_SHARED=0 CIRCL_HIP_KEM_SMALL_SHARED_DECAPS=0"; do
    echo "== ML-KEM-$p ${e:-defaults}"; env $e timeout 200 python tests/gpu_microbench.py 18 latency $p 2>&1 | grep "n=2\|decaps " | grep -v host-buffer | sed 's/per call -> .*//' | paste - - | sed 's/ML-KEM-[0-9]* *//g' | cut -c1-230
  done
done
