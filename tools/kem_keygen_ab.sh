#!/bin/bash
# Not a test: key generation latency, H(ek) per lane (CIRCL_HIP_KEM_SPLIT=0 CIRCL_HIP_KEM_COOP=0) against the small-batch forms.
export CIRCL_LATENCY_LOGNS=${LOGNS:-0,10,12,14,16,17} CIRCL_LATENCY_ALL=1
for e in "CIRCL_HIP_KEM_SPLIT=0 CIRCL_HIP_KEM_COOP=0" "CIRCL_HIP_KEM_COOP=0" ""; do
  echo "== $e"; env $e timeout 200 python tests/gpu_microbench.py 18 latency 2>&1 | grep "keygen" | sed 's/.*| keygen/   keygen/'
done
