#!/bin/bash
# Not a test: signing latency of small / medium batches against the speculation width (CIRCL_HIP_SIGN_SPEC = list entries per CU)
# and the schedule's stopping rule (CIRCL_HIP_SIGN_EPS_LOG2: expected unsigned items behind the planned rounds, 2^-x).
for e in "" "CIRCL_HIP_SIGN_SPEC=64" "CIRCL_HIP_SIGN_SPEC=32" "CIRCL_HIP_SIGN_SPEC=16" "CIRCL_HIP_SIGN_EPS_LOG2=20" "CIRCL_HIP_SIGN_SPEC=32 CIRCL_HIP_SIGN_EPS_LOG2=20"; do
  echo "== $e"; env $e python tools/dsa_latency.py ${PARAM:-65} | sed 's/verify.*| sign/sign/; s/| keygen.*//'
done
