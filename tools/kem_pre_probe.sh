#!/bin/bash
# Not a test: which role of the small-batch pre kernel takes its time, and how many ring-phase workgroups per CU are best.
for dbg in 1 2; do
  echo "== pre kernel role $dbg only (1 = hashes, 2 = expansion)"
  CIRCL_HIP_DEBUG_PRE=$dbg LOGNS="${LOGNS:-12 13 14}" bash tools/kem_small_trace.sh 2>&1 | grep "==\|small_pre"
done
for w in 4 8 16 24 32; do
  echo "== ring-phase workgroups per CU: $w"
  CIRCL_HIP_KEM_SMALL_WGS=$w CIRCL_HIP_KEM_SMALL=15 LOGNS="11 12 13 14 15" bash tools/kem_small_trace.sh 2>&1 | grep "encrypt_kernel<3, 0, 0, true, 2>" | awk '{print "   " $0}'
done
