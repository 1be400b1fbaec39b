#!/bin/bash
# Not a test: where the small-batch routes of ML-KEM should switch (CIRCL_HIP_KEM_COOP / CIRCL_HIP_KEM_SMALL), measured.
export CIRCL_LATENCY_LOGNS=${LOGNS:-12,13,14,15}
for coop in 12 13 14 15; do
  for small in 14 15; do
    [ $coop -gt $small ] && continue
    echo "== COOP=$coop SMALL=$small"
    CIRCL_HIP_KEM_COOP=$coop CIRCL_HIP_KEM_SMALL=$small timeout 120 python tests/gpu_microbench.py 18 latency 2>&1 | grep "encaps  n=2"
  done
done
