#!/bin/bash
# Not a test: the defaults, then wider speculation with a larger entry capacity for 2^12 .. 2^16 items.
python tools/dsa_latency.py ${PARAM:-65} | sed 's/verify.*| sign/sign/; s/| keygen.*//'
for e in "CIRCL_HIP_SIGN_MIN_ENTRIES=65536 CIRCL_HIP_SIGN_SPEC=256" "CIRCL_HIP_SIGN_MIN_ENTRIES=131072 CIRCL_HIP_SIGN_SPEC=512"; do
  echo "== $e"; for l in 12 14 16; do env $e python tools/sign_rate.py 65 $l 5 | tail -1; done
done
echo "== defaults"; for l in 12 14 16; do python tools/sign_rate.py 65 $l 5 | tail -1; done
