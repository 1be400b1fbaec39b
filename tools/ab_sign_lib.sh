#!/bin/bash
# Not a test: signing rate under library variants from tools/bin, alternating on ONE box:  tools/ab_sign_lib.sh <variant> ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp circl_amd/libcirclhip.so tools/bin/libcirclhip_cur.so
for rep in 1 2; do
  for v in "$@"; do
    cp tools/bin/libcirclhip_$v.so circl_amd/libcirclhip.so
    echo "$v: $(python tools/sign_rate.py ${PARAM:-65} 18 4 2>&1 | grep ML-DSA)"
  done
done
cp tools/bin/libcirclhip_cur.so circl_amd/libcirclhip.so
