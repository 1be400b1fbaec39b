#!/bin/bash
# Not a test: run the small-batch latency curve under library variants built by tools/variant_lib.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
export CIRCL_LATENCY_LOGNS=${LOGNS:-8,12,14}
cp circl_amd/libcirclhip.so build/libcirclhip_cur.so
for v in cur "$@"; do
  cp build/libcirclhip_$v.so circl_amd/libcirclhip.so
  echo "== $v"; timeout 120 python tests/gpu_microbench.py 18 latency 2>&1 | grep "encaps  n=2"
done
cp build/libcirclhip_cur.so circl_amd/libcirclhip.so
