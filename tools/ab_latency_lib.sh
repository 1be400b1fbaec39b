#!/bin/bash
# Not a test: the ML-KEM latency table (tests/gpu_microbench.py latency) under library variants from tools/bin
# (tools/variant_lib.sh), alternating with the current build on ONE box:   LOGNS=13,14 tools/ab_latency_lib.sh <variant> [...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp circl_amd/libcirclhip.so tools/bin/libcirclhip_cur.so
for rep in 1 2; do
  for v in cur "$@"; do
    cp tools/bin/libcirclhip_$v.so circl_amd/libcirclhip.so
    echo "== $v (rep $rep)"
    CIRCL_LATENCY_ALL=1 CIRCL_LATENCY_LOGNS=${LOGNS:-13,14,15} python tests/gpu_microbench.py ${PARAM:-18} latency 2>&1 | grep "n=2\|decaps"
  done
done
cp tools/bin/libcirclhip_cur.so circl_amd/libcirclhip.so
