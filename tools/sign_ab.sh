#!/bin/bash
# Not a test: signing rate of the current library against build/libcirclhip_<base>.so, with CIRCL_HIP_SIGN_* variants.
#   tools/sign_ab.sh <base> "<env for current 1>" "<env 2>" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
base=$1; shift
cp circl_amd/libcirclhip.so build/libcirclhip_new.so
cp build/libcirclhip_$base.so circl_amd/libcirclhip.so
echo "== $base"; python tools/sign_rate.py ${PARAM:-65} ${LOGN:-18} 5 | tail -1
cp build/libcirclhip_new.so circl_amd/libcirclhip.so
for e in "" "$@"; do
  echo "== new $e"; env $e python tools/sign_rate.py ${PARAM:-65} ${LOGN:-18} 5 | tail -1
done
