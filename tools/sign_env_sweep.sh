#!/bin/bash
# Not a test: signing rate under CIRCL_HIP_SIGN_* variants.   tools/sign_env_sweep.sh "<env 1>" "<env 2>" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for e in "" "$@"; do
  env $e python tools/sign_rate.py ${PARAM:-65} ${LOGN:-18} 5 | tail -1
done
