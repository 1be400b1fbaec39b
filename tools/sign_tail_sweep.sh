cd $GRAFT_REPO_ROOT
for t in 2 4 8 16 32 64; do echo "== tail $t"; CIRCL_HIP_SIGN_TAIL=$t python tests/gpu_microbench.py 18 2>&1 | grep "DSA-65 sign"; CIRCL_HIP_SIGN_TAIL=$t python tests/gpu_microbench.py 16 2>&1 | grep "DSA-65 sign"; done
