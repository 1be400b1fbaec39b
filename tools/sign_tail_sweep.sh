cd $GRAFT_REPO_ROOT
for t in 1 2; do echo "== tail $t per CU"; for p in 65 44 87; do CIRCL_HIP_SIGN_TAIL=$t python - <<PY 2>&1 | grep sign
import sys; sys.argv=["x","18"]
exec(open("tests/gpu_microbench.py").read().split('if __name__ == "__main__"')[0])
dsa($p, 1 << 16)
PY
done; done
