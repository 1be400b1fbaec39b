#!/bin/bash
# round 5, run x: the shorter cooperative permutation (keccak_f1600_coop2: ~43 instead of ~70 instructions per round).
# 1. the tests of every path that runs on it;  2. the small-call latencies under the previous library (tools/bin/libcirclhip_prev.so,
# built from HEAD) and the current one, alternating on ONE box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05x; mkdir -p $OUT
python -m pytest tests/test_gpu_prims.py tests/test_gpu_keytable.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_mldsa.py tests/test_gpu_mlkem.py -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
cp circl_amd/libcirclhip.so tools/bin/libcirclhip_cur.so
for rep in 1 2; do
  for v in prev cur; do
    cp tools/bin/libcirclhip_$v.so circl_amd/libcirclhip.so
    echo "== $v (rep $rep)"
    CIRCL_LATENCY_ALL=1 CIRCL_LATENCY_LOGNS=0,6,10 python tests/gpu_microbench.py 0 latency 2>&1 | grep "n=2\|decaps"
    CIRCL_LATENCY_LOGNS=0,6 python tools/dsa_latency.py 65 2>&1 | grep ML-DSA
    CIRCL_LATENCY_LOGNS=0 python tools/dsa_latency.py 44 2>&1 | grep ML-DSA
    CIRCL_LATENCY_LOGNS=0 python tools/dsa_latency.py 87 2>&1 | grep ML-DSA
    python tools/table_latency.py 2>&1 | grep "n=2^0 \|n=2^6 "
  done
done > $OUT/ab.txt 2>&1
cp tools/bin/libcirclhip_cur.so circl_amd/libcirclhip.so
cat $OUT/ab.txt
