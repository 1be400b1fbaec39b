#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round3.py -q -k "sign_round_modes" > gpurun_out/r3c_tests.log 2>&1
echo "modes rc=$?" > gpurun_out/r3c_box.txt
timeout 900 python -m pytest tests/test_gpu_mldsa.py tests/test_gpu_fullsize.py -q -x --durations=12 > gpurun_out/r3c_dsa_tests.log 2>&1
echo "dsa tests rc=$?" >> gpurun_out/r3c_box.txt
bash tools/sign_trace.sh 65 18 > gpurun_out/r3c_sign_trace_pair_65.txt 2>&1
CIRCL_HIP_SIGN_PAIR=0 bash tools/sign_trace.sh 65 18 > gpurun_out/r3c_sign_trace_single_65.txt 2>&1
tail -3 gpurun_out/r3c_tests.log; tail -16 gpurun_out/r3c_dsa_tests.log; cat gpurun_out/r3c_box.txt; head -8 gpurun_out/r3c_sign_trace_pair_65.txt; grep "r01" gpurun_out/r3c_sign_trace_pair_65.txt; head -8 gpurun_out/r3c_sign_trace_single_65.txt;  grep "r01" gpurun_out/r3c_sign_trace_single_65.txt
