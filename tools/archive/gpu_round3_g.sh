#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/r3g_latency.txt
timeout 900 python -m pytest tests/test_gpu_mlkem.py tests/test_gpu_hybrid.py tests/test_gpu_round2.py tests/test_gpu_fullsize.py -q -x > gpurun_out/r3g_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r3g_box.txt
for c in 0 12 13; do echo "CIRCL_HIP_KEM_COOP=$c" >> gpurun_out/r3g_latency.txt; CIRCL_HIP_KEM_COOP=$c python tests/gpu_microbench.py 0 latency 2>&1 | grep "ML-KEM" | head -8 >> gpurun_out/r3g_latency.txt; done
tail -3 gpurun_out/r3g_tests.log; cat gpurun_out/r3g_box.txt gpurun_out/r3g_latency.txt
