#!/bin/bash
# GPU call b: TSan retry, signing round modes, sign timing pair vs single
mkdir -p gpurun_out
cat /proc/sys/vm/mmap_rnd_bits > gpurun_out/r3b_box.txt
timeout 900 python -m pytest tests/test_gpu_sanitizers.py tests/test_gpu_round3.py -q -k "sanitizer or sign_round_modes or config3_stated" --durations=8 > gpurun_out/r3b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3b_box.txt
timeout 600 python -m pytest tests/test_gpu_mldsa.py tests/test_gpu_fullsize.py -q -x > gpurun_out/r3b_dsa_tests.log 2>&1
echo "dsa tests rc=$?" >> gpurun_out/r3b_box.txt
for p in 65 44 87; do
  bash tools/sign_trace.sh $p 18 > gpurun_out/r3b_sign_trace_pair_$p.txt 2>&1
done
CIRCL_HIP_SIGN_PAIR=0 bash tools/sign_trace.sh 65 18 > gpurun_out/r3b_sign_trace_single_65.txt 2>&1
tail -4 gpurun_out/r3b_tests.log; tail -3 gpurun_out/r3b_dsa_tests.log; cat gpurun_out/r3b_box.txt; head -8 gpurun_out/r3b_sign_trace_pair_65.txt; head -8 gpurun_out/r3b_sign_trace_single_65.txt
