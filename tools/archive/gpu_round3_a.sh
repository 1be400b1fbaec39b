#!/bin/bash
# first GPU call of round 3: the new tests, the old suite, one bench line
mkdir -p gpurun_out
{ free -g; nproc; lscpu | grep -i "model name\|socket\|numa"; rocm-smi --showmeminfo vram 2>/dev/null | head -5; } > gpurun_out/r3_box.txt 2>&1
t0=$(date +%s)
timeout 2000 python -m pytest tests/test_gpu_lane_prims.py tests/test_gpu_sanitizers.py tests/test_gpu_round3.py -q --durations=25 > gpurun_out/r3_new_tests.log 2>&1
echo "new tests rc=$? $(( $(date +%s) - t0 )) s" >> gpurun_out/r3_box.txt
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q --durations=15 --ignore=tests/test_gpu_lane_prims.py --ignore=tests/test_gpu_sanitizers.py --ignore=tests/test_gpu_round3.py > gpurun_out/r3_old_tests.log 2>&1
echo "old tests rc=$? $(( $(date +%s) - t0 )) s" >> gpurun_out/r3_box.txt
t0=$(date +%s)
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03_a.json 2> gpurun_out/bench_r03_a.err
echo "bench rc=$? $(( $(date +%s) - t0 )) s" >> gpurun_out/r3_box.txt
tail -5 gpurun_out/r3_new_tests.log; tail -3 gpurun_out/r3_old_tests.log; cat gpurun_out/r3_box.txt
