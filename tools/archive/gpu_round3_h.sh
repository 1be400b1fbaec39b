#!/bin/bash
mkdir -p gpurun_out
tools/bin/ablate > gpurun_out/r3h_ablate_w4.txt 2>&1
tools/bin/ablate_w8 > gpurun_out/r3h_ablate_w8.txt 2>&1
CIRCL_BENCH_WRITE_PMC=gpurun_out/pmc_json timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03_b.json 2> gpurun_out/bench_r03_b.err
echo "bench rc=$?"
grep -A12 "probe 3b" gpurun_out/r3h_ablate_w4.txt; grep -A30 "probe 3a" gpurun_out/r3h_ablate_w8.txt | grep -v "NTT\|wave/SIMD: [0-9.]* ms" ; grep "scratch" gpurun_out/r3h_ablate_w4.txt; tail -3 gpurun_out/bench_r03_b.err
