#!/bin/bash
# Round-5 evidence, collected on the GPU box (run through gpurun); raw output stays in gpurun_out/prof_r05 (scratch), the condensed files
# are copied into profiles/ afterwards.
#   1. bench.py as the driver runs it (live PMC passes and the live VALU probe inside; whole-batch oracle parity)  -> r05_bench.json
#   2. rocprofv3 --kernel-trace --stats of a bench.py run with every config                                         -> r05_kernel_stats.txt
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* of the headline kernels (separate passes)                      -> r05_pmc.txt, traffic.json, valu.json
#   4. device-resident microbench at 2^18 / 2^20, small-batch latency sweep                                          -> r05_microbench*.txt, r05_latency.txt
#   5. signing rates of every parameter set, ML-DSA latencies                                                        -> r05_sign_rates.txt, r05_dsa_latency.txt
# (concurrent callers, zero-copy, host cost of a node, signing A/B: tools/archive/gpu_r05_{a,b,c}.sh -> profiles/r05_*.txt)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r05
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CIRCL_BENCH_WRITE_PMC="$OUT/pmc_json" python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r05_bench.json" 2> "$OUT/r05_bench.err"
CMD="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --sample-parity"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- $CMD > "$OUT/kt.log" 2>&1
PCMD="python $ROOT/bench.py --pmc-child"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $PCMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $PCMD > "$OUT/write.log" 2>&1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/sq$i" -o sq$i -- $PCMD > "$OUT/sq$i.log" 2>&1
done
cd "$ROOT"
python profiles/summarize.py "$OUT" r05 > "$OUT/summary_r05.log" 2>&1
python tests/gpu_microbench.py 18 2>&1 | grep -v amdgpu.ids > "$OUT/r05_microbench.txt"
python tests/gpu_microbench.py 20 2>&1 | grep -v amdgpu.ids > "$OUT/r05_microbench_2p20.txt"
CIRCL_LATENCY_ALL=1 python tests/gpu_microbench.py 0 latency 2>&1 | grep -v amdgpu.ids > "$OUT/r05_latency.txt"
{ for p in 65 44 87; do python tools/sign_rate.py $p 18 4; done; python tools/sign_rate.py 65 16 4; } 2>&1 | grep "ML-DSA" > "$OUT/r05_sign_rates.txt"
{ for p in 44 65 87; do python tools/dsa_latency.py $p; done; python tools/dsa_sign_small.py 65; } 2>&1 | grep -v amdgpu.ids > "$OUT/r05_dsa_latency.txt"
tail -5 "$OUT/summary_r05.log"; head -c 600 "$OUT/r05_bench.json"; echo; tail -3 "$OUT/r05_bench.err"; cat "$OUT/r05_sign_rates.txt"
