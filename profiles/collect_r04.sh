#!/bin/bash
# Round-4 evidence, collected on the GPU box (run through gpurun); raw output stays in gpurun_out/prof_r04 (scratch), the
# condensed files are copied into profiles/ afterwards.
#   1. bench.py as the driver runs it (live PMC passes inside)                    -> r04_bench.json
#   2. rocprofv3 --kernel-trace --stats of a bench.py run with every config        -> r04_kernel_stats.txt
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* of the headline kernels      -> r04_pmc.txt, traffic.json, valu.json
#   4. device-resident microbench at 2^18 / 2^20, small-batch latency sweep        -> r04_microbench*.txt, r04_latency.txt
#   5. resident keys: cost per call (one launch up to 2^10 items), host small calls -> r04_table_latency.txt, r04_host_small.txt
#   6. batch signing: rates of every parameter set, SQ counters of the round kernels, ML-DSA latencies
#                                                                                  -> r04_sign_rates.txt, r04_sign_pmc.txt, r04_dsa_latency.txt
# (unchanged since round 3 and not re-collected: hbm / pcie probes, host path sweep, hybrids, XOF, message-length sweep)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r04
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CIRCL_BENCH_WRITE_PMC="$OUT/pmc_json" python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r04_bench.json" 2> "$OUT/r04_bench.err"
CMD="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc"
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
python profiles/summarize.py "$OUT" r04 > "$OUT/summary_r04.log" 2>&1
python tests/gpu_microbench.py 18 2>&1 | grep -v amdgpu.ids > "$OUT/r04_microbench.txt"
python tests/gpu_microbench.py 20 2>&1 | grep -v amdgpu.ids > "$OUT/r04_microbench_2p20.txt"
CIRCL_LATENCY_ALL=1 python tests/gpu_microbench.py 0 latency 2>&1 | grep -v amdgpu.ids > "$OUT/r04_latency.txt"
{ echo "default (resident-key calls up to 2^10 items, unparsed keys up to 2^9: one launch)"; python tools/table_latency.py; echo "CIRCL_HIP_KEM_CHAIN=0 CIRCL_HIP_KEM_CHAIN_ENCAPS=0 (the round-3 routes), same box"; CIRCL_HIP_KEM_CHAIN=0 CIRCL_HIP_KEM_CHAIN_ENCAPS=0 python tools/table_latency.py | head -4; } 2>&1 | grep -v amdgpu.ids > "$OUT/r04_table_latency.txt"
{ python tools/host_small.py; CIRCL_HIP_KEM_CHAIN=0 CIRCL_HIP_KEM_CHAIN_ENCAPS=0 python tools/host_small.py; } 2>&1 | grep -v amdgpu.ids > "$OUT/r04_host_small.txt"
{ for p in 65 44 87; do python tools/sign_rate.py $p 18 4; done; python tools/sign_rate.py 65 16 4; CIRCL_HIP_SIGN_PAIR=1 python tools/sign_rate.py 65 18 4; } 2>&1 | grep "ML-DSA" > "$OUT/r04_sign_rates.txt"
{ for p in 44 65 87; do python tools/dsa_latency.py $p; done; python tools/dsa_sign_small.py 65; } 2>&1 | grep -v amdgpu.ids > "$OUT/r04_dsa_latency.txt"
{ AGG=max bash tools/pmc_any.sh sign_ python $ROOT/tools/sign_only.py 65 17; bash tools/pmc_any.sh mldsa_verify_kernel python $ROOT/tools/verify_only.py 65 18; } 2>&1 | grep -v amdgpu.ids > "$OUT/r04_sign_pmc.txt"
tail -5 "$OUT/summary_r04.log"; head -c 400 "$OUT/r04_bench.json"; echo; cat "$OUT/r04_table_latency.txt" | cut -c1-130; cat "$OUT/r04_sign_rates.txt"
