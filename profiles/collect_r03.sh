#!/bin/bash
# Round-3 evidence, collected on the GPU box (run through gpurun); raw output stays in gpurun_out/prof_r03 (scratch), the
# condensed files are copied into profiles/ by hand afterwards.
#   1. bench.py as the driver runs it (live PMC passes inside)          -> r03_bench.json
#   2. rocprofv3 --kernel-trace --stats of a bench.py run with every config -> r03_kernel_stats.txt
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* of the headline kernels (own passes) -> r03_pmc.txt, traffic.json, valu.json
#   4. device-resident microbench of every operation at 2^18 and at the BASELINE sizes -> r03_microbench*.txt
#   5. host-buffer path (page-locked / pageable) and the PCIe probe behind its design -> r03_host_path.txt, r03_pcie_probe.txt
#   6. per-round trace of one batch signing call -> r03_sign_trace.txt
#   7. SURVEY 8(f) rows f2 / f4: hybrid KEMs + X25519 and the XOF / K12 service: rates and rocprofv3 kernel stats
#      -> r03_hybrid.txt, r03_xof.txt, r03_f2_f4_kernel_stats.txt
#   8. SQ counters of the ML-DSA verify kernel and of the signing round kernels -> r03_pmc_mldsa.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r03
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CIRCL_BENCH_WRITE_PMC="$OUT/pmc_json" python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r03_bench.json" 2> "$OUT/r03_bench.err"
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
python profiles/summarize.py "$OUT" r03 > "$OUT/summary_r03.log" 2>&1
python tests/gpu_microbench.py 18 2>&1 | grep -v amdgpu.ids > "$OUT/r03_microbench.txt"
python tests/gpu_microbench.py 20 2>&1 | grep -v amdgpu.ids > "$OUT/r03_microbench_2p20.txt"
CIRCL_LATENCY_ALL=1 python tests/gpu_microbench.py 0 latency 2>&1 | grep -v amdgpu.ids > "$OUT/r03_latency.txt"
{ for p in 44 65 87; do python tools/dsa_latency.py $p; done; python tools/dsa_sign_small.py 65; } 2>&1 | grep -v amdgpu.ids > "$OUT/r03_dsa_latency.txt"
python tools/hbm_probe.py 2>&1 | grep -v amdgpu.ids > "$OUT/r03_hbm_probe.txt"
bash tools/kem_params_check.sh 2>&1 | grep -v amdgpu.ids > "$OUT/r03_latency_params.txt"
python tools/hybrid_latency.py 2>&1 | grep -v amdgpu.ids > "$OUT/r03_hybrid_latency.txt"
python tools/table_latency.py 2>&1 | grep -v amdgpu.ids > "$OUT/r03_table_latency.txt"
python tools/host_small.py 2>&1 | grep -v amdgpu.ids > "$OUT/r03_host_small.txt"
{ python tools/host_path.py 20; CIRCL_HIP_HOST_AHEAD=0 python tools/host_path.py 20; CIRCL_HIP_HOST_CHUNK=14 python tools/host_path.py 20; CIRCL_HIP_HOST_CHUNK=16 python tools/host_path.py 20; CIRCL_HIP_HOST_THREADS=8 python tools/host_path.py 20; } 2>&1 | grep -v amdgpu.ids > "$OUT/r03_host_path.txt"
tools/bin/pcie_probe > "$OUT/r03_pcie_probe.txt" 2>&1
tools/sign_trace.sh 65 18 > "$OUT/r03_sign_trace.txt" 2>&1
python tools/hybrid_bench.py 20 2>&1 | grep -v amdgpu.ids > "$OUT/r03_hybrid.txt"
python tools/xof_bench.py 2>&1 | grep -v amdgpu.ids > "$OUT/r03_xof.txt"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/f2" -o f2 -- python $ROOT/tools/hybrid_bench.py 18 > "$OUT/f2.log" 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/f4" -o f4 -- python $ROOT/tools/xof_bench.py > "$OUT/f4.log" 2>&1 )
python - "$OUT" > "$OUT/r03_f2_f4_kernel_stats.txt" <<'PY'
import csv, glob, sys
for tag, what in (("f2", "tools/hybrid_bench.py 18 (X25519, X-Wing, X25519MLKEM768)"), ("f4", "tools/xof_bench.py (SHAKE128 batch, KangarooTwelve)")):
    print("== rocprofv3 --kernel-trace --stats of", what)
    for f in glob.glob(sys.argv[1] + "/" + tag + "/**/*kernel_stats.csv", recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
        for r in rows[:14]:
            print(f"  {r['Name'].split('(')[0][:90]:90s} calls {int(r['Calls']):5d}  total {float(r['TotalDurationNs'])/1e6:9.3f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
PY
{ AGG=max bash tools/pmc_any.sh sign_ python $ROOT/tools/sign_only.py 65 17; bash tools/pmc_any.sh mldsa_verify_kernel python $ROOT/tools/verify_only.py 65 18; } 2>&1 | grep -v amdgpu.ids > "$OUT/r03_pmc_mldsa.txt"
#   9. (round 3) message-length sweep of ML-DSA, signing round-mode sweep, ablation probes (<= 64-VGPR Keccak, rotation encodings),
#      BASELINE configs[2] in its stated shape over 8 logical devices, the sanitizer runs
python tools/msglen_bench.py 14 2>&1 | grep -v amdgpu.ids > "$OUT/r03_msglen.txt"
{ for pair in 0 1; do CIRCL_HIP_SIGN_PAIR=$pair python tools/sign_rate.py 65 18 4; done; for p in 44 87; do python tools/sign_rate.py $p 18 4; done; } 2>&1 | grep "ML-DSA" > "$OUT/r03_sign_rates.txt"
{ tools/bin/ablate; tools/bin/ablate_w8; } > "$OUT/r03_ablation.txt" 2>&1
CIRCL_HIP_LOGICAL_DEVICES=8 python tests/logical_worker.py config3 23 2>&1 | grep -v amdgpu.ids > "$OUT/r03_config3_stated_shape.json"
python -m pytest tests/test_gpu_sanitizers.py -q 2>&1 | tail -3 > "$OUT/r03_sanitizers.txt"
tail -5 "$OUT/summary_r03.log"; head -c 600 "$OUT/r03_bench.json"; echo; cat "$OUT/r03_host_path.txt"
