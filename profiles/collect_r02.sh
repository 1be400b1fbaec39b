#!/bin/bash
# Round-2 evidence, collected on the GPU box (run through gpurun); raw output stays in gpurun_out/prof_r02 (scratch), the
# condensed files are copied into profiles/ by hand afterwards.
#   1. bench.py as the driver runs it (live PMC passes inside)          -> r02_bench.json
#   2. rocprofv3 --kernel-trace --stats of a bench.py run with every config -> r02_kernel_stats.txt
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* of the headline kernels (own passes) -> r02_pmc.txt, traffic.json, valu.json
#   4. device-resident microbench of every operation at 2^18 and at the BASELINE sizes -> r02_microbench*.txt
#   5. host-buffer path (page-locked / pageable) and the PCIe probe behind its design -> r02_host_path.txt, r02_pcie_probe.txt
#   6. per-round trace of one batch signing call -> r02_sign_trace.txt
#   7. SURVEY 8(f) rows f2 / f4: hybrid KEMs + X25519 and the XOF / K12 service: rates and rocprofv3 kernel stats
#      -> r02_hybrid.txt, r02_xof.txt, r02_f2_f4_kernel_stats.txt
#   8. SQ counters of the ML-DSA verify kernel and of the signing round kernels -> r02_pmc_mldsa.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r02
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r02_bench.json" 2> "$OUT/r02_bench.err"
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
python profiles/summarize.py "$OUT" r02 > "$OUT/summary_r02.log" 2>&1
python tests/gpu_microbench.py 18 2>&1 | grep -v amdgpu.ids > "$OUT/r02_microbench.txt"
python tests/gpu_microbench.py 20 2>&1 | grep -v amdgpu.ids > "$OUT/r02_microbench_2p20.txt"
python tests/gpu_microbench.py 0 latency 2>&1 | grep -v amdgpu.ids > "$OUT/r02_latency.txt"
{ python tools/host_path.py 20; CIRCL_HIP_HOST_AHEAD=0 python tools/host_path.py 20; CIRCL_HIP_HOST_CHUNK=14 python tools/host_path.py 20; CIRCL_HIP_HOST_CHUNK=16 python tools/host_path.py 20; CIRCL_HIP_HOST_THREADS=8 python tools/host_path.py 20; } 2>&1 | grep -v amdgpu.ids > "$OUT/r02_host_path.txt"
tools/bin/pcie_probe > "$OUT/r02_pcie_probe.txt" 2>&1
tools/sign_trace.sh 65 18 > "$OUT/r02_sign_trace.txt" 2>&1
python tools/hybrid_bench.py 20 2>&1 | grep -v amdgpu.ids > "$OUT/r02_hybrid.txt"
python tools/xof_bench.py 2>&1 | grep -v amdgpu.ids > "$OUT/r02_xof.txt"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/f2" -o f2 -- python $ROOT/tools/hybrid_bench.py 18 > "$OUT/f2.log" 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/f4" -o f4 -- python $ROOT/tools/xof_bench.py > "$OUT/f4.log" 2>&1 )
python - "$OUT" > "$OUT/r02_f2_f4_kernel_stats.txt" <<'PY'
import csv, glob, sys
for tag, what in (("f2", "tools/hybrid_bench.py 18 (X25519, X-Wing, X25519MLKEM768)"), ("f4", "tools/xof_bench.py (SHAKE128 batch, KangarooTwelve)")):
    print("== rocprofv3 --kernel-trace --stats of", what)
    for f in glob.glob(sys.argv[1] + "/" + tag + "/**/*kernel_stats.csv", recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
        for r in rows[:14]:
            print(f"  {r['Name'].split('(')[0][:90]:90s} calls {int(r['Calls']):5d}  total {float(r['TotalDurationNs'])/1e6:9.3f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
PY
{ AGG=max bash tools/pmc_any.sh sign_ python $ROOT/tools/sign_only.py 65 17; bash tools/pmc_any.sh mldsa_verify_kernel python $ROOT/tools/verify_only.py 65 18; } 2>&1 | grep -v amdgpu.ids > "$OUT/r02_pmc_mldsa.txt"
tail -5 "$OUT/summary_r02.log"; head -c 600 "$OUT/r02_bench.json"; echo; cat "$OUT/r02_host_path.txt"
