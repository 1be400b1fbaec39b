#!/bin/bash
# round 5, what is left of the GPU budget (8 minutes): the bench line with the batch-vectorised CPU baseline (oracle/vec) on the GPU box's
# host, the concurrent-caller figures on the final library, and the kernel-stats / counter passes of profiles/collect_r05.sh on the final library.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/r05last; mkdir -p $OUT/prof
T0=$SECONDS
sha256sum circl_amd/libcirclhip.so > $OUT/lib.sha256
grep -m1 "model name" /proc/cpuinfo > $OUT/cpu.txt; grep -m1 flags /proc/cpuinfo | tr ' ' '\n' | grep -c avx512 >> $OUT/cpu.txt
timeout 60 python -m pytest tests/test_oracle_vec.py -q -p no:cacheprovider 2>&1 | tail -2
CIRCL_BENCH_WRITE_PMC="$OUT/prof/pmc_json" timeout 200 python bench.py --gpus 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$? after $((SECONDS - T0)) s"; tail -3 $OUT/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05last/bench.json").read().strip().splitlines()[-1])
    print("value %.4e ms %.3f frac %.4f mix %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], (d["roofline"].get("valu") or {}).get("frac_of_mix_ceiling")))
    c = d["cpu_baseline"]; print("cpu", c["value"], c["cores"], c.get("vectorized"), (c.get("scalar_oracle") or {}).get("value"))
except Exception as e:
    print("no bench line:", e)
PY
B=tools/bin/concurrent_bench
{ timeout 40 $B encaps 256 0 1 2 1 32 64 96; timeout 30 $B decaps 256 0 1 2 64 128; timeout 30 $B verify 256 0 1 2 64 128; timeout 30 $B sign 256 0 1 2 64 128; } > $OUT/concurrent.txt 2>&1
cut -c1-200 $OUT/concurrent.txt
echo "concurrent done after $((SECONDS - T0)) s"
P=$OUT/prof
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --sample-parity"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/kt" -o kt -- $CMD > "$P/kt.log" 2>&1
echo "kt done after $((SECONDS - T0)) s"
PCMD="python $ROOT/bench.py --pmc-child"
timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$P/fetch" -o fetch -- $PCMD > "$P/fetch.log" 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$P/write" -o write -- $PCMD > "$P/write.log" 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d "$P/sq1" -o sq1 -- $PCMD > "$P/sq1.log" 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d "$P/sq2" -o sq2 -- $PCMD > "$P/sq2.log" 2>&1
cd "$ROOT"
python profiles/summarize.py "$P" r05 > "$P/summary_r05.log" 2>&1
tail -5 "$P/summary_r05.log"
# the raw traces are large: keep the condensed files only
rm -rf "$P/kt" "$P/fetch" "$P/write" "$P/sq1" "$P/sq2"
echo "done after $((SECONDS - T0)) s"
