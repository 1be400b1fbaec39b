#!/bin/bash
# SQ-level counters for the encrypt kernel (own passes, kernel-trace only): where do the issue cycles go?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -o g$i -- $CMD > "$OUT/g$i.log" 2>&1
  tail -2 "$OUT/g$i.log" | cut -c1-300
done
cd "$ROOT" && python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sq/g*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mlkem" not in k: continue
        k = k.split("(")[0][:90]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
import json
out = {}
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v); print(f"   {c:28s} median {v[len(v)//2]:.4e}  n={len(v)}")
    if "mlkem_encrypt_kernel<3, 0, 0, true, 0>" in k:  # the distinct-key kernel (not bench.py's shared-key steps)
        med = lambda c: sorted(d[c])[len(d[c]) // 2]
        out = {"mlkem768_encrypt_valu_insts_per_launch_2p20": med("SQ_INSTS_VALU"), "salu": med("SQ_INSTS_SALU"), "lds": med("SQ_INSTS_LDS"),
               "grbm_gui_active_per_xcd": med("GRBM_GUI_ACTIVE") / 8, "wave_quad_cycles": med("SQ_WAVE_CYCLES"),
               "method": "rocprofv3 --pmc SQ_* (own passes, kernel-trace only), median over the launches of bench.py --steps 2; GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_WAVE_CYCLES counts in units of 4 cycles"}
json.dump(out, open("gpurun_out/sq/valu.json", "w"), indent=1)
PY
