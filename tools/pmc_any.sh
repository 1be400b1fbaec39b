#!/bin/bash
# SQ counters of every kernel matching $1 while running "$2..." under rocprofv3 --pmc (own passes, kernel-trace only)
#   tools/pmc_any.sh mldsa_verify_kernel python tools/verify_only.py 65 18
#   AGG=max ...: the largest invocation of each kernel instead of the median (round kernels of signing: the first round)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; PAT=$1; shift
OUT=$ROOT/gpurun_out/pmc_any; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
# GROUPS="A B|C D" overrides the counter groups (one rocprofv3 pass per |-separated group)
if [ -n "${GROUPS_OVERRIDE:-}" ]; then IFS='|' read -ra GRPS <<< "$GROUPS_OVERRIDE"; else GRPS=(); fi
for grp in "${GRPS[@]:-}" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  ( cd "$ROOT" && cd /tmp && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -o g$i -- "$@" > "$OUT/g$i.log" 2>&1 )
done
cd "$ROOT" && python - "$PAT" <<'PY'
import csv, glob, collections, sys, os
pat = sys.argv[1]
pick = (lambda v: max(v)) if os.environ.get("AGG") == "max" else (lambda v: sorted(v)[len(v) // 2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_any/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat not in k: continue
        agg[k.split("(")[0][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_any/g1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]: dur[r["Kernel_Name"].split("(")[0][:80]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in agg.items():
    print(k, " duration %.3f ms" % (pick(dur[k]) / 1e6 if dur.get(k) else -1), "(%d launches)" % len(dur.get(k, [])))
    med = {c: pick(v) for c, v in d.items()}
    for c, v in sorted(med.items()): print(f"   {c:26s} {v:.4e}")
    if "SQ_INSTS_VALU" in med and dur.get(k):
        t = pick(dur[k]) * 1e-9
        print(f"   -> VALU wave-insts/s {med['SQ_INSTS_VALU'] / t:.3e}; cycles per VALU inst per SIMD at 2.4 GHz: {1024 * 2.4e9 * t / med['SQ_INSTS_VALU']:.2f}")
    if "SQ_WAVE_CYCLES" in med and "SQ_BUSY_CYCLES" in med:
        print(f"   -> resident waves per SIMD ~ {4 * med['SQ_WAVE_CYCLES'] / (med.get('GRBM_GUI_ACTIVE', 0) / 8 * 1024 + 1e-9):.2f} (4 x WAVE_CYCLES / (GUI_ACTIVE per XCD x 1024 SIMDs))")
PY
