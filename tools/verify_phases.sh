#!/bin/bash
# tools/verify_phases.sh [65|87] [log2 n] -- where mldsa_verify_kernel's time goes, phase by phase.
# tools/bin/ablate_dsa phases runs the kernel with phase subsets compiled out (template ABLATE masks: the kernel name carries the mask);
# three rocprofv3 --pmc passes (kernel-trace only, one counter group each) give every variant's VALU / LDS / VMEM instruction counts, its
# wave-cycle split and the clock it really ran at (GRBM_GUI_ACTIVE / 8 XCDs / duration); the same passes over the library's VALU probe
# kernels give the rates (and the clock) the live mix ceiling of bench.py is built from.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; MODE=${1:-65}; LOGN=${2:-18}
OUT=$ROOT/gpurun_out/verify_phases_$MODE; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_BRANCH"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/a$i" -o a$i -- "$ROOT/tools/bin/ablate_dsa" phases $MODE $LOGN > "$OUT/a$i.log" 2>&1 )
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -o p$i -- python -c "
import sys; sys.path.insert(0, '$ROOT')
from circl_amd import device as cdev
print(cdev.valu_probe(0, 4))" > "$OUT/p$i.log" 2>&1 )
done
grep -h "mask\|==" "$OUT/a1.log"
cd "$ROOT" && python - "$OUT" "$MODE" "$LOGN" <<'PY'
import csv, glob, collections, sys, re
out, mode, logn = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = 1 << logn
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/*1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
names = {0: "full", 6: "phase A (ExpandA)", 14: "phase A, no row stores", 5: "phase 1 (decode, z-hat, c-hat)", 3: "phases 2+3 (A z, t1, w1)", 1: "phases 1+2+3", 7: "nothing"}
rows = {}
for k in cnt:
    m = re.search(r"mldsa_verify_kernel<(\d+), (\d+), 0>", k)
    if m and int(m.group(1)) == mode:
        rows[int(m.group(2))] = k
    elif "rate_probe" in k:
        rows[k.split("::")[-1].split("<")[0]] = k
print("\n%-34s %8s %11s %9s %7s %9s %9s | %6s %6s %6s | %9s %9s %8s" % ("variant", "ms", "VALU/item", "c/i@2.4", "GHz", "c/i real", "LDS/item", "active", "w_inst", "w_any", "VMEM rd", "VMEM wr", "bankconf"))
base = {}
for key in [0, 6, 14, 5, 3, 1, 7] + [k for k in rows if isinstance(k, str)]:
    if key not in rows: continue
    k = rows[key]; c = {a: med(b) for a, b in cnt[k].items()}; t = med(dur[k]) * 1e-9
    valu = c.get("SQ_INSTS_VALU", float("nan")); ghz = c.get("GRBM_GUI_ACTIVE", float("nan")) / 8 / t / 1e9
    per = n if not isinstance(key, str) else 1
    wc = c.get("SQ_WAVE_CYCLES", float("nan"))
    print("%-34s %8.3f %11.1f %9.2f %7.3f %9.2f %9.1f | %6.3f %6.3f %6.3f | %9.3e %9.3e %8.3f" % (
        names.get(key, key), t * 1e3, valu / per, 1024 * 2.4e9 * t / valu, ghz, 1024 * ghz * 1e9 * t / valu, c.get("SQ_INSTS_LDS", 0) / per,
        c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_WAIT_ANY", 0) / wc,
        c.get("SQ_INSTS_VMEM_RD", 0), c.get("SQ_INSTS_VMEM_WR", 0), c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
    base[key] = (t, valu)
if 0 in base and 6 in base and 5 in base and 3 in base and 7 in base:
    t0, v0 = base[0]; z = base[7]
    print("\nsum of the parts: A %.3f + 1 %.3f + 2/3 %.3f - 2 x nothing %.3f = %.3f ms against full %.3f ms;  VALU %.0f + %.0f + %.0f = %.0f against %.0f per item" % (
        base[6][0] * 1e3, base[5][0] * 1e3, base[3][0] * 1e3, z[0] * 1e3, (base[6][0] + base[5][0] + base[3][0] - 2 * z[0]) * 1e3, t0 * 1e3,
        base[6][1] / n, base[5][1] / n, base[3][1] / n, (base[6][1] + base[5][1] + base[3][1] - 2 * z[1]) / n, v0 / n))
PY
