#!/bin/bash
# usage: tools/pmc_any.sh <tag> <command...> : per-kernel instruction mix and cycles (two PMC passes)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o a -- "$@" > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $OUT/b -o b -- "$@" > $OUT/b.log 2>&1
cd $ROOT; python - <<PY
import csv,collections,glob
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("circl::","")
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(d.items()):
    m={c: sorted(x)[len(x)//2] for c,x in v.items()}
    cyc=m.get("GRBM_GUI_ACTIVE",0)/8
    if cyc < 1e5: continue
    valu=m.get("SQ_INSTS_VALU",0)/1024
    print(f"{k[:58]:58s} n={len(v['SQ_INSTS_VALU']):3d} cyc/XCD {cyc:.3e} VALU/SIMD {valu:.3e} cyc/VALU {cyc/max(valu,1):5.2f} SALU {m.get('SQ_INSTS_SALU',0)/1024:.2e} LDS {m.get('SQ_INSTS_LDS',0)/1024:.2e} VMEM {(m.get('SQ_INSTS_VMEM_RD',0)+m.get('SQ_INSTS_VMEM_WR',0))/1024:.2e} occ(waves/SIMD) {m.get('SQ_WAVE_CYCLES',0)*4/1024/max(cyc,1):.2f} wait_inst {m.get('SQ_WAIT_INST_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1):.2f} lds_conf {m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_ACTIVE_INST_LDS',1),1):.2f}")
PY
