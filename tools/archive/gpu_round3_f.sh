#!/bin/bash
mkdir -p gpurun_out
export GROUPS_OVERRIDE="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum|VmemLatency|TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum|SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM|LdsLatency"
AGG=max CIRCL_HIP_SIGN_PAIR=0 bash tools/pmc_any.sh sign_ python $GRAFT_REPO_ROOT/tools/sign_only.py 65 17 > gpurun_out/r3f_pmc_sign_single.txt 2>&1
cat gpurun_out/r3f_pmc_sign_single.txt | head -120
