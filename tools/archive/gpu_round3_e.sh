#!/bin/bash
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/r3e_counters.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -o "Name:\s*[A-Za-z0-9_]*" gpurun_out/r3e_counters.txt | awk '{print $2}' | sort -u > gpurun_out/r3e_counter_names.txt
grep -i "UTCL\|LATENCY\|LEVEL\|WAIT" gpurun_out/r3e_counter_names.txt | tr '\n' ' ' | head -c 3000
