#!/bin/bash
# What a bench box offers for the reference's own CPU baseline (SURVEY.md 8d option A): run on the GPU box, output to profiles/.
echo "== date"; date -u
echo "== go"; (command -v go && go version) 2>&1 || echo "go: not found"
for p in /usr/local/go/bin/go /usr/lib/go/bin/go /opt/go/bin/go /root/go/bin/go; do [ -x "$p" ] && echo "found $p: $($p version)"; done
echo "== CIRCL_REFERENCE=${CIRCL_REFERENCE:-<unset>}"; [ -n "$CIRCL_REFERENCE" ] && ls "$CIRCL_REFERENCE" | head
echo "== /root/reference"; ls /root/reference 2>&1 | head -3
echo "== other toolchains"; for t in gcc g++ hipcc javac node rustc cargo; do printf "%s: " $t; (command -v $t || echo absent); done
echo "== cpu"; nproc; grep -m1 "model name" /proc/cpuinfo; echo "affinity: $(python3 -c 'import os;print(len(os.sched_getaffinity(0)))')"
cat /sys/fs/cgroup/cpu.max 2>/dev/null
echo "== memory"; free -g | head -2
echo "== gpus"; python3 -c "import torch;print(torch.cuda.device_count(), [torch.cuda.get_device_name(i) for i in range(torch.cuda.device_count())])"
rocm-smi --showproductname 2>/dev/null | head -12
