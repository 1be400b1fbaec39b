#!/bin/bash
# Builds the native measurement tools of tools/ into tools/bin/ (git-ignored; travels to the GPU box with the tree).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
g++ -std=c++17 -O2 -pthread -Iinclude tools/concurrent_bench.cpp -Lcircl_amd -lcirclhip \
    -Wl,-rpath,'$ORIGIN/../../circl_amd' -Wl,-rpath,/opt/rocm/lib -o tools/bin/concurrent_bench
echo built tools/bin/concurrent_bench
# the phase-ablation build of mldsa_verify_kernel (tools/verify_phases.sh)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icircl_amd/csrc -Iinclude tools/ablate_dsa.hip -o tools/bin/ablate_dsa
echo built tools/bin/ablate_dsa
# the headline kernel with the shader clock read at its phase boundaries (tools/gpu_round.sh kem_clocks), with and without the ring phase's priority
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icircl_amd/csrc -Iinclude tools/clocks_kem.hip -o tools/bin/clocks_kem &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icircl_amd/csrc -Iinclude -DCIRCL_KEM_RING_PRIO=0 tools/clocks_kem.hip -o tools/bin/clocks_kem_prio0 &
wait
echo built tools/bin/clocks_kem tools/bin/clocks_kem_prio0
# the same tool with the verify kernel's build-time variants, for A/B runs on one box (tools/gpu_round.sh verify_variants)
if [ "${VARIANTS:-0}" = 1 ]; then
  for v in "w5:-DCIRCL_DSA_WAVES_PER_EU=5" "prio0:-DCIRCL_DSA_VERIFY_PRIO=0" "prio3:-DCIRCL_DSA_VERIFY_PRIO=3"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icircl_amd/csrc -Iinclude ${v#*:} tools/ablate_dsa.hip -o tools/bin/ablate_dsa_${v%%:*} &
  done
  wait
  echo built tools/bin/ablate_dsa_{w5,prio0,prio3}
fi
