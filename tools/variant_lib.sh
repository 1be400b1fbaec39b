#!/bin/bash
# Build a variant of the library with extra -D flags for ONE translation unit:  tools/variant_lib.sh <name> <unit.hip> -DFOO=1 ...
# -> build/libcirclhip_<name>.so (the other objects are the current build's)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); name=$1; unit=$2; shift 2
mkdir -p $ROOT/build/variant_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable "$@" -c $ROOT/circl_amd/csrc/$unit -o $ROOT/build/variant_$name/${unit%.hip}.o 2>&1 | grep -v "occupancy\|warning" || true
objs=""
for u in host_runtime host_coalesce api_mlkem api_mldsa api_prims api_x25519 api_hybrid; do
  if [ "$u.hip" = "$unit" ]; then objs="$objs $ROOT/build/variant_$name/$u.o"; else objs="$objs $ROOT/build/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/libcirclhip_$name.so $objs -lpthread
mkdir -p $ROOT/tools/bin; cp $ROOT/build/libcirclhip_$name.so $ROOT/tools/bin/  # (build/ does not travel to the GPU box, tools/bin does)
echo built build/libcirclhip_$name.so
