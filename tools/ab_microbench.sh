#!/bin/bash
# tools/ab_microbench.sh <libA.so> <libB.so> <grep pattern> [rounds] [logn] -- two builds of the library alternating on ONE box: the lines of
# tests/gpu_microbench.py (device-resident rates of every batch operation at 2^logn) that match the pattern; the original library is put back.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
A=$1; B=$2; PAT=$3; ROUNDS=${4:-3}; LOGN=${5:-18}
cp circl_amd/libcirclhip.so /tmp/libcirclhip_keep.so
for r in $(seq 1 $ROUNDS); do
  for v in "$A" "$B"; do
    cp "$v" circl_amd/libcirclhip.so
    python tests/gpu_microbench.py $LOGN 2>&1 | grep -E "$PAT" | sed "s/^/round $r $(basename $v): /"
  done
done
cp /tmp/libcirclhip_keep.so circl_amd/libcirclhip.so
