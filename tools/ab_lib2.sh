#!/bin/bash
# tools/ab_lib2.sh <libA.so> <libB.so> [rounds] -- A/B of two builds of the library on ONE box, alternating: the headline figure of bench.py
# (--no-extras: the headline alone, sampled parity) with each library in place of circl_amd/libcirclhip.so; the original is put back.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
A=$1; B=$2; ROUNDS=${3:-3}
cp circl_amd/libcirclhip.so /tmp/libcirclhip_keep.so
for r in $(seq 1 $ROUNDS); do
  for v in "$A" "$B"; do
    cp "$v" circl_amd/libcirclhip.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extras --sample-parity --extras-file /tmp/ab_extras.json 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('round $r $(basename $v): encaps/s %.4e  ms/step %.3f  encrypt kernel ms %.3f  parity %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['parity']['bit_exact_vs_oracle']))"
  done
done
cp /tmp/libcirclhip_keep.so circl_amd/libcirclhip.so
