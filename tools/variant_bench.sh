#!/bin/bash
# Not a test: bench.py's headline under library variants built by tools/variant_lib.sh, alternating with the current build.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp circl_amd/libcirclhip.so build/libcirclhip_cur.so
for rep in 1 2; do
  for v in cur "$@"; do
    cp build/libcirclhip_$v.so circl_amd/libcirclhip.so
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'encaps/s %.4e' % d['value'], 'encrypt ms %.3f' % d['roofline']['avg_launch_ms'], d['parity']['bit_exact_vs_oracle'])"
  done
done
cp build/libcirclhip_cur.so circl_amd/libcirclhip.so
