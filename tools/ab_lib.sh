#!/bin/bash
# Not a test: bench.py's headline (and decaps) under library variants from tools/bin (tools/variant_lib.sh), alternating with the
# current build on ONE box:   tools/ab_lib.sh <variant> [...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp circl_amd/libcirclhip.so tools/bin/libcirclhip_cur.so
for rep in 1 2; do
  for v in cur "$@"; do
    cp tools/bin/libcirclhip_$v.so circl_amd/libcirclhip.so
    if [ "${MODE:-encaps}" = "encaps" ]; then
      python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'encaps/s %.4e' % d['value'], 'encrypt ms %.3f' % d['roofline']['avg_launch_ms'], d['parity']['bit_exact_vs_oracle'])"
    else
      python bench.py --mode $MODE --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['configs']['$MODE']; print('$v', '$MODE %.4e' % d['value'], c['kernel_ms_per_step'], c['parity']['bit_exact_vs_oracle'] if 'bit_exact_vs_oracle' in c['parity'] else c['parity'])"
    fi
  done
done
cp tools/bin/libcirclhip_cur.so circl_amd/libcirclhip.so
