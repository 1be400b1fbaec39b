#!/bin/bash
# A/B on one GPU box: build/libcirclhip_old.so vs the current circl_amd/libcirclhip.so, alternating runs.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp circl_amd/libcirclhip.so build/libcirclhip_new.so
for rep in 1 2 3; do
  for v in old new; do
    cp build/libcirclhip_$v.so circl_amd/libcirclhip.so
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'encaps/s %.4e' % d['value'], 'encrypt ms %.3f' % d['roofline']['avg_launch_ms'], d['parity']['bit_exact_vs_oracle'])"
    [ -n "${AB_EXTRA:-}" ] && python tests/gpu_microbench.py 18 2>&1 | grep -E "$AB_EXTRA" | sed "s/^/   $v /"
  done
done
cp build/libcirclhip_new.so circl_amd/libcirclhip.so
