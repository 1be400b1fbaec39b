ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/prof_r04b; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/full.log 2>&1; tail -3 $OUT/full.log
python bench.py --steps 20 --warmup 5 > $OUT/r04_bench.json 2> $OUT/r04_bench.err
CIRCL_LATENCY_ALL=1 python tests/gpu_microbench.py 0 latency 2>&1 | grep -v amdgpu.ids > $OUT/r04_latency.txt
{ echo "default (resident-key calls up to 2^10 items, unparsed keys up to 2^9: one launch)"; python tools/table_latency.py; echo "CIRCL_HIP_KEM_CHAIN=0 CIRCL_HIP_KEM_CHAIN_ENCAPS=0 CIRCL_HIP_KEM_CHAIN_ITEM=0 CIRCL_HIP_SIGN_COOP_LOG2=0 (the round-3 routes), same box"; CIRCL_HIP_KEM_CHAIN=0 CIRCL_HIP_KEM_CHAIN_ENCAPS=0 CIRCL_HIP_KEM_CHAIN_ITEM=0 CIRCL_HIP_SIGN_COOP_LOG2=0 python tools/table_latency.py | head -4; } 2>&1 | grep -v amdgpu.ids > $OUT/r04_table_latency.txt
{ python tools/host_small.py; } 2>&1 | grep -v amdgpu.ids > $OUT/r04_host_small.txt
{ for p in 44 65 87; do python tools/dsa_latency.py $p; done; python tools/dsa_sign_small.py 65; } 2>&1 | grep -v amdgpu.ids > $OUT/r04_dsa_latency.txt
python -c "
import json; d=json.load(open('$OUT/r04_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['valu']['frac_of_mix_ceiling']['at_4_waves_per_simd'])"
cat $OUT/r04_host_small.txt; head -4 $OUT/r04_table_latency.txt | cut -c1-200
