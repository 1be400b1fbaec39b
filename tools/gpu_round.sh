#!/bin/bash
# tools/gpu_round.sh <step> [<step> ...] -- every GPU-side evidence command of round 6 behind ONE script (26 one-shot
# gpu_r05_[a-z]*.sh made it impossible to tell which command produced which file).  Run through gpurun from the repo root:
#
#     gpurun --timeout 900 -- 'bash tools/gpu_round.sh bench verify_pmc'
#
# Each step writes gpurun_out/r06/<file>, whose FIRST LINE names the command that made it; the files worth keeping are copied to
# profiles/r06_<file> by hand (same name).  Steps:
#
#   bench        bench.py exactly as the driver runs it (N = 1): the contract line + bench_extras.json        -> bench.json, bench_extras.json
#   stats        rocprofv3 --kernel-trace --stats of a bench.py run                                            -> kernel_stats.txt
#   verify_pmc   SQ / TCC counters of mldsa_verify_kernel<65> (2^18) and <87> (2^16)                         -> verify_pmc.txt
#   verify_phases  phase ablation of mldsa_verify_kernel + the VALU probe under --pmc: per-phase cycles per instruction -> verify_phases.txt
#   verify_clocks  the FULL verify kernel with the shader clock read at its phase boundaries (tools/bin/ablate_dsa clocks)  -> verify_clocks.txt
#   kem_clocks   the headline kernel with the shader clock read at its phase boundaries (tools/bin/clocks_kem)             -> kem_clocks.txt
#   tests        the whole GPU suite + smoke()                                                                  -> gpu_tests.log
#   async        tools/bin/concurrent_bench --async: R reactor threads x W outstanding one-item requests       -> async.txt
#   one_call     where the microseconds of ONE one-item table call go (library's own timestamps)               -> one_call.txt
#   routes       tools/route_check.py: both sides of every route cut-over on this box                          -> routes.txt
#   verify_ab    A/B of mldsa_verify_kernel variants (CIRCL_HIP_DSA_VERIFY_* knobs), alternating               -> verify_ab.txt
#   concurrent   the blocking coalescer's T-thread table (round 5's measurement, for comparison)               -> concurrent.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=$ROOT/gpurun_out/r06; mkdir -p "$OUT"
export TMPDIR=/tmp
T0=$SECONDS
note() { echo "[gpu_round +$((SECONDS - T0))s] $*"; }
hdr() { echo "# command: $1"; echo "# box: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2 | xargs), $(nproc) CPUs usable; lib sha256 $(sha256sum circl_amd/libcirclhip.so | cut -c1-16); $(date -u +%FT%TZ)"; }

step_bench() {
  local cmd="python bench.py --gpus 1 --steps 20 --warmup 5"
  ( cd "$ROOT" && timeout 600 $cmd > "$OUT/bench.stdout" 2> "$OUT/bench.err" ); local rc=$?
  tail -n 1 "$OUT/bench.stdout" > "$OUT/bench.json"
  cp -f bench_extras.json "$OUT/bench_extras.json" 2>/dev/null
  note "bench rc=$rc, line $(wc -c < "$OUT/bench.json") bytes, stdout lines $(wc -l < "$OUT/bench.stdout")"
  python - <<'PY'
import json
def bad(c): raise ValueError(c)
try:
    d = json.loads(open("gpurun_out/r06/bench.json").read(), parse_constant=bad)
    r = d["roofline"]
    print("value %.4e ms/step %.3f frac %.4f mix %s traffic/alg %s cpu %.3e (%s, %d cores) host_abi %s wall %.0f s" % (
        d["value"], d["ms_per_step"], r["frac"], (r.get("valu") or {}).get("frac_of_mix_ceiling"), r.get("traffic_over_algorithmic"),
        d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["cores"], d.get("value_host_abi"), d["bench_wall_s"]))
    print("parity", d["parity"]); print("configs", json.dumps(d.get("configs")))
except Exception as e:
    print("NO PARSABLE LINE:", repr(e))
PY
  tail -n 3 "$OUT/bench.err"
}

step_stats() {
  local cmd="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --sample-parity"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- $cmd > "$OUT/kt.log" 2>&1 )
  python profiles/summarize.py "$OUT" r06 > "$OUT/summarize.log" 2>&1
  { hdr "rocprofv3 --kernel-trace --stats --output-format csv -- $cmd  (condensed by profiles/summarize.py)"; cat "$OUT/r06_kernel_stats.txt"; } > "$OUT/kernel_stats.txt"
  note "stats: $(wc -l < "$OUT/kernel_stats.txt") lines"; head -n 30 "$OUT/kernel_stats.txt"
}

step_verify_pmc() {
  { hdr "tools/pmc_any.sh mldsa_verify_kernel python tools/verify_only.py {65 18 | 87 16}   (one rocprofv3 --pmc pass per counter group, kernel-trace only)"
    for pn in "65 18" "87 16"; do
      echo "== ML-DSA-$pn"
      GROUPS_OVERRIDE="SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA|TCC_HIT_sum TCC_MISS_sum|TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
        timeout 900 bash tools/pmc_any.sh mldsa_verify_kernel python "$ROOT/tools/verify_only.py" $pn 2>&1 | grep -v amdgpu.ids
    done; } > "$OUT/verify_pmc.txt"
  note "verify_pmc"; cat "$OUT/verify_pmc.txt"
}

step_verify_phases() {
  { hdr "tools/verify_phases.sh 65 18; tools/verify_phases.sh 87 16   (phase ablation of mldsa_verify_kernel under rocprofv3 --pmc, three counter groups)"
    timeout 600 bash tools/verify_phases.sh 65 18 2>&1 | grep -v amdgpu.ids
    timeout 600 bash tools/verify_phases.sh 87 16 2>&1 | grep -v amdgpu.ids; } > "$OUT/verify_phases.txt"
  note "verify_phases"; cat "$OUT/verify_phases.txt"
}

step_verify_clocks() {
  { hdr "tools/bin/ablate_dsa clocks {65 18 | 87 16 }   (mldsa_verify_kernel<MODE, 64 | ...>: s_memtime at every phase boundary, summed per workgroup)"
    timeout 300 tools/bin/ablate_dsa clocks 65 18 2>&1 | grep -v amdgpu.ids
    timeout 300 tools/bin/ablate_dsa clocks 87 16 2>&1 | grep -v amdgpu.ids; } > "$OUT/verify_clocks.txt"
  note "verify_clocks"; cat "$OUT/verify_clocks.txt"
}

step_kem_clocks() {
  { hdr "tools/bin/clocks_kem 20; tools/bin/clocks_kem_prio0 20   (mlkem_encrypt_kernel<K, ENCAPS, 8>: s_memtime at the phase boundaries, summed per workgroup)"
    timeout 300 tools/bin/clocks_kem 20 2>&1 | grep -v amdgpu.ids
    timeout 300 tools/bin/clocks_kem_prio0 20 2>&1 | grep -v amdgpu.ids; } > "$OUT/kem_clocks.txt"
  note "kem_clocks"; cat "$OUT/kem_clocks.txt"
}

step_verify_variants() {
  # build-time variants of the verify kernel, alternating on this box (VARIANTS=1 bash tools/build_tools.sh built them)
  { hdr "for r in 1 2 3: tools/bin/ablate_dsa{,_prio0,_prio1,_prio3} clocks {65 18|87 16} 16; tools/bin/ablate_dsa_w5 clocks ... 20   (row 'full' of each)"
    for r in 1 2 3; do
      for pn in "65 18" "87 16"; do
        for v in "" _prio0 _prio1 _prio3 _w5; do
          [ -x tools/bin/ablate_dsa$v ] || continue
          bpc=16; [ "$v" = _w5 ] && bpc=20
          echo -n "round $r ML-DSA-$pn variant '${v:-base}' ($bpc per CU): "
          timeout 120 tools/bin/ablate_dsa$v clocks $pn $bpc 2>&1 | grep "^full  " | cut -c1-230
        done
      done
    done; } > "$OUT/verify_variants.txt"
  note "verify_variants"; cat "$OUT/verify_variants.txt"
}

step_tests() {
  sha256sum circl_amd/libcirclhip.so > "$OUT/lib.sha256"
  { hdr "python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12; python -c 'import __graft_entry__ as g; g.smoke()'"
    timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=8 --durations=12 2>&1; echo "pytest rc=$?"
    timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 3; } > "$OUT/gpu_tests.log"
  note "tests"; tail -n 25 "$OUT/gpu_tests.log"
}

step_async() {
  { hdr "tools/bin/concurrent_bench --async <op> <reactors> <window> <seconds>   (R reactor threads, W one-item requests outstanding each)"
    for op in encaps decaps; do
      for rw in "1 64" "1 256" "2 128" "4 64" "4 128" "4 256" "4 1024"; do
        timeout 60 tools/bin/concurrent_bench --async $op $rw 2.0 2>&1 | grep -v amdgpu.ids
      done
    done
    echo "# ML-DSA-65 verification through a resident public-key table (circl_hip_mldsa_verify_table_submit)"
    for rw in "1 64" "4 64" "4 128"; do
      timeout 60 tools/bin/concurrent_bench --async verify $rw 2.0 2>&1 | grep -v amdgpu.ids
    done
    echo "# keys that come WITH the call (circl_hip_queue: every item brings its own key -- a TLS server's encapsulation to the client's ephemeral key)"
    for op in encaps_call decaps_call; do
      for rw in "1 256" "4 64" "4 128" "4 256"; do
        timeout 60 tools/bin/concurrent_bench --async $op $rw 2.0 2>&1 | grep -v amdgpu.ids
      done
    done; } > "$OUT/async.txt"
  note "async"; cat "$OUT/async.txt"
}

step_one_call() {
  { hdr "tools/bin/concurrent_bench --one-call   (circl_hip_profile_call_stamps: host timestamps of one one-item table call, medians of 2000)"
    timeout 120 tools/bin/concurrent_bench --one-call 2>&1 | grep -v amdgpu.ids; } > "$OUT/one_call.txt"
  note "one_call"; cat "$OUT/one_call.txt"
}

step_routes() {
  { hdr "python tools/route_check.py   (both routes of every cut-over at 1/2, 1, 2 and 4 x its threshold)"
    timeout 900 python tools/route_check.py 2>&1 | grep -v amdgpu.ids; echo "route_check rc=${PIPESTATUS[0]}"; } > "$OUT/routes.txt"
  note "routes"; cat "$OUT/routes.txt"
}

step_verify_ab() {
  { hdr "tools/verify_ab.sh   (variants alternating on one box, 5 rounds)"
    timeout 900 bash tools/verify_ab.sh 2>&1 | grep -v amdgpu.ids; } > "$OUT/verify_ab.txt"
  note "verify_ab"; cat "$OUT/verify_ab.txt"
}

step_concurrent() {
  { hdr "CIRCL_HIP_COALESCE_DONE={0,2} tools/bin/concurrent_bench <op> 256 0 1 2.0 1 16 64   (T blocking callers, one item each, coalesced; completion by stream sync / by a polled flag)"
    for op in encaps decaps; do
      for dm in 0 2; do
        echo "-- $op, CIRCL_HIP_COALESCE_DONE=$dm"
        CIRCL_HIP_COALESCE_DONE=$dm timeout 120 tools/bin/concurrent_bench $op 256 0 1 2.0 1 16 64 2>&1 | grep -v amdgpu.ids
      done
    done; } > "$OUT/concurrent.txt"
  note "concurrent"; cat "$OUT/concurrent.txt"
}

for s in "$@"; do
  if declare -f "step_$s" > /dev/null; then note "== $s"; "step_$s"; else echo "unknown step $s"; fi
done
note "done"
