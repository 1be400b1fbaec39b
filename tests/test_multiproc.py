"""N>1 control flow on CPU: two gloo ranks run the same sharding / barrier / max-over-ranks /
whole-job aggregation code that bench.py uses with RCCL (no data-path collective exists)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_exactly():
    from circl_amd.parallel import shard_bounds
    for n in (0, 1, 7, 8, 1 << 20, (1 << 23) + 5):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def test_two_rank_gloo_aggregation():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    prog = textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %r)
        from circl_amd import parallel
        r = parallel.Ranks("gloo")
        lo, hi = parallel.shard_bounds(1001, r.world, r.rank)
        r.barrier()
        elapsed = 1.0 + r.rank          # rank 1 is the slow one
        value, worst = parallel.whole_job_rate(r, hi - lo, elapsed)
        r.barrier()
        if r.rank == 0:
            print(json.dumps({"value": value, "worst": worst, "world": r.world}))
        r.close()
    """ % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", prog], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["world"] == 2 and res["worst"] == 2.0 and abs(res["value"] - 1001 / 2.0) < 1e-9


def test_bench_py_refuses_world_size_mismatch():
    # bench.py under a launcher whose WORLD_SIZE differs from --gpus must fail before anything is timed (rc 3), GPU or not
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--no-extras", "--no-pmc"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "refusing" in r.stderr, (r.returncode, r.stderr[-500:])
