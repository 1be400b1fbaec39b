"""Round-4 GPU tests: bench.py launches and verifies its own ranks, the RCCL process-group branch executes on a GPU.

SURVEY.md 8(e): n/G items per device, no data-path collective; the shape of the work is the reference's
kem/schemes/schemes_test.go:28-51 (Encapsulate over a scheme)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from benchrec import bench_record

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_py_launches_its_own_eight_ranks(tmp_path):
    # `python bench.py --gpus 8` WITHOUT a launcher: it must start 8 ranks itself and report n_gpus 8, with the strong-scaling
    # block (one batch of --batch items split n/G) beside the weak figure (ranks share this box's GPU, process group on gloo)
    env = dict(os.environ, CIRCL_DIST_BACKEND="gloo", CIRCL_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    xf = str(tmp_path / "extras.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--batch", str(1 << 12),
                        "--mode", "encaps", "--no-extras", "--no-pmc", "--extras-file", xf], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    line, out = bench_record(r, xf)
    # what the driver's SCALE run keeps of an 8-rank line: the strong block and the per-rank rates, inside the 6 KB
    assert line["n_gpus"] == 8 and line["strong"]["items_per_rank"] == [512] * 8 and len(line["per_rank"]["encaps_per_s"]) == 8
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["value"] > 0 and out["parity"]["ranks_failing"] == 0
    st = out["strong"]
    assert st["scaling"] == "strong" and st["batch_total"] == 1 << 12 and st["items_per_rank"] == [512] * 8
    assert st["parity"]["ranks_failing"] == 0 and st["value"] > 0 and len(st["per_rank_encaps_per_s"]) == 8


def test_bench_py_refuses_a_rank_count_it_was_not_asked_for():
    # under a launcher: WORLD_SIZE != --gpus must fail (a SCALE run would otherwise report the wrong n_gpus)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, CIRCL_DIST_BACKEND="gloo", CIRCL_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0", "--batch", "256",
                        "--no-extras", "--no-pmc"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "refusing" in r.stderr
    # without the share-GPU aid: more GPUs asked for than the box has
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("CIRCL_BENCH_SHARE_GPU", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(torch.cuda.device_count() + 1), "--no-extras", "--no-pmc"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 3 and "refusing" in r.stderr


def test_rccl_process_group_branch_runs_on_the_gpu():
    # circl_amd/parallel.py with backend nccl (= RCCL on ROCm): one rank, device-bound process group; barrier, all_reduce (max, sum)
    # and all_gather on GPU tensors -- the only collectives bench.py ever issues (timing barrier and reductions of timings)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    prog = textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %r)
        import torch
        from circl_amd import parallel
        torch.cuda.set_device(0)
        r = parallel.Ranks("nccl", torch.device("cuda", 0))
        assert r.dist is not None and r.backend == "nccl" and r.dist.get_backend() == "nccl"
        t = r._tensor(3.5)
        assert t.is_cuda
        r.barrier()
        v, worst = parallel.whole_job_rate(r, 1000, 2.0)
        g = r.gather(7.25)
        r.barrier()
        print(json.dumps({"value": v, "worst": worst, "gather": g, "max": r.max(4.0), "sum": r.sum(5.0)}))
        r.close()
    """ % ROOT)
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CIRCL_DIST_FORCE_PG="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CIRCL_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-5000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res == {"value": 500.0, "worst": 2.0, "gather": [7.25], "max": 4.0, "sum": 5.0}


def test_resident_decapsulation_chain_route_boundaries():
    # kem/mlkem/mlkem768/kyber.go:103-137 EncapsulateTo / :144-184 DecapsulateTo on parsed keys (encapsulation: one launch, a wavefront per
    # item, up to 2^CIRCL_HIP_KEM_CHAIN_ENCAPS items): up to 2^CIRCL_HIP_KEM_CHAIN items a resident-key decapsulation is
    # ONE launch (a two-wavefront workgroup per item: J beside Decrypt -> G -> PRF -> re-encryption); beyond it the three-launch routes.
    # The same bytes on both sides of the switch and with the route disabled / widened -- implicit rejection (every third ciphertext
    # tampered with), a key whose stored hash is wrong (kem.ErrPrivKey: zeros, status 2), an index vector and none -- and the oracle's.
    prog = textwrap.dedent("""
        import sys, hashlib
        sys.path.insert(0, %r)
        import numpy as np
        from circl_amd import hostapi
        from oracle import orc
        h = hashlib.sha256()
        for param in (512, 768, 1024):
            rng = np.random.default_rng(param)
            nk = 5
            ek, dk = orc.mlkem_keygen(param, rng.integers(0, 256, (nk, 64), dtype=np.uint8))
            dk_bad = dk.copy()
            dk_bad[3, -40] ^= 1
            ek_nc = ek.copy()
            ek_nc[4, 0] = 0xff
            ek_nc[4, 1] |= 0x0f                                    # first coefficient of entry 4 = 0xfff >= q (cpapke.go:45-55)
            pub, prv = hostapi.KeyTable("mlkem-public", param, ek_nc), hostapi.KeyTable("mlkem-private", param, dk_bad)
            for n in (1, 2, 63, 1023, 1024, 1025, 3000):
                m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
                idx = rng.integers(0, nk, n).astype(np.uint32)
                ct, ss, ste = pub.encaps(m, idx)
                ct_o, ss_o, st_o = orc.mlkem_encaps(param, ek_nc[idx], m)   # (entry 4 is not canonical: kem.ErrPubKey, zeros)
                assert (ste == st_o).all() and (ct == ct_o).all() and (ss == ss_o).all() and (ste == (idx == 4)).all(), (param, n)
                ct1, ss1, st1 = pub.encaps(m)                                # no index vector: entry 0
                ct1_o, ss1_o, _ = orc.mlkem_encaps(param, np.tile(ek[:1], (n, 1)), m)
                assert not st1.any() and (ct1 == ct1_o).all() and (ss1 == ss1_o).all(), (param, n)
                h.update(ct.tobytes() + ss.tobytes() + ste.tobytes() + ct1.tobytes())
                ct, ss, _ = orc.mlkem_encaps(param, ek[idx], m)
                ct[::3, 9] ^= 0x20
                got, st = prv.decaps(ct, idx)
                want, _ = orc.mlkem_decaps(param, dk[idx], ct)
                bad = idx == 3
                assert (st[bad] == 2).all() and not got[bad].any() and not st[~bad].any() and (got[~bad] == want[~bad]).all(), (param, n)
                ok_items = ~bad & (np.arange(n) %% 3 != 0)              # untouched ciphertexts of good keys decapsulate to the encapsulated secret
                assert (got[ok_items] == ss[ok_items]).all(), (param, n)
                got0, st0 = prv.decaps(ct)                      # no index vector: entry 0
                want0, _ = orc.mlkem_decaps(param, np.tile(dk[:1], (n, 1)), ct)
                assert not st0.any() and (got0 == want0).all(), (param, n)
                h.update(got.tobytes() + st.tobytes() + got0.tobytes())
        print("chain digest", h.hexdigest())
    """ % ROOT)
    digests = []
    for chain in ("10", "0", "12", "1"):
        env = dict(os.environ, CIRCL_HIP_KEM_CHAIN=chain, CIRCL_HIP_KEM_CHAIN_ENCAPS=chain)
        r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "chain digest" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
        digests.append(r.stdout.strip().split()[-1])
    assert len(set(digests)) == 1, digests


def test_c_examples_build_and_run():
    # examples/*.c: the C ABI from plain C, as a cgo stub would call it -- batches of distinct keys, parsed key objects (resident tables), and the
    # asynchronous form from ONE event-loop thread with the queue's eventfd in its epoll set
    out = os.path.join(ROOT, "build")
    os.makedirs(out, exist_ok=True)
    for name, args, needle in (("encaps_batch", ["3000"], " 0 mismatches"), ("resident_keys", [], "mismatches 0"), ("async_epoll", [], "mismatches 0")):
        exe = os.path.join(out, name)
        subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".c"), "-L", os.path.join(ROOT, "circl_amd"),
                               "-lcirclhip", "-Wl,-rpath," + os.path.join(ROOT, "circl_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and needle in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize("name", ["ML-DSA-44", "ML-DSA-65", "ML-DSA-87"])
def test_resident_key_verification_in_one_launch(name):
    """mldsa_verify_chain_kernel (a workgroup of K + 1 wavefronts per item, batches <= 2^CIRCL_HIP_DSA_CHAIN to a resident table):
    the Wycheproof verification set (malformed hints, out-of-range z, contexts: sign/schemes/wycheproof_test.go:116-151) key by key
    through resident tables, and a boundary batch with long messages, over-long contexts and several keys against the oracle."""
    import numpy as np
    from conftest import hx, load_golden
    from circl_amd import hostapi
    from oracle import orc
    p = {"ML-DSA-44": 44, "ML-DSA-65": 65, "ML-DSA-87": 87}[name]
    PK, SIG = hostapi.DSA_SIZES[p]
    checked = valid = 0
    for g in load_golden("mldsa_wycheproof_verify.json.gz")[name]:
        pk = hx(g["pk"])
        if len(pk) != PK:
            continue
        tests = [t for t in g["tests"] if len(hx(t["sig"])) == SIG]
        if not tests:
            continue
        t = hostapi.KeyTable("mldsa-public", p, np.frombuffer(pk, np.uint8).reshape(1, PK))
        sigs = b"".join(hx(x["sig"]) for x in tests)
        ok = t.verify(sigs, [hx(x["msg"]) for x in tests], ctxs=[hx(x["ctx"]) for x in tests])   # <= 256 per key: the one-launch route
        t.close()
        assert len(tests) <= 256
        bad = [(x["id"], int(o)) for x, o in zip(tests, ok.tolist()) if bool(o) != (x["result"] == "valid")]
        assert not bad, bad
        checked += len(tests)
        valid += int(ok.sum())
    assert checked >= 45 and 0 < valid < checked
    # the route boundary (2^10 items), several keys, ragged and long messages, contexts up to 255 bytes and beyond
    rng = np.random.default_rng(400 + p)
    nkeys = 5
    pk, sk = orc.mldsa_keygen(p, rng.integers(0, 256, (nkeys, 32), dtype=np.uint8))
    tab = hostapi.KeyTable("mldsa-public", p, pk)
    for n in (257, 1024, 1025):                                           # (2^10 items: the last batch of the one-launch route)
        idx = rng.integers(0, nkeys, n).astype(np.uint32)
        msgs = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 400, n)]
        msgs[5] = bytes(rng.integers(0, 256, 7000, dtype=np.uint8))      # 52 blocks of M'
        msgs[6] = b""
        msgs[7] = bytes(70)                                                # tr || M' ends exactly on a block boundary (64 + 2 + 70 = 136)
        ctxs = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 256, n)]
        ctxs[7] = b""
        sig = hostapi.mldsa_sign(p, sk[idx], msgs, ctxs=ctxs)
        want = np.ones(n, bool)
        sig[10, 0] ^= 1; want[10] = False                                  # c~
        sig[11, SIG - 1] ^= 0x40; want[11] = False                         # the hint counters
        sig[12, 40:44] = 0; want[12] = False                               # z
        z_of_13 = orc.mldsa_verify(p, pk[idx[13:14]], sig[13:14], msgs[13:14], ctxs=ctxs[13:14])
        assert z_of_13.all()
        ctx_long = list(ctxs)
        ctx_long[14] = bytes(256)                                          # a context of 256 bytes never verifies (dilithium.go:116-118)
        want14 = want.copy(); want14[14] = False
        ok = tab.verify(sig, msgs, ctxs=ctx_long, key_idx=idx).astype(bool)
        ref = orc.mldsa_verify(p, pk[idx], sig, msgs, ctxs=ctxs).astype(bool)
        ref[14] = False
        assert (ok == ref).all() and (ok == want14).all(), (n, np.nonzero(ok != ref)[0][:8])
    tab.close()


def test_one_launch_verification_routes_agree_with_the_older_ones():
    """The same seeded batches (valid, corrupted, contexts, a long message) through circl_hip_mldsa_verify and a resident table with the
    one-launch routes on (default), off, and stretched: identical verdicts, equal to the oracle's."""
    prog = textwrap.dedent("""
        import hashlib, sys
        import numpy as np
        sys.path.insert(0, %r)
        from circl_amd import hostapi
        from oracle import orc
        h = hashlib.sha256()
        for p in (44, 65, 87, 3):
            r3 = p == 3
            rng = np.random.default_rng(900 + p)
            nkeys = 3
            pk, sk = orc.mldsa_keygen(p, rng.integers(0, 256, (nkeys, 32), dtype=np.uint8))
            tab = hostapi.KeyTable("mldsa-public", p, pk)
            for n in (1, 2, 37, 300):
                idx = rng.integers(0, nkeys, n).astype(np.uint32)
                msgs = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 260, n)]
                if n > 2:
                    msgs[2] = bytes(rng.integers(0, 256, 3000, dtype=np.uint8))
                ctxs = None if r3 else [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 60, n)]
                sig = hostapi.mldsa_sign(p, sk[idx], msgs, ctxs=ctxs)
                sig[0::3, 5] ^= 4
                want = orc.mldsa_verify(p, pk[idx], sig, msgs, ctxs=ctxs).astype(bool)
                assert not want[0::3].any() and want[1::3].all()
                got_item = hostapi.mldsa_verify(p, pk[idx], sig, msgs, ctxs).astype(bool)
                got_tab = tab.verify(sig, msgs, ctxs=ctxs, key_idx=idx).astype(bool)
                assert (got_item == want).all() and (got_tab == want).all(), (p, n)
                sig0 = hostapi.mldsa_sign(p, np.tile(sk[:1], (n, 1)), msgs, ctxs=ctxs)      # ONE unparsed key for the batch
                sig0[1::4, 9] ^= 8
                got_one = hostapi.mldsa_verify_shared(p, pk[:1], sig0, msgs, ctxs).astype(bool)
                want_one = np.ones(n, bool)
                want_one[1::4] = False
                assert (got_one == want_one).all(), (p, n)
                h.update(got_item.tobytes() + got_tab.tobytes() + got_one.tobytes())
            tab.close()
        print("verify digest", h.hexdigest())
    """ % ROOT)
    digests = []
    for chain in ("", "0", "9"):
        env = dict(os.environ)
        if chain:
            env.update(CIRCL_HIP_DSA_CHAIN=chain, CIRCL_HIP_DSA_CHAIN_ITEM=chain)
        r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "verify digest" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
        digests.append(r.stdout.strip().split()[-1])
    assert len(set(digests)) == 1, digests


@pytest.mark.parametrize("chain", ["0", "8", "10"], ids=["four-launches", "default", "one-launch-up-to-1024"])
def test_mldsa_keygen_routes_equal_the_oracle(chain):
    """NewKeyFromSeed (sign/mldsa/mldsa65/internal/dilithium.go:181-267) as ONE launch per small batch (mldsa_keygen_chain_kernel: a workgroup of K
    wavefronts per key) against the four-launch path: the same pk / sk bytes as the oracle on both, every parameter set incl. round-3 Dilithium,
    sizes on both sides of the switch (the knob is read once: a process per setting)."""
    code = r"""
import numpy as np
from circl_amd import hostapi
from oracle import orc
rng = np.random.default_rng(7)
for param in (44, 65, 87, 2, 3, 5):
    for n in (1, 3, 40, 65, 300):
        seeds = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        pk, sk = hostapi.mldsa_keygen(param, seeds)
        pk0, sk0 = orc.mldsa_keygen(param, seeds)
        assert (pk == pk0).all() and (sk == sk0).all(), (param, n)
print("ok")
"""
    env = dict(os.environ, CIRCL_HIP_DSA_KEYGEN_CHAIN=chain, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
