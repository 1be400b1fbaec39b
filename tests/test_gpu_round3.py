"""Round-3 GPU tests: the multi-device host paths executed on whatever box runs the tests (logical devices), BASELINE
configs[2] in its stated shape, bench.py with 8 ranks, and the randomised soak as a bounded test.

SURVEY.md 8(e): contiguous split per device, one host thread per device, no collective.  The shape of the work is the
reference's kem/schemes/schemes_test.go:28-51 (Encapsulate / Decapsulate over a scheme), sharded."""
import json
import os
import socket
import subprocess
import sys

import pytest

from benchrec import bench_record

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "logical_worker.py")


def run_worker(args, logical, extra_env=None, timeout=1500):
    env = dict(os.environ, CIRCL_HIP_LOGICAL_DEVICES=str(logical))
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, WORKER] + [str(a) for a in args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_eight_logical_devices_equal_one_device_everywhere():
    # device = -1 over 8 logical devices (shard()'s thread-per-device branch, 8 staging pools / stream sets / mover pools)
    # == device 0 == the last logical device alone, bit for bit: ML-KEM keygen / encaps / decaps (+ shared, keyed, round 3),
    # ML-DSA keygen / sign / verify (+ shared, keyed), hybrids, X25519, primitives, XOF; ragged n incl. n < 8; oracle samples
    rep = run_worker(["parity"], 8)
    assert rep["logical_devices"] == 8 and rep["checks"] > 150


def test_three_logical_devices_uneven_split():
    rep = run_worker(["parity"], 3, {"CIRCL_HIP_HOST_SLOTS": "2", "CIRCL_HIP_HOST_CHUNK": "10"})  # tiny chunks, two slots per device
    assert rep["logical_devices"] == 3


def test_concurrent_callers_over_eight_logical_devices():
    rep = run_worker(["concurrent"], 8)
    assert rep["callers"] == 4
    # and with one staging slot per device and no mover threads (the caller thread of each shard does everything)
    run_worker(["concurrent"], 8, {"CIRCL_HIP_HOST_SLOTS": "1", "CIRCL_HIP_HOST_THREADS": "0"})


def test_config3_stated_shape_through_one_call():
    # BASELINE configs[2]: ML-KEM-768 Encaps + Decaps, 2^23 items, 8 contiguous shards of 2^20, ONE circl_hip_mlkem_* call each.
    # Needs ~45 GB of host memory for the key / ciphertext arrays; a smaller box runs the largest power of two that fits.
    import psutil
    avail = psutil.virtual_memory().available
    lg = 23
    while lg > 16 and (1 << lg) * 5600 > 0.7 * avail:  # 64 + 1184 + 2400 + 32 + 1088 + 2 * 32 + staging ~ 5.6 KB per item
        lg -= 1
    rep = run_worker(["config3", lg], 8, timeout=2400)
    assert rep["n"] == 1 << lg and rep["shard_items"] == (1 << lg) // 8
    assert rep["all_items_ss_dec_equals_ss_enc"] and rep["bit_exact_vs_oracle"] and rep["oracle_sample"] >= min(1 << 16, 1 << lg)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "config3_stated_shape.json"), "w") as f:
        json.dump(rep, f)


@pytest.mark.parametrize("mode", ["config3", "config5"])
def test_bench_py_eight_ranks(mode, tmp_path):
    # bench.py --gpus 8 as the driver launches it, with the process group on gloo and the 8 ranks sharing this box's GPU(s):
    # per-rank batches, barriers, max-over-ranks timing, whole-job aggregation, per-rank gathers, ONE JSON line from rank 0
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, CIRCL_DIST_BACKEND="gloo", CIRCL_BENCH_SHARE_GPU="1")
    xf = str(tmp_path / "extras.json")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--batch", str(1 << 12),
                        "--mode", mode, "--no-pmc", "--extras-file", xf], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    line, out = bench_record(r, xf)   # the contract line (< 6 KB) and the full record it names
    assert line["n_gpus"] == 8 and line["config"]["mode"] == mode and line["value"] > 0
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["mode"] == mode and out["value"] > 0
    cfg = out["configs"]
    if mode == "config3":
        assert cfg["decaps"]["parity"]["ranks_failing"] == 0 and out["parity"]["ranks_failing"] == 0
    else:
        assert cfg["config5"]["parity"]["ranks_failing"] == 0 and len(cfg["config5"]["per_rank_per_s"]) == 8


def test_randomised_soak_against_the_oracle():
    # tools/stress.py: random parameter sets and batch sizes through every host-buffer path (ML-KEM, ML-DSA incl. the
    # round-based signer and its speculative tail, X25519, the hybrids), each result compared with the oracle; bounded to ~50 s
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress.py"), "20260924", "50"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "stress ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---- batch signing: every round mode against the oracle ---------------------------------------------------------------------
@pytest.mark.parametrize("env_extra", [{"CIRCL_HIP_SIGN_SPEC": "1", "CIRCL_HIP_SIGN_PAIR": "1"}, {"CIRCL_HIP_SIGN_SPEC": "1"},
                                       {"CIRCL_HIP_SIGN_SPEC": "4", "CIRCL_HIP_SIGN_PAIR": "1"}, {}, {"CIRCL_HIP_SIGN_BATCHED_MIN": "100000"},
                                       {"CIRCL_HIP_SIGN_SPLIT_LOG2": "0", "CIRCL_HIP_SIGN_EPS_LOG2": "40", "CIRCL_HIP_SIGN_COOP_LOG2": "0"},
                                       {"CIRCL_HIP_SIGN_COOP_LOG2": "0"}, {"CIRCL_HIP_SIGN_COOP_LOG2": "16"},
                                       {"CIRCL_HIP_SIGN_CHAIN_LOG2": "0"}, {"CIRCL_HIP_SIGN_CHAIN_LOG2": "16"}, {"CIRCL_HIP_SIGN_FRONT": "0"}],
                         ids=["lazy-pairs", "single-attempts", "pairs-then-speculation", "default", "persistent-kernel", "lane-per-stream",
                              "lane-pairs-in-short-rounds", "cooperative-everywhere", "no-one-launch-rounds", "one-launch-rounds-everywhere", "front-end-in-three-launches"])
@pytest.mark.parametrize("param,n,shared", [(65, 1500, ""), (44, 777, ""), (87, 600, ""), (3, 640, ""), (65, 900, "shared"), (65, 1, ""), (87, 5, ""), (44, 3, "shared")])
def test_sign_round_modes(env_extra, param, n, shared):
    # sign/mldsa/mldsa65/internal/dilithium.go:340-470: the signature is the one of the FIRST attempt that passes the norm tests,
    # whatever the round structure: lazy pairs (two attempts share the matrix reads, the second one's tests run only after the
    # first was rejected), single attempts, wide speculation (lowest successful attempt wins), and the mixes a real batch goes through
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sign_worker.py"), str(param), str(n)] + ([shared] if shared else []), cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "sign worker ok" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]


# ---- ML-KEM: every batch route against the oracle ----------------------------------------------------------------------------
@pytest.mark.parametrize("env_extra", [{}, {"CIRCL_HIP_KEM_COOP": "0", "CIRCL_HIP_KEM_CHAIN_ITEM": "0"}, {"CIRCL_HIP_KEM_COOP": "15", "CIRCL_HIP_KEM_CHAIN_ITEM": "0"},
                                       {"CIRCL_HIP_KEM_COOP": "0", "CIRCL_HIP_KEM_SPLIT": "0", "CIRCL_HIP_KEM_CHAIN_ITEM": "0"},
                                       {"CIRCL_HIP_KEM_SMALL": "0", "CIRCL_HIP_KEM_SMALL_SHARED": "0", "CIRCL_HIP_KEM_SMALL_SHARED_DECAPS": "0", "CIRCL_HIP_KEM_CHAIN_ITEM": "0"},
                                       {"CIRCL_HIP_KEM_SMALL_WGS": "1", "CIRCL_HIP_KEM_CHAIN_ITEM": "0"}, {"CIRCL_HIP_KEM_CHAIN_ITEM": "0"}, {"CIRCL_HIP_KEM_CHAIN_ITEM": "12"}],
                         ids=["default", "lane-pairs", "two-per-wavefront", "lane-per-sponge", "big-batch-routes", "many-items-per-group", "no-chain",
                              "chain-everywhere"])
@pytest.mark.parametrize("param", [512, 768, 1024])
def test_kem_batch_routes(env_extra, param):
    # kem/mlkem/mlkem768/kyber.go:150-232 EncapsulateTo / DecapsulateTo: the same bytes whichever way a batch is laid over the
    # chip -- the routes switch on the batch size, so the environment forces each of them at sizes the oracle finishes quickly
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "kem_routes_worker.py"), str(param)], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "kem routes ok" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]


# ---- ML-DSA: medium batches (hash chains on lane pairs) against the oracle ------------------------------------------------------
@pytest.mark.parametrize("param", [44, 65, 87, 3])
def test_mldsa_medium_batch_against_the_oracle(param):
    # 1 024 < n <= 2^14: tr = H(pk) of key generation and of verification, and c' = H(mu || w1), run with an item per lane pair
    # (keccak_f1600_split) instead of per lane; keys, signatures and verdicts must be the reference's
    # (sign/mldsa/mldsa65/internal/dilithium.go:78-147 NewKeyFromSeed, :181-257 Verify)
    import numpy as np
    from circl_amd import hostapi
    from oracle import orc
    rng = np.random.default_rng(77 + param)
    n = 1500 + 7
    seeds = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk, sk = hostapi.mldsa_keygen(param, seeds)
    pk0, sk0 = orc.mldsa_keygen(param, seeds)
    assert (pk == pk0).all() and (sk == sk0).all()
    r3 = param in (2, 3, 5)
    msgs = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 150, n)]
    ctxs = None if r3 else [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 20, n)]
    sig = hostapi.mldsa_sign(param, sk, msgs, ctxs=ctxs)
    assert (sig[::50] == orc.mldsa_sign(param, sk0[::50], msgs[::50], ctxs=None if r3 else ctxs[::50])).all()
    bad = sig.copy()
    bad[3::7, 5] ^= 1          # c~ differs
    bad[5::11, 200] ^= 0x10    # z differs
    ok = hostapi.mldsa_verify(param, pk, bad, msgs, ctxs=ctxs)
    want = np.ones(n, bool)
    want[3::7] = False
    want[5::11] = False
    assert (ok.astype(bool) == want).all()
    assert hostapi.mldsa_verify(param, pk, sig, msgs, ctxs=ctxs).all()


@pytest.mark.parametrize("param", [768, 1024])
def test_kem_route_boundaries(param):
    # the batch sizes on both sides of every route switch (2^11, 2^14, 2^15, 2^17) give the same bytes through the latency-oriented
    # routes and through the big-batch routes (which the oracle checks at its own sizes above)
    digests = []
    for env_extra in ({}, {"CIRCL_HIP_KEM_SMALL": "0", "CIRCL_HIP_KEM_SMALL_SHARED": "0", "CIRCL_HIP_KEM_SMALL_SHARED_DECAPS": "0"},
                      {"CIRCL_HIP_KEM_COOP": "13", "CIRCL_HIP_KEM_SMALL": "17", "CIRCL_HIP_KEM_SMALL_SHARED_DECAPS": "17"}):
        env = dict(os.environ)
        env.update(env_extra)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "kem_boundary_worker.py"), str(param)], cwd=ROOT, env=env, capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0 and "kem boundary digest" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
        digests.append(r.stdout.strip().split()[-1])
    assert digests[0] == digests[1] == digests[2], digests


@pytest.mark.parametrize("param", [44, 65])
def test_mldsa_route_boundaries(param):
    # batch sizes on both sides of the ML-DSA route switches (2^10: cooperative pre-pass / lane pairs, 2^14: lane pairs / a lane per
    # item): key generation and verification of the SAME items through different routes give the same bytes and verdicts
    import numpy as np
    from circl_amd import hostapi
    rng = np.random.default_rng(500 + param)
    n = (1 << 14) + 1
    seeds = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk, sk = hostapi.mldsa_keygen(param, seeds)
    for k in (1023, 1024, 1025, 1 << 14):
        pk_k, sk_k = hostapi.mldsa_keygen(param, seeds[:k])
        assert (pk_k == pk[:k]).all() and (sk_k == sk[:k]).all(), k
    msgs = [bytes(rng.integers(0, 256, 1 + (i % 50), dtype=np.uint8)) for i in range(n)]
    sig = hostapi.mldsa_sign(param, sk, msgs)
    for k in (1023, 1025):
        assert (hostapi.mldsa_sign(param, sk[:k], msgs[:k]) == sig[:k]).all(), k
    bad = sig.copy()
    bad[3::7, 5] ^= 1
    bad[5::11, 300] ^= 0x10
    want = np.ones(n, bool)
    want[3::7] = False
    want[5::11] = False
    for k in (1023, 1024, 1025, 1 << 14, n):
        ok = hostapi.mldsa_verify(param, pk[:k], bad[:k], msgs[:k]).astype(bool)
        assert (ok == want[:k]).all(), k


# ---- long and ragged messages ---------------------------------------------------------------------------
@pytest.mark.parametrize("param", [44, 65, 87, 3])
def test_long_and_ragged_messages_against_the_oracle(param):
    # mu = SHAKE256(tr || M') (sign/mldsa/mldsa65/dilithium.go:115-132) over messages from empty to 70 KB in one batch, lengths on
    # both sides of the 136-byte block edges and of the long-message threshold (2048 bytes of M': those items are hashed two per
    # wavefront ahead of the per-lane kernels), contexts of every length class; sign and verify, per-item / one key / key table
    import numpy as np
    from circl_amd import hostapi
    from oracle import orc
    rng = np.random.default_rng(1000 + param)
    r3 = param in (2, 3, 5)
    lens = [0, 1, 70, 71, 72, 73, 135, 136, 137, 207, 208, 209, 2043, 2044, 2045, 2046, 2047, 2048, 2049, 2050, 2100, 2181, 2182, 2183, 4096, 5000, 8191, 70001]
    lens += [int(x) for x in rng.integers(0, 400, 60)] + [int(x) for x in rng.integers(1900, 2300, 20)]
    n = len(lens)
    msgs = [bytes(rng.integers(0, 256, k, dtype=np.uint8)) for k in lens]
    ctxs = None if r3 else [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.choice([0, 1, 5, 54, 55, 56, 255], n)]
    pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (n, 32), dtype=np.uint8))
    sig = hostapi.mldsa_sign(param, sk, msgs, ctxs=ctxs)
    assert (sig == orc.mldsa_sign(param, sk, msgs, ctxs=ctxs)).all()
    bad = sig.copy()
    bad[::5, 37] ^= 1
    want = np.ones(n, np.uint8)
    want[::5] = 0
    assert (hostapi.mldsa_verify(param, pk, bad, msgs, ctxs=ctxs) == want).all()
    assert hostapi.mldsa_verify(param, pk, sig, msgs, ctxs=ctxs).tolist() == orc.mldsa_verify(param, pk, sig, msgs, ctxs=ctxs).tolist()
    # one key for the batch, and a key table
    sg1 = hostapi.mldsa_sign_shared(param, sk[:1], msgs, ctxs=ctxs)
    assert (sg1 == orc.mldsa_sign(param, np.tile(sk[:1], (n, 1)), msgs, ctxs=ctxs)).all()
    assert hostapi.mldsa_verify_shared(param, pk[:1], sg1, msgs, ctxs=ctxs).all()
    idx = rng.integers(0, 7, n).astype(np.uint32)
    okk = hostapi.mldsa_verify_keyed(param, pk[:7], idx, sig, msgs, ctxs=ctxs)
    assert (okk == (idx == np.arange(n)).astype(np.uint8)).all()
    # more long messages than the pre-pass takes (4096): everything through the per-lane kernels again
    if param == 65:
        n2 = 4200
        pk2, sk2 = orc.mldsa_keygen(65, rng.integers(0, 256, (1, 32), dtype=np.uint8))
        m2 = [bytes(rng.integers(0, 256, 2100 + (i % 7), dtype=np.uint8)) for i in range(n2)]
        sg2 = hostapi.mldsa_sign_shared(65, sk2, m2)
        assert (sg2[:64] == orc.mldsa_sign(65, np.tile(sk2, (64, 1)), m2[:64])).all()
        assert hostapi.mldsa_verify_shared(65, pk2, sg2, m2).all()
