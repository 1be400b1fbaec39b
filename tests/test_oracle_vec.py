"""oracle/vec (the batch-vectorised CPU encapsulation that bench.py's cpu_baseline times next to the scalar oracle) is pinned
against the oracle: same ciphertexts, shared secrets and status bytes for both instruction sets, ragged batch sizes, several
threads, keys that kem.ErrPubKey rejects (kem/mlkem/mlkem768/kyber.go:247-263), and the reference's own ACVP encapsulation
vectors; its Keccak-f[1600] x4 / x8 against the scalar permutation (simd/keccakf1600/f1600x_test.go:13-60's shape)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import orc

ISAS = [i for i in (1, 2) if orc.vec_isa(i) == i]
pytestmark = pytest.mark.skipif(not ISAS, reason="this CPU has no AVX2")


@pytest.mark.parametrize("isa", ISAS)
def test_permutation_on_four_or_eight_states(isa):
    n = 4 if isa == 1 else 8
    st = np.random.default_rng(isa).integers(0, 1 << 63, (n, 25), dtype=np.uint64)
    st[0] = 0  # keccak_f1600_of_zero is one of the golden vectors
    got = orc.vec_keccak_f1600(st, isa)
    for l in range(n):
        assert (got[l] == orc.keccak_f1600(st[l].copy(), 24)).all()
    assert (got[0] == np.array(load_golden("fixed_vectors.json.gz")["keccak_f1600_of_zero"], dtype=np.uint64)).all()


@pytest.mark.parametrize("param", [768, 1024])
@pytest.mark.parametrize("isa", ISAS)
def test_batches_equal_the_oracle(param, isa):
    rng = np.random.default_rng(100 * isa + param)
    n = 2 * 32 * 3 + 19  # whole groups for three threads and a ragged tail
    ek, _ = orc.mlkem_keygen(param, rng.integers(0, 256, (n, 64), dtype=np.uint8))
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ek = ek.copy()
    ek[3, 0:2] = 0xff         # a coefficient of 4095: ErrPubKey
    ek[n - 1, 382:384] = 0xff  # ... in the ragged tail
    ct0, ss0, st0 = orc.mlkem_encaps(param, ek, m)
    assert st0.sum() == 2
    for threads in (1, 3):
        ct, ss, st = orc.mlkem_encaps_vec(param, ek, m, threads=threads, isa=isa)
        assert (st == st0).all() and (ct == ct0).all() and (ss == ss0).all()
    for k in (1, 15, 33):  # batches smaller than a vector, one thread per group
        ct, ss, st = orc.mlkem_encaps_vec(param, ek[:k], m[:k], threads=4, isa=isa)
        assert (st == st0[:k]).all() and (ct == ct0[:k]).all() and (ss == ss0[:k]).all()


@pytest.mark.parametrize("param", [768, 1024])
@pytest.mark.parametrize("isa", ISAS)
def test_one_key_for_the_batch(param, isa):
    # the parsed-key shape of the reference's BenchmarkEncapsulate (kem/schemes/schemes_test.go:28-38): the key's stages once per thread
    rng = np.random.default_rng(7 * isa + param)
    n = 5 * 32 + 3
    ek, _ = orc.mlkem_keygen(param, rng.integers(0, 256, (2, 64), dtype=np.uint8))
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ct0, ss0 = orc.mlkem_encaps_shared(param, ek[:1], m)
    for threads in (1, 2):
        ct, ss, st = orc.mlkem_encaps_vec(param, ek[:1], m, threads=threads, isa=isa, shared=True)
        assert not st.any() and (ct == ct0).all() and (ss == ss0).all()
    bad = ek[1:2].copy()
    bad[0, 0:2] = 0xff
    ct, ss, st = orc.mlkem_encaps_vec(param, bad, m, threads=2, isa=isa, shared=True)
    assert (st == 1).all() and not ct.any() and not ss.any()


@pytest.mark.parametrize("name", ["ML-KEM-768", "ML-KEM-1024"])
@pytest.mark.parametrize("isa", ISAS)
def test_acvp_encapsulation_vectors(name, isa):
    # the reference's ACVP encapsulation vectors (kem/mlkem/acvp_test.go), the cases the scalar oracle is pinned by
    cases = load_golden("mlkem_acvp.json.gz")[name]["encap"]
    ek = np.stack([np.frombuffer(bytes.fromhex(t["ek"]), np.uint8) for t in cases])
    m = np.stack([np.frombuffer(bytes.fromhex(t["m"]), np.uint8) for t in cases])
    ct, ss, st = orc.mlkem_encaps_vec(int(name.split("-")[-1]), ek, m, threads=2, isa=isa)
    assert not st.any()
    for i, t in enumerate(cases):
        assert ct[i].tobytes() == bytes.fromhex(t["c"]) and ss[i].tobytes() == bytes.fromhex(t["k"])


def test_bench_cpu_baseline_leg_reports_both_ports():
    # bench.py's cpu_baseline on a stand-in workload (no GPU): the vectorised figure in `value`, the scalar oracle beside it, the bytes compared
    import importlib.util
    import os
    import types

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rng = np.random.default_rng(11)
    n = 1 << 11
    ek, _ = orc.mlkem_keygen(768, rng.integers(0, 256, (n, 64), dtype=np.uint8))
    work = types.SimpleNamespace(param=768, ek=torch.from_numpy(ek), m=torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)))
    r = bench.cpu_baseline(work, budget_s=0.5)
    assert r["kind"] == "port" and r["unit"] == "encaps/s" and r["cores"] >= 1 and r["value"] > 0 and "sample" in r
    # every figure ONCE (the vectorised block used to be printed twice): `value` / `per_thread` / `shared_key` are the vector
    # port's, checked against the scalar oracle; the scalar oracle's own figures sit under `scalar_oracle`
    assert "vectorized" not in r and "vectorized_failed" not in r
    assert r["equals_scalar_oracle_on_first_items"][1] and r["shared_key"]["equals_scalar_oracle_on_first_items"][1]
    assert r["isa"].startswith("AVX") and r["per_thread"] * r["cores"] == pytest.approx(r["value"])
    assert r["scalar_oracle"]["value"] > 0 and r["scalar_oracle"]["shared_key"]["value"] > 0
    assert r["value"] > r["scalar_oracle"]["value"]  # a vectorised port slower than the scalar restatement would be a bug


@pytest.mark.parametrize("isa", ISAS)
def test_two_independent_cpu_implementations_agree_on_a_large_batch(isa):
    # 2^14 distinct keys: ~1 % of the 9 * 2^14 SHAKE128 streams need a fourth block (the straggler loop of the vectorised sampler), every
    # compression / packing residue occurs; the scalar oracle and oracle/vec share no code below the Keccak round constants
    rng = np.random.default_rng(2024 + isa)
    n = 1 << 14
    ek, _ = orc.mlkem_keygen(768, rng.integers(0, 256, (n, 64), dtype=np.uint8))
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ct0, ss0, st0 = orc.mlkem_encaps(768, ek, m)
    ct, ss, st = orc.mlkem_encaps_vec(768, ek, m, isa=isa)
    assert not st0.any() and not st.any()
    assert (ct == ct0).all() and (ss == ss0).all()
