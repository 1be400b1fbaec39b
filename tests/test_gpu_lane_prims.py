"""Coefficient-level laws and sampler vectors checked ON THE GPU, against exact integer arithmetic / the oracle / the
reference's fixed vectors -- not against the host instantiation of the same source (tests/test_hostsim.py does that):
the device branch of kyber_dev.h / dilithium_dev.h uses __mul24, __umulhi, V_BITOP3, V_ALIGNBIT and V_DOT2 where the host
branch is plain C.

Reference tests mirrored here:
  pke/kyber/internal/common/poly_test.go:351-378      compress = exact rounding, decompress within the bound
  pke/kyber/internal/common/field_test.go             Montgomery / Barrett reductions
  pke/kyber/internal/common/sample_test.go:23-138     DeriveNoise2 / DeriveNoise3 (nonce 37), DeriveUniform(seed, 1, 0)
  sign/mldsa/mldsa65/internal/rounding_test.go:14-67  decompose law for every a < q; makeHint / useHint law
  sign/mldsa/mldsa65/internal/sample_test.go:12-63    PolyDeriveUniform(seed, 30000) and nonces 0..99
  sign/internal/dilithium/field_test.go               Montgomery reduction, ReduceLe2Q, power2round
"""
import numpy as np
import pytest

import lane_laws as laws
from conftest import load_golden
from lane_laws import DQ, Q

pytestmark = pytest.mark.gpu


def gpu_lane(op, a, b=None, arg=0, two=False):
    from circl_amd import hostapi
    return hostapi.lane_op(op, a, b, arg, two)


@pytest.mark.parametrize("d", [4, 5, 10, 11])
def test_compress_is_exact_rounding_for_every_representative(d):
    laws.law_compress_is_exact_rounding_for_every_representative(gpu_lane, d)


@pytest.mark.parametrize("d", [1, 4, 5, 10, 11])
def test_decompress_and_round_trip_bound(d):
    laws.law_decompress_and_round_trip_bound(gpu_lane, d)


def test_message_bit_normalize_barrett():
    laws.law_message_bit_normalize_barrett(gpu_lane)


def test_mulc_and_reduce32_at_their_bounds():
    laws.law_mulc_and_reduce32_at_their_bounds(gpu_lane)


def test_cbd2_word_and_dot2():
    laws.law_cbd2_word_and_dot2(gpu_lane)


@pytest.mark.parametrize("gamma2", [95232, 261888])
def test_decompose_law_for_every_a(gamma2):
    laws.law_decompose_law_for_every_a(gpu_lane, gamma2)


@pytest.mark.parametrize("gamma2", [95232, 261888])
def test_make_hint_use_hint_law(gamma2):
    laws.law_make_hint_use_hint_law(gpu_lane, gamma2)


def test_power2round_exceeds_normalize():
    laws.law_power2round_exceeds_normalize(gpu_lane)


def test_montgomery_products_at_their_bounds():
    laws.law_montgomery_products_at_their_bounds(gpu_lane)


def test_kyber_samplers_on_the_reference_vectors_and_the_oracle():
    from circl_amd import hostapi
    from oracle import orc
    fx = load_golden("fixed_vectors.json.gz")
    seed = np.arange(32, dtype=np.uint8)
    # sample_test.go:23-138: seed[i] = i; DeriveNoise3 / DeriveNoise2 with nonce 37, DeriveUniform(seed, 1, 0)
    assert hostapi.kyber_sample_cbd(3, seed[None])[0, 37].tolist() == fx["kyber_noise3_seed_i_nonce37"]
    assert hostapi.kyber_sample_cbd(2, seed[None])[0, 37].tolist() == fx["kyber_noise2_seed_i_nonce37"]
    assert hostapi.kyber_sample_uniform(seed[None], np.array([[1, 0]], np.uint8))[0].tolist() == fx["kyber_uniform_seed_i_x1_y0"]
    rng = np.random.default_rng(3)
    seeds = rng.integers(0, 256, (70, 32), dtype=np.uint8)                           # more than one workgroup, ragged
    for eta in (2, 3):
        got = hostapi.kyber_sample_cbd(eta, seeds)
        for i in (0, 33, 69):
            for nonce in range(64):
                assert (got[i, nonce] == orc.kyber_noise(bytes(seeds[i]), nonce, eta)).all(), (eta, i, nonce)
    # uniform: every (x, y) byte pair shape incl. 255, and seeds chosen so that 4th blocks occur (0.8 % of the streams)
    n = 3000
    seeds = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    xy = rng.integers(0, 256, (n, 2), dtype=np.uint8)
    xy[:4] = [[0, 0], [255, 255], [0, 255], [3, 2]]
    got = hostapi.kyber_sample_uniform(seeds, xy)
    assert ((got >= 0) & (got < Q)).all()
    for i in range(0, n, 7):
        assert (got[i] == orc.kyber_uniform(bytes(seeds[i]), int(xy[i, 0]), int(xy[i, 1]))).all(), i


def test_mldsa_uniform_sampler_on_the_reference_vectors_and_the_oracle():
    from circl_amd import hostapi
    from oracle import orc
    fx = load_golden("fixed_vectors.json.gz")
    seed = np.arange(32, dtype=np.uint8)
    # sample_test.go:12-52 TestVectorDeriveUniform: seed[i] = i, nonce 30000
    assert hostapi.mldsa_sample_uniform(seed[None], [30000])[0].tolist() == fx["dilithium_uniform_seed_i_nonce30000"]
    # sample_test.go:54-63 TestDeriveUniform: seed = LE64(i), nonce i, i < 100: every coefficient < q (and here: == the oracle)
    seeds = np.zeros((100, 32), np.uint8)
    seeds[:, 0] = np.arange(100)
    got = hostapi.mldsa_sample_uniform(seeds, np.arange(100))
    assert (got < DQ).all()
    for i in range(100):
        assert (got[i] == orc.dilithium_uniform(bytes(seeds[i]), i)).all(), i
    rng = np.random.default_rng(6)
    n = 5000
    seeds = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    nonces = rng.integers(0, 1 << 16, n).astype(np.uint16)
    nonces[:3] = [0, 65535, 0x0807]
    got = hostapi.mldsa_sample_uniform(seeds, nonces)
    assert (got < DQ).all()
    for i in range(0, n, 11):
        assert (got[i] == orc.dilithium_uniform(bytes(seeds[i]), int(nonces[i]))).all(), i
