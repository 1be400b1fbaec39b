"""The device-resident key-table entry points (`*_keyed_dev`, `*_table_dev`) cannot report a bad index vector -- it is device
memory, and checking it would cost a synchronisation -- so the kernels bound every read of it to the table (KeyIdx,
circl_amd/csrc/keccak_dev.h): an index >= nkeys uses the LAST entry, never memory behind the table.  The host forms keep
returning CIRCL_HIP_EPARAM (tests/test_gpu_keytable.py).  Here: every route that reads the vector (one-launch chain kernels, the
batched kernels, the signing rounds), with indices just past the end and far past it, against the same call on the bounded
vector and against the oracle on the gathered keys."""
import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu

NKEYS = 5


def _indices(rng, n):
    idx = rng.integers(0, NKEYS, n).astype(np.uint32)
    wild = np.array([NKEYS, NKEYS + 1, 0xFFFFFFFF, 1 << 20, NKEYS + 7], np.uint32)
    pos = np.arange(0, n, 3)
    idx[pos] = wild[np.arange(len(pos)) % len(wild)]
    return idx, np.minimum(idx, NKEYS - 1).astype(np.uint32)


def _cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32) if a.dtype == np.uint32 else np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("param", [512, 768, 1024])
@pytest.mark.parametrize("n", [1, 4, 700, 6000])
def test_mlkem_device_index_vector_is_bounded_to_the_table(param, n):
    import torch
    from circl_amd import device as cdev, hostapi
    rng = np.random.default_rng(31 * param + n)
    ek, dk = orc.mlkem_keygen(param, rng.integers(0, 256, (NKEYS, 64), dtype=np.uint8))
    pub, prv = hostapi.KeyTable("mlkem-public", param, ek), hostapi.KeyTable("mlkem-private", param, dk)
    idx, bounded = _indices(rng, n)
    m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    k = min(n, 400)
    ct0, ss0, _ = orc.mlkem_encaps(param, ek[bounded[:k]], m[:k])
    eng = cdev.MLKEMDevice(param, n)
    d_m, d_idx, d_ek, d_dk = _cuda(m), _cuda(idx), _cuda(ek), _cuda(dk)
    for form in ("table", "keyed"):
        if form == "table":
            ct, ss, st = eng.encaps_table(pub, d_m, d_idx)
        else:
            ct, ss, st = eng.encaps_keyed(d_ek, d_idx, d_m)
        torch.cuda.synchronize()
        ct, ss, st = ct.cpu().numpy().copy(), ss.cpu().numpy().copy(), st.cpu().numpy().copy()
        assert (st == 0).all() and (ct[:k] == ct0).all() and (ss[:k] == ss0).all(), (form, n)
        ct[::2, 9] ^= 4                                                       # every other item: implicit rejection
        d_ct = _cuda(ct)
        if form == "table":
            got, st = eng.decaps_table(prv, d_ct, d_idx)
        else:
            got, st = eng.decaps_keyed(d_dk, d_idx, d_ct)
        torch.cuda.synchronize()
        want, _ = orc.mlkem_decaps(param, dk[bounded[:k]], ct[:k])
        assert (st.cpu().numpy() == 0).all() and (got.cpu().numpy()[:k] == want).all(), (form, n)
        if n > k:                                                            # the rest: the same call on the bounded vector
            got2, _ = (eng.decaps_table(prv, d_ct, _cuda(bounded), ss=torch.empty_like(got)) if form == "table"
                       else eng.decaps_keyed(d_dk, _cuda(bounded), d_ct, ss=torch.empty_like(got)))
            torch.cuda.synchronize()
            assert bool((got2 == got).all()), (form, n)
    pub.close()
    prv.close()


@pytest.mark.parametrize("param", [44, 65, 87])
@pytest.mark.parametrize("n", [1, 6, 900])
def test_mldsa_device_index_vector_is_bounded_to_the_table(param, n):
    import ctypes as C
    import torch
    from circl_amd import _native as nat, device as cdev, hostapi
    L = nat.lib()
    rng = np.random.default_rng(57 * param + n)
    pk, sk = orc.mldsa_keygen(param, rng.integers(0, 256, (NKEYS, 32), dtype=np.uint8))
    idx, bounded = _indices(rng, n)
    msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    d_msg = _cuda(np.concatenate([msgs.reshape(-1), np.zeros(16, np.uint8)]))
    d_idx = _cuda(idx)
    eng = cdev.MLDSADevice(param, n, nkeys=NKEYS, sign=True)
    signer, verifier = hostapi.KeyTable("mldsa-private", param, sk), hostapi.KeyTable("mldsa-public", param, pk)
    # signing with entry key_idx[i] of a table of prepared private keys (deterministic: rnd = 0, empty context)
    sig = torch.empty(n * eng.SIG + 16, dtype=torch.uint8, device="cuda")[:n * eng.SIG].view(n, eng.SIG)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.circl_hip_mldsa_sign_table_keyed_dev(signer.handle, d_idx.data_ptr(), d_msg.data_ptr(), eng.off.data_ptr(), None, None, eng.rnd0.data_ptr(), 0,
                                                sig.data_ptr(), n, eng.sws.data_ptr(), eng.swsb, stream)
    nat.check(rc, "mldsa_sign_table_keyed_dev")
    torch.cuda.synchronize()
    got = sig.cpu().numpy().copy()
    k = min(n, 60)                                                            # (the oracle signs ~1 ms per signature)
    want = orc.mldsa_sign(param, sk[bounded[:k]], [bytes(r) for r in msgs[:k]], ctxs=[b""] * k)
    assert (got[:k] == want).all(), n
    assert hostapi.mldsa_verify(param, pk[bounded], got, [bytes(r) for r in msgs], ctxs=[b""] * n).all(), n
    # verification under entry key_idx[i]: the resident table and the table in the call
    bad = got.copy()
    bad[1::4, 50] ^= 1
    d_sig = _cuda(bad)
    expect = np.ones(n, bool)
    expect[1::4] = False
    ok = eng.verify_table(verifier, d_sig, d_msg, d_idx)
    torch.cuda.synchronize()
    assert (ok.cpu().numpy().astype(bool) == expect).all(), n
    ok = eng.verify_keyed(_cuda(pk), d_idx, d_sig, d_msg)
    torch.cuda.synchronize()
    assert (ok.cpu().numpy().astype(bool) == expect).all(), n
    signer.close()
    verifier.close()


@pytest.mark.parametrize("scheme_name", ["xwing", "x25519mlkem768"])
def test_hybrid_device_index_vector_is_bounded_to_the_table(scheme_name):
    import ctypes as C
    import torch
    from circl_amd import _native as nat, hostapi
    L = nat.lib()
    scheme = hostapi.XWING if scheme_name == "xwing" else hostapi.X25519MLKEM768
    sz = hostapi.HYBRID_SIZES[scheme]
    rng = np.random.default_rng(len(scheme_name))
    pk, sk = hostapi.hybrid_keygen(scheme, rng.integers(0, 256, (NKEYS, sz["seed"]), dtype=np.uint8))
    pub, prv = hostapi.KeyTable("hybrid-public", scheme, pk), hostapi.KeyTable("hybrid-private", scheme, sk)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in (2, 500):
        idx, bounded = _indices(rng, n)
        es = rng.integers(0, 256, (n, sz["eseed"]), dtype=np.uint8)
        ct0, ss0, _ = pub.hybrid_encaps(es, bounded)                                # the host form on the bounded vector (oracle-checked elsewhere)
        wsb = L.circl_hip_hybrid_workspace_size(scheme, n)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        d_es, d_idx = _cuda(es), _cuda(idx)
        ct = torch.empty((n, sz["ct"]), dtype=torch.uint8, device="cuda")
        ss = torch.empty((n, sz["ss"]), dtype=torch.uint8, device="cuda")
        st = torch.empty(n, dtype=torch.uint8, device="cuda")
        nat.check(L.circl_hip_hybrid_encaps_table_dev(pub.handle, d_idx.data_ptr(), d_es.data_ptr(), ct.data_ptr(), ss.data_ptr(), st.data_ptr(), n,
                                                      ws.data_ptr(), wsb, stream), "hybrid_encaps_table_dev")
        torch.cuda.synchronize()
        assert (ct.cpu().numpy() == ct0).all() and (ss.cpu().numpy() == ss0).all() and int(st.sum()) == 0, n
        ss2 = torch.empty_like(ss)
        nat.check(L.circl_hip_hybrid_decaps_table_dev(prv.handle, d_idx.data_ptr(), ct.data_ptr(), ss2.data_ptr(), st.data_ptr(), n, ws.data_ptr(), wsb,
                                                      stream), "hybrid_decaps_table_dev")
        torch.cuda.synchronize()
        assert (ss2.cpu().numpy() == ss0).all() and int(st.sum()) == 0, n
    pub.close()
    prv.close()
