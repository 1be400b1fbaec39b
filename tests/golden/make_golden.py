#!/usr/bin/env python3
"""Regenerate tests/golden/*.json.gz from the reference checkout (/root/reference).

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_golden.py

What is extracted (data only -- NIST ACVP / Wycheproof / Keccak-team KAT vectors that the
reference's own tests read):
  mlkem_acvp.json.gz   kem/mlkem/testdata/ML-KEM-{keyGen,encapDecap}-FIPS203 (test logic:
                       kem/mlkem/acvp_test.go:35-164).  keyGen outputs are stored as SHA-256
                       digests of ek and dk (the inputs d,z are what matter).
  mldsa_acvp.json.gz   sign/mldsa/testdata/ML-DSA-{keyGen,sigGen,sigVer}-FIPS204 (test logic:
                       sign/mldsa/mldsa65/acvp_test.go:39-161).  keyGen outputs and sigGen
                       signatures as SHA-256 digests; sigVer complete.
  mldsa_wycheproof_verify.json.gz  sign/schemes/testdata/wycheproof/mldsa_{44,65,87}_verify_test
                       (test logic: sign/schemes/wycheproof_test.go:116-151).
  mldsa_wycheproof_sign.json.gz  sign/schemes/testdata/wycheproof/mldsa_*_sign_{seed,noseed}_test (test logic:
                       sign/schemes/wycheproof_test.go:57-115; deterministic signatures through the public Sign
                       with contexts).  Signatures are stored as SHA-256 digests.
  sha3_kats.json.gz    internal/sha3/testdata/keccakKats.json.deflate: every 8th ShortMsgKAT of
                       SHA3-256, SHA3-512, SHAKE128, SHAKE256 (byte-aligned lengths only).
  fixed_vectors.json.gz  the literal expected arrays of the reference's fixed-vector unit tests:
                       pke/kyber/internal/common/sample_test.go:23-138 (CBD3, CBD2, uniform; seed[i]=i),
                       sign/mldsa/mldsa65/internal/sample_test.go:12-63 (uniform, nonce 30000),
                       simd/keccakf1600/f1600x_test.go:9-19 (Keccak-f[1600] of the zero state).
  x25519.json.gz       dh/x25519/testdata/{rfc7748_kat_test, rfc7748_times_test, wycheproof_kat}.json.gz, the vectors of
                       dh/x25519/key_test.go:24-46 (TestRFC7748Kat), :53-84 (TestRFC7748Times), :114-150 (TestWycheproof).
All binary fields are hex strings.
"""
import re
import gzip
import hashlib
import io
import json
import os
import zlib

REF = os.environ.get("CIRCL_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def load_gz(p):
    with gzip.open(os.path.join(REF, p)) as f:
        return json.load(f)


def dump(name, obj):
    raw = json.dumps(obj, separators=(",", ":"), sort_keys=True).encode()
    buf = io.BytesIO()
    with gzip.GzipFile(fileobj=buf, mode="wb", mtime=0, compresslevel=9) as f:
        f.write(raw)
    with open(os.path.join(OUT, name), "wb") as f:
        f.write(buf.getvalue())
    print(name, len(buf.getvalue()), "bytes")


def sha(hexstr):
    return hashlib.sha256(bytes.fromhex(hexstr)).hexdigest()


def acvp(dirname):
    pr = load_gz(dirname + "/prompt.json.gz")
    ex = load_gz(dirname + "/expectedResults.json.gz")
    res = {}
    for g in ex["testGroups"]:
        for t in g["tests"]:
            res[t["tcId"]] = t
    return pr["testGroups"], res


def mlkem():
    out = {}
    groups, res = acvp("kem/mlkem/testdata/ML-KEM-keyGen-FIPS203")
    for g in groups:
        ps = out.setdefault(g["parameterSet"], {"keygen": [], "encap": [], "decap": []})
        for t in g["tests"]:
            r = res[t["tcId"]]
            ps["keygen"].append({"d": t["d"], "z": t["z"], "ek_sha256": sha(r["ek"]), "dk_sha256": sha(r["dk"])})
    groups, res = acvp("kem/mlkem/testdata/ML-KEM-encapDecap-FIPS203")
    for g in groups:
        ps = out[g["parameterSet"]]
        if g["testType"] == "AFT":
            for t in g["tests"]:
                r = res[t["tcId"]]
                ps["encap"].append({"ek": t["ek"], "m": t["m"], "c": r["c"], "k": r["k"]})
        else:
            ps["decap"].append({"dk": g["dk"], "cases": [{"c": t["c"], "k": res[t["tcId"]]["k"]} for t in g["tests"]]})
    dump("mlkem_acvp.json.gz", out)


def mldsa():
    out = {}
    groups, res = acvp("sign/mldsa/testdata/ML-DSA-keyGen-FIPS204")
    for g in groups:
        ps = out.setdefault(g["parameterSet"], {"keygen": [], "siggen": [], "sigver": []})
        for t in g["tests"]:
            r = res[t["tcId"]]
            ps["keygen"].append({"seed": t["seed"], "pk_sha256": sha(r["pk"]), "sk_sha256": sha(r["sk"])})
    groups, res = acvp("sign/mldsa/testdata/ML-DSA-sigGen-FIPS204")
    for g in groups:
        ps = out[g["parameterSet"]]
        # all 10 cases of both groups per parameter set (deterministic and hedged), as sign/mldsa/mldsa65/acvp_test.go:81-120
        # runs them; all use Sign_internal
        for t in g["tests"]:
            r = res[t["tcId"]]
            ps["siggen"].append({"sk": t["sk"], "message": t["message"],
                                 "rnd": t.get("rnd", "00" * 32) if not g["deterministic"] else "00" * 32,
                                 "sig_sha256": sha(r["signature"])})
    groups, res = acvp("sign/mldsa/testdata/ML-DSA-sigVer-FIPS204")
    for g in groups:
        ps = out[g["parameterSet"]]
        ps["sigver"].append({"pk": g["pk"], "cases": [
            {"message": t["message"], "signature": t["signature"], "passed": res[t["tcId"]]["testPassed"]}
            for t in g["tests"]]})
    dump("mldsa_acvp.json.gz", out)


def wycheproof():
    out = {}
    for mode in (44, 65, 87):
        w = load_gz(f"sign/schemes/testdata/wycheproof/mldsa_{mode}_verify_test.json.gz")
        gs = []
        for g in w["testGroups"]:
            gs.append({"pk": g["publicKey"], "tests": [
                {"id": t["tcId"], "comment": t["comment"], "msg": t["msg"], "ctx": t.get("ctx", ""),
                 "sig": t["sig"], "result": t["result"]} for t in g["tests"]]})
        out[f"ML-DSA-{mode}"] = gs
    dump("mldsa_wycheproof_verify.json.gz", out)


def wycheproof_sign():
    out = {}
    names = {44: ("mldsa_44_sign_seed_test", "mldsa_44_sign_noseed_test"),
             65: ("mldsa_65_seed_sign_test", "mldsa_65_noseed_sign_test"),
             87: ("mldsa_87_sign_seed_test", "mldsa_87_sign_noseed_test")}
    for mode, files in names.items():
        gs = []
        for fn in files:
            w = load_gz(f"sign/schemes/testdata/wycheproof/{fn}.json.gz")
            for g in w["testGroups"]:
                gs.append({"seed": g.get("privateSeed"), "sk": g.get("privateKey"), "tests": [
                    {"id": t["tcId"], "comment": t["comment"], "msg": t["msg"], "ctx": t.get("ctx", ""),
                     "sig_sha256": sha(t["sig"]) if t["sig"] else "", "result": t["result"]} for t in g["tests"]]})
        out[f"ML-DSA-{mode}"] = gs
    dump("mldsa_wycheproof_sign.json.gz", out)


def sha3():
    with open(os.path.join(REF, "internal/sha3/testdata/keccakKats.json.deflate"), "rb") as f:
        kats = json.loads(zlib.decompress(f.read(), -15))["kats"]
    out = {}
    for alg in ("SHA3-256", "SHA3-512", "SHAKE128", "SHAKE256"):
        sel = [k for k in kats[alg] if k["length"] % 8 == 0][::8]
        out[alg] = [{"msg": k["message"][: k["length"] // 4], "digest": k["digest"]} for k in sel]
    dump("sha3_kats.json.gz", out)


def _array_after(path, marker):
    src = open(os.path.join(REF, path)).read()
    i = src.index(marker)
    body = src[src.index("{", i) + 1: src.index("}", i)]
    return [int(x, 0) for x in re.findall(r"-?(?:0x[0-9A-Fa-f]+|\d+)", body)]


def fixed():
    ks = "pke/kyber/internal/common/sample_test.go"
    out = {
        "kyber_noise3_seed_i_nonce37": _array_after(ks, "func TestPolyDeriveNoise3Ref"),
        "kyber_noise2_seed_i_nonce37": _array_after(ks, "func TestPolyDeriveNoise2Ref"),
        "kyber_uniform_seed_i_x1_y0": _array_after(ks, "func TestPolyDeriveUniformRef"),
        "dilithium_uniform_seed_i_nonce30000": _array_after(
            "sign/mldsa/mldsa65/internal/sample_test.go", "p2 = common.Poly"),
        "keccak_f1600_of_zero": _array_after("simd/keccakf1600/f1600x_test.go", "var permutationOfZeroes"),
    }
    assert all(len(out[k]) == 256 for k in out if k != "keccak_f1600_of_zero"), {k: len(v) for k, v in out.items()}
    assert len(out["keccak_f1600_of_zero"]) == 25
    dump("fixed_vectors.json.gz", out)


def x25519():
    d = "dh/x25519/testdata/"
    dump("x25519.json.gz", {
        "rfc7748_kat": load_gz(d + "rfc7748_kat_test.json.gz"),
        "rfc7748_times": load_gz(d + "rfc7748_times_test.json.gz"),
        "wycheproof": [{k: v[k] for k in ("tcId", "public", "private", "shared", "result")} for v in load_gz(d + "wycheproof_kat.json.gz")],
    })


if __name__ == "__main__":
    fixed()
    x25519()
    mlkem()
    mldsa()
    wycheproof()
    wycheproof_sign()
    sha3()
