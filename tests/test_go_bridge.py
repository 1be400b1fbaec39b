"""The cgo bridge under go/ has never met a Go compiler (no toolchain on any box: profiles/r04_box_probe.txt).  tools/gocheck.py does
the part of a compiler's front end that can be done here -- C symbols, argument counts and types of every C call, arity of
package-level calls, unused imports / variables, undefined names, and every CIRCL identifier, interface method, argument count and
result count against the names and shapes the reference exports (tests/golden/go_api_symbols.json, regenerated from the reference
by `tools/gocheck.py --update-fixture`).  The bridge's own Go tests (go/*/hipbatch/hipbatch_test.go: batch results against CIRCL's
scheme objects, the parity test a maintainer runs) are checked the same way.
The mutation cases prove the checks fire.
"""
import os
import re
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gocheck  # noqa: E402


def test_bridge_is_clean_and_covered():
    findings, stats = gocheck.run()
    assert findings == []
    assert stats["files"] == 21 and stats["c_calls"] >= 70   # 15 bridge files (two of them the round-6 reactors) + the six parity tests a maintainer runs
    # every argument of every C call was typed from the Go source and compared with the prototype
    assert stats["c_args_typed"] == stats["c_args"] >= 270, stats["c_args_untyped"]
    assert stats["api_idents"] >= 140 and stats["api_methods"] >= 140 and stats["api_shapes"] >= 170


def test_every_exported_entry_point_family_is_bound():
    """the bridge binds the host-buffer entry points a Go caller needs (the _dev forms are for device-resident callers)"""
    src = "".join(open(p).read() for p in gocheck.glob.glob(os.path.join(ROOT, "go", "**", "*.go"), recursive=True))
    used = set(re.findall(r"C\.(circl_hip_\w+)\(", src))
    protos, _types, _macros = gocheck.parse_header(os.path.join(ROOT, "include", "circl_hip.h"))
    for name in ("circl_hip_mlkem_encaps", "circl_hip_mlkem_decaps", "circl_hip_mlkem_keygen", "circl_hip_mlkem_encaps_shared",
                 "circl_hip_mldsa_verify", "circl_hip_mldsa_sign", "circl_hip_mldsa_keygen", "circl_hip_mlkem_keytable_new",
                 "circl_hip_mldsa_privkeys_new", "circl_hip_hybrid_encaps", "circl_hip_hybrid_decaps", "circl_hip_xof", "circl_hip_keccak_f1600",
                 "circl_hip_mlkem_public_from_private", "circl_hip_mldsa_public_from_private"):
        assert name in protos and name in used, name
    assert used <= set(protos)


MUTATIONS = [
    # (file, old, new, expected finding fragment)
    ("kem/mlkem/hipbatch/hipbatch.go", "C.circl_hip_mlkem_encaps(p, ptr(eks), ptr(seeds), ptr(cts), ptr(sss), ptr(st), C.size_t(n), C.int(device))",
     "C.circl_hip_mlkem_encaps(p, ptr(eks), ptr(seeds), ptr(cts), ptr(sss), C.size_t(n), C.int(device))", "takes 8 argument(s), called with 7"),
    ("kem/mlkem/hipbatch/hipbatch.go", "C.circl_hip_mlkem_encaps(p, ptr(eks), ptr(seeds), ptr(cts), ptr(sss), ptr(st), C.size_t(n), C.int(device))",
     "C.circl_hip_mlkem_encaps(p, ptr(eks), ptr(seeds), ptr(cts), ptr(sss), ptr(st), C.int(n), C.int(device))", "int passed for size_t"),
    ("kem/mlkem/hipbatch/hipbatch.go", "C.circl_hip_mlkem_encaps(p,", "C.circl_hip_mlkem_encapsulate(p,", "is not declared by include/circl_hip.h"),
    ("kem/mlkem/hipbatch/hipbatch.go", "return kem.ErrPubKey //", "return kem.ErrPublicKey //", "kem.ErrPublicKey does not exist in the reference"),
    ("kem/mlkem/hipbatch/hipbatch.go", "if len(eks)%s.PublicKeySize() != 0 {", "if len(eks)%s.PubKeySize() != 0 {", "kem.Scheme has no method PubKeySize"),
    ("kem/mlkem/hipbatch/hipbatch.go", "\tsss = make([]byte, n*s.SharedKeySize())\n\tst := make([]byte, n)\n\tif ok3 {",
     "\tsss = make([]byte, n*s.SharedKeySize())\n\tst := make([]byte, n)\n\tunused := 3\n\tif ok3 {", "unused declared and not used"),
    ("kem/mlkem/hipbatch/hipbatch.go", '\t"errors"\n', '\t"errors"\n\t"strings"\n', 'import "strings" is not used'),
    ("sign/mldsa/hipbatch/keytable.go", "C.size_t(r.n), C.int(device), &r.t)\n\t} else {", "C.size_t(r.n), C.int(device), r.t)\n\t} else {",
     "circl_hip_keytable* passed for circl_hip_keytable**"),
    ("kem/mlkem/hipbatch/scheme.go", "ct, ss, errs, err := EncapsulateBatch(s.Scheme, eks, seeds, device)", "ct, ss, errs, err := EncapsulateBatch(s.Scheme, eks, device)",
     "EncapsulateBatch takes 4 argument(s), called with 3"),
    ("kem/mlkem/hipbatch/scheme.go", "ct, ss, errs, err := EncapsulateBatch(s.Scheme, eks, seeds, device)", "ct, ss, err := EncapsulateBatch(s.Scheme, eks, seeds, device)",
     "assignment mismatch: 3 variables but EncapsulateBatch returns 4 values"),
    ("kem/mlkem/hipbatch/scheme.go", "return rows(ct, s.CiphertextSize()), rows(ss, s.SharedKeySize()), errs, nil", "return rowz(ct, s.CiphertextSize()), rows(ss, s.SharedKeySize()), errs, nil",
     "undefined: rowz"),
    ("kem/mlkem/hipbatch/hipbatch.go", '\t"fmt"\n', "", "undefined: fmt"),
    ("kem/mlkem/hipbatch/scheme.go", "return rows(ct, s.CiphertextSize()), rows(ss, s.SharedKeySize()), errs, nil", "return rows(ct, s.CiphertextSize()), errs, nil",
     "wrong number of return values: have 3, want 4"),
    ("kem/mlkem/hipbatch/hipbatch_test.go", "pks, sks, err := s.DeriveKeyPairBatch(seeds, AllDevices)", "pks, sks, err := s.DeriveKeyPairBatch(seeds, AllDevs)", "undefined: AllDevs"),
    ("sign/mldsa/hipbatch/hipbatch.go", "\tfor k, i := range idx {\n\t\tres[i] = okb[k] == 1\n\t}\n\treturn res, nil", "\tfor k, i := range idx {\n\t\tres[i] = okb[k] == 1\n\t}\n\treturn rez, nil", "undefined: rez"),
    ("sign/mldsa/hipbatch/hipbatch_test.go", "want := s.Sign(keys[i], msgs[i], &sign.SignatureOpts{Context: ctxs[i]})", "want := s.SignMessage(keys[i], msgs[i], &sign.SignatureOpts{Context: ctxs[i]})",
     "sign.Scheme has no method SignMessage"),
    ("kem/mlkem/hipbatch/hipbatch_test.go", "ct, ss, err := s.EncapsulateDeterministically(pk, row(eseeds, s.EncapsulationSeedSize(), i))",
     "ct, ss, err := s.EncapsulateDeterministically(pk)", "kem.Scheme.EncapsulateDeterministically takes 2 argument(s), called with 1"),
    ("kem/mlkem/hipbatch/hipbatch_test.go", "pk, sk := s.DeriveKeyPair(row(kseeds, s.SeedSize(), i))", "pk, sk, err := s.DeriveKeyPair(row(kseeds, s.SeedSize(), i))",
     "assignment mismatch: 3 variables but"),
    ("sign/mldsa/hipbatch/hipbatch_test.go", "s := schemes.ByName(name)", "s := schemes.ByName(name, 1)", "schemes.ByName takes 1 argument(s), called with 2"),
    ("kem/hybrid/hipbatch/hipbatch_test.go", 'circl "github.com/cloudflare/circl/kem/schemes"', '"github.com/cloudflare/circl/kem/schemes"',
     "schemes redeclared in this block"),
    ("kem/mlkem/hipbatch/hipbatch_test.go", 't.Fatalf("key pair %d differs", i)', 't.Fatalf("key pair %d of %d differs", i)', "Fatalf format has 2 verb(s), 1 argument(s)"),
    ("xof/hipbatch/hipbatch.go", "/*\n#cgo", "/*\n#include <no_such_header.h>\n#cgo", "cgo preamble does not compile"),
    ("dh/x25519/hipbatch/hipbatch.go", "package hipbatch", "package hipbatch\n\nfunc broken( {", "unclosed {"),
]


@pytest.mark.parametrize("case", range(len(MUTATIONS)))
def test_mutations_are_caught(tmp_path, case):
    rel, old, new, expect = MUTATIONS[case]
    shutil.copytree(os.path.join(ROOT, "go"), tmp_path / "go")
    os.symlink(os.path.join(ROOT, "include"), tmp_path / "include")
    p = tmp_path / "go" / rel
    src = p.read_text()
    assert src.count(old) >= 1, "the mutation's anchor text is gone from %s: update the test" % rel
    p.write_text(src.replace(old, new, 1))
    findings, _stats = gocheck.run(go_root=str(tmp_path / "go"))
    assert any(expect in f for f in findings), (expect, findings)


def test_fixture_matches_the_reference_when_it_is_here():
    """regenerate the names from the reference tree (this container only) and compare with the committed fixture"""
    ref = os.environ.get("CIRCL_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "kem")):
        pytest.skip("reference tree not on this box (the committed fixture stands)")
    import json
    committed = json.load(open(gocheck.FIXTURE))["packages"]
    saved = gocheck.FIXTURE
    try:
        gocheck.FIXTURE = os.path.join(os.environ.get("TMPDIR", "/tmp"), "go_api_symbols.regen.json")
        gocheck.update_fixture(ref)
        assert json.load(open(gocheck.FIXTURE))["packages"] == committed
    finally:
        gocheck.FIXTURE = saved
