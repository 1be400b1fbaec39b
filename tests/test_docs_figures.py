"""DESIGN.md / README.md quote the committed bench record, not a memory of some earlier run (DESIGN said 1.43e8 where the
file it cited held 1.409e8).  Every figure of DESIGN section 5's "Current figures" sentence and of README's "Numbers" bullet that names
`profiles/r06_bench.json` is parsed here and compared with that file."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _num(text):
    """'1.428×10^8' -> 1.428e8, '0.838' -> 0.838"""
    m = re.fullmatch(r"([0-9.]+)(?:×10\^([0-9]+))?", text)
    assert m, text
    return float(m.group(1)) * (10 ** int(m.group(2)) if m.group(2) else 1)


def _close(doc, actual, what, tol=0.006):
    assert abs(doc - actual) <= tol * abs(actual), "%s: the document says %g, the bench record %g" % (what, doc, actual)


def _record():
    with open(os.path.join(ROOT, "profiles", "r06_bench.json")) as f:
        return json.load(f)


def test_design_section_5_quotes_the_committed_bench_record():
    d = _record()
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"Current figures \(one MI355X, round 6's final library,\s*`profiles/r06_bench\.json`\):(.*?)the GPU kernel's own measure", text, re.S)
    assert m, "DESIGN.md section 5 no longer has its 'Current figures' sentence"
    s = m.group(1)
    N = r"([0-9.]+(?:×10\^[0-9]+)?)"

    def grab(pattern):
        g = re.search(pattern, s, re.S)
        assert g, pattern
        return [_num(x) for x in g.groups()]
    c = d["configs"]
    _close(grab(r"headline \*\*" + N + r"/s\*\*")[0], d["value"], "headline")
    ms, hash_ms, enc_ms = grab(r"\(" + N + r" ms per 2\^20: hash " + N + r" \+ encrypt " + N)
    _close(ms, d["ms_per_step"], "ms per step")
    _close(enc_ms, d["roofline"]["avg_launch_ms"], "encrypt kernel ms")
    _close(hash_ms + enc_ms, d["ms_per_step"], "hash + encrypt = a step", tol=0.01)
    _close(grab(r"decaps " + N)[0], c["decaps"]["value"], "decaps")
    _close(grab(r"pairs " + N)[0], c["config3"]["value"], "config 3")
    _close(grab(r"ML-DSA-65 verify " + N)[0], c["config4"]["value"], "config 4")
    _close(grab(r"config 5 " + N)[0], c["config5"]["value"], "config 5")
    pageable, pinned = grab(r"host ABI " + N + r" pageable /\s*" + N + r" page-locked")
    _close(pageable, d["value_host_abi"]["value"], "host ABI pageable")
    _close(pinned, d["value_host_abi"]["pinned"], "host ABI page-locked")
    f_head, f65, f87, f1024 = grab(r"`frac_of_mix_ceiling` " + N + r" \(headline\), " + N + r" \(`mldsa_verify_kernel<65>`\), " + N + r" \(`<87>`\), " + N)
    _close(f_head, d["roofline"]["valu"]["frac_of_mix_ceiling"], "mix ceiling, headline")
    _close(f65, c["config4"]["roofline"]["valu_frac_of_mix_ceiling"], "mix ceiling, verify<65>")
    _close(f87, c["config5"]["roofline_mldsa87"]["valu_frac_of_mix_ceiling"], "mix ceiling, verify<87>")
    _close(f1024, c["config5"]["roofline_mlkem1024"]["valu_frac_of_mix_ceiling"], "mix ceiling, encrypt<4>")
    cpu, shared, scalar = grab(N + r"/s vectorised \(one parsed key.*?" + N + r"/s\), " + N + r"/s the scalar oracle")
    _close(cpu, d["cpu_baseline"]["value"], "cpu_baseline.value")
    _close(shared, d["cpu_baseline"]["shared_key"]["value"], "cpu_baseline.shared_key")
    _close(scalar, d["cpu_baseline"]["scalar_oracle"]["value"], "cpu_baseline.scalar_oracle")


def test_readme_numbers_quote_the_committed_bench_record():
    d = _record()
    text = open(os.path.join(ROOT, "README.md")).read()
    m = re.search(r"\*\*Numbers \(round 6, one MI355X; `profiles/r06_bench\.json`\)\*\*:(.*?)\n\* \*\*", text, re.S)
    assert m, "README.md no longer has its round-6 'Numbers' bullet"
    s = m.group(1)
    N = r"([0-9.]+(?:×10\^[0-9]+)?)"
    c = d["configs"]
    for pattern, actual, what in (
            (N + r" ML-KEM-768 encapsulations/s at batch 2\^20", d["value"], "headline"),
            (N + r" of a\s+VALU ceiling", d["roofline"]["valu"]["frac_of_mix_ceiling"], "mix ceiling"),
            (N + r" decapsulations/s", c["decaps"]["value"], "decaps"),
            (N + r" ML-DSA-65 verifications/s", c["config4"]["value"], "config 4"),
            (N + r" encapsulations/s \(PCIe-bound", d["value_host_abi"]["value"], "host ABI"),
            (r"EPYC 9575F: " + N + r"/s", d["cpu_baseline"]["value"], "cpu baseline")):
        g = re.search(pattern, s, re.S)
        assert g, pattern
        _close(_num(g.group(1)), actual, what)


def test_sync_docs_has_nothing_to_do():
    """tools/sync_docs.py --check: the documents already carry the committed record's figures (and its patterns still find their sentences)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sync_docs.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
