#!/usr/bin/env python3
"""tools/sync_docs.py -- rewrite the figures DESIGN.md section 5 and README.md quote from profiles/r06_bench.json, so that a new evidence run is
one command away from consistent documents (tests/test_docs_figures.py is the check; this is the edit).  Only the sentences that name
`profiles/r06_bench.json` are touched; prose, ranges over several boxes and figures of other files stay as written.

    python tools/sync_docs.py            # edit in place
    python tools/sync_docs.py --check    # exit 1 if an edit would change something
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sci(x, digits=3):
    e = len(str(int(x))) - 1
    return "%.*f×10^%d" % (digits, x / 10 ** e, e)


def main(argv):
    with open(os.path.join(ROOT, "profiles", "r06_bench.json")) as f:
        d = json.load(f)
    c = d["configs"]
    v, ms, enc = d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]
    mix = d["roofline"]["valu"]["frac_of_mix_ceiling"]
    edits = {
        "DESIGN.md": [
            (r"headline \*\*[0-9.]+×10\^8/s\*\* \([0-9.]+ ms per 2\^20: hash [0-9.]+ \+ encrypt [0-9.]+;",
             "headline **%s/s** (%.2f ms per 2^20: hash %.2f + encrypt %.2f;" % (sci(v), ms, ms - enc, enc)),
            (r"decaps [0-9.]+×10\^8, pairs [0-9.]+×10\^7, ML-DSA-65 verify [0-9.]+×10\^7 \(3\.11×10\^7 before\), config 5 [0-9.]+×10\^7 mixed items/s; host ABI [0-9.]+×10\^7 pageable /\n"
             r"[0-9.]+×10\^7 page-locked; `frac_of_mix_ceiling` [0-9.]+ \(headline\), [0-9.]+ \(`mldsa_verify_kernel<65>`\), [0-9.]+ \(`<87>`\), [0-9.]+ \(`mlkem_encrypt_kernel<4>`\)",
             "decaps %s, pairs %s, ML-DSA-65 verify %s (3.11×10^7 before), config 5 %s mixed items/s; host ABI %s pageable /\n%s page-locked; `frac_of_mix_ceiling` %.3f (headline), "
             "%.3f (`mldsa_verify_kernel<65>`), %.3f (`<87>`), %.3f (`mlkem_encrypt_kernel<4>`)" % (
                 sci(c["decaps"]["value"]), sci(c["config3"]["value"], 2), sci(c["config4"]["value"], 2), sci(c["config5"]["value"], 2), sci(d["value_host_abi"]["value"], 2),
                 sci(d["value_host_abi"]["pinned"], 2), mix, c["config4"]["roofline"]["valu_frac_of_mix_ceiling"], c["config5"]["roofline_mldsa87"]["valu_frac_of_mix_ceiling"],
                 c["config5"]["roofline_mlkem1024"]["valu_frac_of_mix_ceiling"])),
            (r"[0-9.]+×10\^6/s vectorised \(one parsed key for the batch, the shape of the reference's own benchmark:\n[0-9.]+×10\^6/s\), [0-9.]+×10\^5/s the scalar oracle",
             "%s/s vectorised (one parsed key for the batch, the shape of the reference's own benchmark:\n%s/s), %s/s the scalar oracle" % (
                 sci(d["cpu_baseline"]["value"], 2), sci(d["cpu_baseline"]["shared_key"]["value"], 2), sci(d["cpu_baseline"]["scalar_oracle"]["value"], 2))),
        ],
        "README.md": [
            (r"[0-9.]+×10\^8 ML-KEM-768 encapsulations/s at batch 2\^20", "%s ML-KEM-768 encapsulations/s at batch 2^20" % sci(v)),
            (r"the dominant kernel runs at [0-9.]+ of a\n  VALU ceiling", "the dominant kernel runs at %.3f of a\n  VALU ceiling" % mix),
            (r"[0-9.]+×10\^8 decapsulations/s", "%s decapsulations/s" % sci(c["decaps"]["value"])),
            (r"[0-9.]+×10\^7 ML-DSA-65 verifications/s over 2\^18", "%s ML-DSA-65 verifications/s over 2^18" % sci(c["config4"]["value"], 2)),
            (r"[0-9.]+×10\^7 encapsulations/s \(PCIe-bound", "%s encapsulations/s (PCIe-bound" % sci(d["value_host_abi"]["value"], 2)),
            (r"EPYC 9575F: [0-9.]+×10\^6/s", "EPYC 9575F: %s/s" % sci(d["cpu_baseline"]["value"], 2)),
        ],
    }
    changed = 0
    for name, subs in edits.items():
        path = os.path.join(ROOT, name)
        text = open(path).read()
        new = text
        for pat, rep in subs:
            new, n = re.subn(pat, lambda _m, rep=rep: rep, new)
            if n != 1:
                print("%s: pattern matched %d times: %s" % (name, n, pat[:70]))
                return 2
        if new != text:
            changed += 1
            if "--check" not in argv:
                open(path, "w").write(new)
            print("%s: %s" % (name, "would change" if "--check" in argv else "updated"))
    return 1 if (changed and "--check" in argv) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
