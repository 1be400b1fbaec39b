#!/bin/bash
# tools/verify_ab.sh -- A/B of mldsa_verify_kernel forms on ONE box, alternating: CIRCL_HIP_DSA_VERIFY_PAIR = 0 | 1,
# ML-DSA-65 at 2^18, ML-DSA-87 at 2^16, ML-DSA-44 at 2^18 (distinct GPU-made keys, valid signatures; tools/verify_only.py times 5 calls
# after a warm-up).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
ROUNDS=${ROUNDS:-4}
for r in $(seq 1 $ROUNDS); do
  for pn in "65 18" "87 16" "44 18"; do
    for pair in 0 1; do
      echo -n "round $r PAIR=$pair  "
      CIRCL_HIP_DSA_VERIFY_PAIR=$pair python tools/verify_only.py $pn 2>&1 | grep "ML-DSA"
    done
  done
done | tee /tmp/verify_ab.raw
python - <<'PY'
import re, collections
d = collections.defaultdict(list)
for ln in open("/tmp/verify_ab.raw"):
    m = re.search(r"PAIR=(\d)\s+ML-DSA-(\d+) verify n=(\d+): ([0-9.]+) ms", ln)
    if m: d[(int(m.group(2)), int(m.group(1)))].append(float(m.group(4)))
print()
for p in (44, 65, 87):
    if (p, 0) in d and (p, 1) in d:
        a, b = sorted(d[(p, 0)]), sorted(d[(p, 1)])
        ma, mb = a[len(a) // 2], b[len(b) // 2]
        print("ML-DSA-%d: one-at-a-time median %.3f ms (min %.3f), paired median %.3f ms (min %.3f): %+.1f %%" % (p, ma, a[0], mb, b[0], (ma / mb - 1) * 100))
PY
