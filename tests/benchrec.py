"""What the tests that run bench.py share: the contract line and the full record behind it."""
import json
import os


def bench_record(r, extras_file):
    """bench.py prints the contract line (< 6 KB: metric, value, config, parity, roofline, cpu_baseline, one figure per BASELINE config) and
    writes the full record to --extras-file: returns (line, record) after checking that the line is what the driver can keep."""
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines  # rank 0 only, once
    assert len(lines[0]) < 6144, len(lines[0])

    def bad(c):
        raise ValueError(c)
    line = json.loads(lines[0], parse_constant=bad)
    assert line["extras"] == os.path.basename(extras_file)
    with open(extras_file) as f:
        rec = json.load(f, parse_constant=bad)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling"):
        assert k in line and (line[k] == rec[k] or abs(line[k] - rec[k]) <= 1e-5 * abs(rec[k])), k  # (the line rounds to 6 digits)
    return line, rec
