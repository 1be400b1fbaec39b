"""The bench contract's stdout line (bench.contract_line): short enough for the driver to keep, strict JSON, every key the judge reads.
VERDICT r05 item 1: the round-5 line had grown to 20 KB and the driver recorded `parsed: null`.  The stand-in record is a real full record
of an earlier run (profiles/r05_bench.json) plus synthetic worst cases (8 ranks, non-finite floats, over-long strings)."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _strict(text):
    def bad(c):
        raise ValueError("non-finite constant %s in the line" % c)
    return json.loads(text, parse_constant=bad)


def _standin():
    with open(os.path.join(ROOT, "profiles", "r05_bench.json")) as f:
        return json.load(f)


REQUIRED = {
    "": ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
         "config", "parity", "roofline", "cpu_baseline", "value_host_abi", "strong", "extras"),
    "config": ("workload", "key_pool", "parallelism", "mode"),
    "parity": ("sampled_items", "whole_batch", "bit_exact_vs_oracle", "ranks_failing"),
    "roofline": ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms",
                 "traffic_over_algorithmic", "valu", "pmc"),
    "roofline.valu": ("frac", "frac_of_mix_ceiling", "ceiling_source"),
    "roofline.pmc": ("source",),
    "cpu_baseline": ("value", "unit", "cores", "kind", "cpu", "per_thread", "sample", "scalar_oracle", "shared_key"),
    "cpu_baseline.scalar_oracle": ("value",),
    "cpu_baseline.shared_key": ("value",),
    "value_host_abi": ("value", "pinned"),
    "strong": ("value", "items_per_rank"),
}


def _check(line):
    for path, keys in REQUIRED.items():
        d = line
        for part in filter(None, path.split(".")):
            d = d[part]
        for k in keys:
            assert k in d, "%s lacks %r" % (path or "line", k)


def test_line_is_short_strict_and_complete():
    text = bench.contract_line(_standin())
    assert "\n" not in text
    assert len(text) < bench.LINE_MAX == 6144, len(text)
    line = _strict(text)
    _check(line)
    assert line["extras"] == "bench_extras.json"
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["unit"] == "GB/s"
    r = line["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert line["cpu_baseline"]["kind"] in ("port", "reference")
    # the other BASELINE configs keep one figure + verdict each
    for c in ("config3", "config4", "config5"):
        assert line["configs"][c]["value"] > 0
    assert line["configs"]["config4"]["parity"]["bit_exact_vs_oracle"] is True


def test_line_survives_eight_ranks_and_bad_floats(tmp_path):
    out = copy.deepcopy(_standin())
    out["n_gpus"] = 8
    out["per_rank"] = {"encaps_per_s": [1.4e8 + i for i in range(8)]}
    out["strong"]["items_per_rank"] = [1 << 17] * 8
    out["strong"]["per_rank_encaps_per_s"] = [1.7e7] * 8
    out["strong"]["parity"] = {"sampled_items": 4096, "bit_exact_vs_oracle": True, "ranks_failing": 0, "oracle_seconds": 1.0}
    out["config"]["workload"] = "x" * 5000
    out["cpu_baseline"]["sample"] = "y" * 5000
    out["roofline"]["traffic"] = float("nan")
    out["configs"]["config4"]["value"] = float("inf")
    path = tmp_path / "bench_extras.json"
    text = bench.emit(out, str(path))
    assert len(text) < bench.LINE_MAX
    line = _strict(text)
    _check(line)
    assert line["roofline"]["traffic"] is None and line["configs"]["config4"]["value"] is None
    assert line["per_rank"]["encaps_per_s"][7] > 0 and len(line["strong"]["items_per_rank"]) == 8
    full = _strict(path.read_text())       # the extras file is strict JSON too and keeps what the line dropped
    assert "configs" in full and "sustained" in full and full["configs"]["config5"]["workload"]
    assert line["extras"] == "bench_extras.json"


def test_line_without_optional_legs():
    out = copy.deepcopy(_standin())
    for k in ("strong", "value_host_abi", "cpu_baseline"):
        out[k] = None
    out["configs"] = {}
    line = _strict(bench.contract_line(out))
    assert "configs" not in line and line["value"] > 0 and line["roofline"]["frac"] > 0


def test_cpu_baseline_prints_each_figure_once():
    """No `vectorized` block repeating value / per_thread / shared_key (bench.py cpu_baseline)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"vectorized": vec' not in src
