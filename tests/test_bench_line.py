"""The line bench.py prints must stay parseable by the driver: one JSON object of less than 4 KB carrying the contract's
fields, `roofline`, `cpu_baseline` and one scalar per leg (round 5's 21 KB line was recorded as `parsed: null`).  The
formatter is run on the full results committed under profiles/ (canned: no GPU needed) and on a worst case."""
import glob
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def canned():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_default.json")) +
                   glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_detail.json")))
    assert files
    return files


def test_line_of_every_committed_full_result_is_short_and_complete():
    seen_big = False
    for f in canned():
        full = json.load(open(f))
        seen_big = seen_big or len(json.dumps(full)) > 15000
        line = bench.compact_line(full, "bench_detail.json")
        text = json.dumps(line)
        assert len(text) < bench.LINE_LIMIT, (f, len(text))
        assert "\n" not in text
        if "roofline" in full and "cpu_baseline" in full:
            for k in CONTRACT:
                assert k in line, (f, k)
            assert line["value"] == float("%.9g" % full["value"])
            assert isinstance(line["config"]["workload"], str) and line["config"]["workload"]
            for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
                assert k in line["roofline"], (f, k)
            for k in ("value", "unit", "cores", "kind", "sample"):
                assert k in line["cpu_baseline"], (f, k)
        for leg in ("frontend", "multi_sequence", "sharded", "semantic_elas", "elas", "msa"):
            if isinstance(full.get(leg), dict) and "value" in full[leg]:
                assert isinstance(line[leg], float), (f, leg)        # ONE scalar per leg
    assert seen_big        # (the 21 KB result of round 5 is among them)


def test_line_stays_below_the_limit_when_every_text_field_is_huge():
    full = json.load(open(canned()[-1]))
    full["config"]["workload"] = "w" * 5000
    full["config"]["parallelism"] = "p" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    for leg in bench.LEG_NAMES:
        full.setdefault(leg, {"value": 1.0, "note": "n" * 3000})
    text = json.dumps(bench.compact_line(full, "bench_detail.json"))
    assert len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line


def test_failed_leg_shows_its_error_not_a_dict():
    full = json.load(open(canned()[-1]))
    full["msa"] = {"error": "RuntimeError('x')"}
    assert bench.compact_line(full)["msa"] == "RuntimeError('x')"
