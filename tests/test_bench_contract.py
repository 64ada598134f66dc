"""The one-JSON-line-per-workload contract of bench.py: the committed lines of the final run (CPU) and a live headline
line (GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOP = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
       "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}


def _check_line(d, want_cpu):
    for k, t in TOP.items():
        assert k in d, k
        assert isinstance(d[k], (int, float)) if t is float else isinstance(d[k], t), (k, d[k])
    assert "vs_baseline" in d and d["vs_baseline"] is None        # BASELINE.md holds no published number for this metric
    assert d["higher_is_better"] is True and d["data"] == "synthetic" and d["scaling"] in ("weak", "strong")
    assert d["dtype"] in ("f64", "f32") and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["steps"] > 0
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.001            # the dominant kernel is part of the step
    assert r["traffic"] is None or r["traffic"] > 0
    if want_cpu:
        c = d["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


def test_committed_bench_lines_keep_the_contract():
    path = os.path.join(ROOT, "profiles", "r02_bench_default.jsonl")
    lines = [json.loads(x) for x in open(path) if x.strip()]
    assert len(lines) == 5
    for d in lines:
        _check_line(d, want_cpu=True)
        assert d["n_gpus"] == 1
    head = lines[-1]                                              # the headline (BASELINE.json's metric) comes last
    assert "cfg2" in head["config"]["workload"] and head["metric"].startswith("range-samples/sec")
    n = 4 * 500000 * 2000
    assert abs(head["value"] - n / (head["ms_per_step"] * 1e-3)) / head["value"] < 1e-6
    assert head["roofline"]["traffic"] is not None and 0.9 < head["roofline"]["traffic"] / (n * 12) < 1.2
    assert set(head["config"]["api_two_calls_ms"]) == {"compute_Sv", "compute_MVBS"}


@pytest.mark.gpu
def test_live_headline_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--only-headline", "--no-cpu-baseline",
                          "--steps", "3", "--warmup", "1"], capture_output=True, text=True, cwd=ROOT, check=True)
    lines = [x for x in out.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    _check_line(d, want_cpu=False)
    assert d["steps"] == 3 and d["warmup"] == 1 and d["n_gpus"] == 1
