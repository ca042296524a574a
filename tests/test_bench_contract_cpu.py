"""The bench.py output contract, checked on the committed results of the real runs (profiles/): every key the
driver reads is present, and the numbers are internally consistent."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


@pytest.mark.parametrize("name", ["r01_bench_default.json", "r01_bench_default_under_rocprof.json", "r01_bench_sequential.json"])
def test_committed_bench_lines_follow_the_contract(name):
    path = os.path.join(ROOT, "profiles", name)
    text = open(path).read().strip()
    assert "\n" not in text, "bench.py prints ONE JSON line"
    d = json.loads(text)
    for k in REQUIRED:
        assert k in d, k
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == baseline["metric"] and d["unit"] == "shapes/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["resolution"] == 512 and d["config"]["diffusion_steps"] == 1000 and d["config"]["shapes_per_gpu"] == 8
    shapes = d["n_gpus"] * d["config"]["shapes_per_gpu"] * d["steps"]
    assert d["value"] == pytest.approx(shapes / (d["ms_per_step"] * 1e-3 * d["steps"]), rel=1e-6)
    r = d["roofline"]
    for k in ["bound", "achieved", "peak", "unit", "frac", "traffic"]:
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0 < r["frac"] < 1
    # achieved = (algorithmic FLOP per point x points, x3 issued products in f16x2 mode) / kernel time
    assert r["achieved"] == pytest.approx(r["algorithmic_tflops"] * r["mfma_flop_per_point"] / r["flop_per_point"], rel=1e-9)
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        for k in ["value", "unit", "cores", "kind", "sample"]:
            assert k in c, k
        assert c["kind"] in ("reference", "port") and c["unit"] == d["unit"] and c["cores"] >= 1
        assert d["value"] / c["value"] > 10          # north_star: >= 10x the reference CPU path
