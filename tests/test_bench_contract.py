"""The committed bench line (profiles/rNN_bench.json) carries every field the benchmark contract
names; bench.py parses.  (CPU: no GPU needed.)"""
import ast
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_py_parses_and_defaults_finish_quickly():
    src = open(os.path.join(ROOT, "bench.py")).read()
    ast.parse(src)
    assert '"--gpus", type=int, default=1' in src
    assert '"--steps", type=int, default=30' in src and '"--warmup", type=int, default=10' in src


def test_committed_bench_lines_follow_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line under profiles/"
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for fn in files:
        j = json.loads(open(fn).read().strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                  "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in j, (fn, k)
        assert j["unit"] == "frames/s" and j["higher_is_better"] is True and j["scaling"] == "weak"
        assert j["vs_baseline"] is None and not base.get("published")      # no published number for this metric
        assert j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
        assert abs(j["value"] - j["n_gpus"] * j["config"]["frames_per_gpu"] / (j["ms_per_step"] * 1e-3)) < 0.01 * j["value"]
        r = j["roofline"]
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 0.01 * r["achieved"]
        c = j["cpu_baseline"]
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
