"""bench.py's output contract (CPU-only checks): the committed round-1 bench lines carry every required key, and the
CPU-baseline leg (the only place bench.py touches the oracle) runs and reports what it sampled."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]
ROOF = ["bound", "achieved", "peak", "unit", "frac", "traffic"]
CPU = ["value", "unit", "cores", "kind", "sample"]


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench.json")))
def test_committed_bench_lines_follow_the_contract(name):
    d = json.load(open(os.path.join(ROOT, "profiles", name)))
    for k in REQUIRED:
        assert k in d, (name, k)
    for k in ROOF:
        assert k in d["roofline"], (name, k)
    assert d["roofline"]["bound"] in ("hbm", "mfma") and d["roofline"]["unit"] in ("GB/s", "TFLOP/s")
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["dtype"] in ("f32", "bf16") and d["higher_is_better"] is True
    assert abs(d["value"] - d["n_gpus"] * d["config"]["batch_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    if "cpu_baseline" in d:
        for k in CPU:
            assert k in d["cpu_baseline"], (name, k)
        assert d["cpu_baseline"]["kind"] in ("port", "reference")


def test_cpu_baseline_leg_runs_and_describes_its_sample():
    import bench
    from lstm_ode_bci_amd import synthetic as syn
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    r = bench.cpu_baseline("coupled", sd, forecast_steps=20, budget_s=25.0)
    for k in CPU:
        assert k in r
    assert r["kind"] == "port" and r["unit"] == "windows/s" and r["cores"] >= 1
    assert r["value"] < r["lstm_windows_per_s"] and r["ode_solves_per_s_1thread"] > 0
    assert "odeint" in r["sample"] and "threads" in r["sample"]
    # SURVEY.md §8d protocol: n = 1 and best-n, B = 32 and 256, iteration counts recorded
    assert set(r["table"]) == {"best_n_B32", "best_n_B256", "n1_B32", "n1_B256"}
    assert all(v["iters"] >= 2 for v in r["table"].values()) and r["table"]["n1_B32"]["threads"] == 1
    assert r["n1_value"] > 0
