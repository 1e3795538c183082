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


def test_plain_multi_gpu_launch_builds_the_distributed_command(monkeypatch):
    """`python bench.py --gpus N` (no launcher): bench.launch_ranks starts `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N ... bench.py <same flags>` as a child and returns its exit code -- checked here without a GPU by
    intercepting the child."""
    import subprocess
    import bench
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None, bufsize=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = iter(['{"n_gpus": 4}\n'])

        def wait(self):
            return 7

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = bench.launch_ranks(4)
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_dispatches_to_the_launcher_before_any_gpu_call(monkeypatch):
    import bench
    import torch
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "launch_ranks", lambda n: 5)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *_: (_ for _ in ()).throw(AssertionError("GPU touched")))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 5
    # under a launcher whose world size disagrees with --gpus the rank refuses instead of mis-reporting n_gpus
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "does not match" in str(e.value.code)
