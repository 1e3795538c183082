"""Multi-process readiness on a one-GPU box (``-m gpu``): the N > 1 product code with the REAL model.

Two fresh child processes (tests/mp_gpu_worker.py), both on cuda:0, gloo backend: `predict_batch_sharded` (window
shards + chunked, side-stream trajectory all-gather) must be bit-identical to one process, and one data-parallel
training step -- per-rank shard, `all_reduce_flat_grad_`, `FusedAdamW(grad_scale=1/world)` -- must equal the
single-process step on the concatenated batch.  A world-size-1 `nccl` run loads RCCL once and pushes the bench's two
collectives through it (the 8-GPU curve itself is the driver's to measure).
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from lstm_ode_bci_amd import synthetic as syn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(world, backend, tmp_path, mode=""):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOB_MP_BACKEND=backend, LOB_MP_OUT=str(tmp_path),
                   LOB_MP_MODE=mode, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_gpu_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]


def _single_process_reference():
    from lstm_ode_bci_amd import CognitiveStateODE, EnhancedLSTMModel, LSTMODEIntegration
    from lstm_ode_bci_amd.training import FusedAdamW, WeightedCrossEntropy
    dev = torch.device("cuda:0")
    C, H, T, N, B = 61, 128, 32, 44, 40
    sd = syn.make_state_dict(C, H, 3, 2, True)
    m = EnhancedLSTMModel(C, H, 3, 2, 0.0, True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    x, y = syn.make_windows(N, T, C, seed=21)
    X = torch.from_numpy(x).to(dev)
    integ = LSTMODEIntegration(m, CognitiveStateODE(), 0.5)
    traj, probs, pred = integ.predict_batch_device(X, forecast_steps=12, batch_size=N)
    ref = dict(traj=traj.cpu().numpy(), probs=probs.cpu().numpy(), pred=pred.cpu().numpy())
    m.train()
    opt = FusedAdamW(m.parameters(), lr=3e-4, weight_decay=1e-4)
    crit = WeightedCrossEntropy(torch.tensor([0.7, 1.3])).to(dev)
    yb = torch.from_numpy(y[:B]).to(dev)
    # data-parallel semantics (as DistributedDataParallel): the MEAN over ranks of each rank's weighted-mean loss
    from lstm_ode_bci_amd import sharding
    opt.zero_grad()
    for r in range(2):
        lo, hi = sharding.shard_bounds(B, 2, r)
        (crit(m(X[lo:hi]), yb[lo:hi]) / 2).backward()
    ref["flat_grad_mean"] = opt.flat_grad.cpu().numpy().copy()
    opt.step(clip_grad_norm=1.0)
    ref["flat_param"] = opt.flat_param.cpu().numpy()
    return ref


def test_two_ranks_real_model_sharded_inference_and_dp_step(tmp_path):
    ref = _single_process_reference()
    ranks = _run_ranks(2, "gloo", tmp_path)
    for r, d in enumerate(ranks):
        # inference: pure data movement around independent windows -> bit-identical on every rank
        assert np.array_equal(d["traj"], ref["traj"]), r
        assert np.array_equal(d["probs"], ref["probs"]) and np.array_equal(d["pred"], ref["pred"]), r
        assert np.array_equal(d["probs_no_traj"], ref["probs"]), r
        assert d["gscale"] == 0.5
    # the all-reduced gradient is the same buffer on both ranks, bit for bit
    assert np.array_equal(ranks[0]["flat_grad_mean"], ranks[1]["flat_grad_mean"])
    assert np.array_equal(ranks[0]["flat_param"], ranks[1]["flat_param"])
    g, gr = ranks[0]["flat_grad_mean"], ref["flat_grad_mean"]
    assert np.abs(g - gr).max() <= 2e-6 * np.abs(gr).max() + 1e-10          # fp32 summation order only
    w, wr = ranks[0]["flat_param"], ref["flat_param"]
    solid = np.abs(gr) > 1e-6 * np.abs(gr).max()        # Adam normalises: elements with a ~zero gradient amplify rounding noise
    assert np.abs(w - wr)[solid].max() <= 2e-6, np.abs(w - wr)[solid].max()
    assert np.abs(w - wr).max() <= 6.1e-4               # nothing moves by more than two steps of lr = 3e-4 even there


def test_rccl_world_size_one_smoke(tmp_path):
    """backend "nccl" IS RCCL on ROCm: initialise it, run the sharded predict, the gradient all-reduce and the bench's
    all_gather_into_tensor through it on one rank."""
    ref = _single_process_reference()
    (d,) = _run_ranks(1, "nccl", tmp_path)
    assert np.array_equal(d["traj"], ref["traj"]) and np.array_equal(d["probs"], ref["probs"])
    assert d["gscale"] == 1.0 and np.array_equal(d["nccl_gather"], np.ones((4, 2), np.float32))


def test_data_parallel_train_model_syncs_start_state_and_class_weights(tmp_path):
    """ranks built from different seeds must not drift apart silently.  train_model(data_parallel=True)
    broadcasts rank 0's parameters before the first step and weights the loss by the class counts of the WHOLE
    training set (04_lstm_model.py:430-432), so both ranks end with bit-identical weights and histories."""
    r0, r1 = _run_ranks(2, "gloo", tmp_path, mode="train")
    assert not np.array_equal(r0["w0"], r1["w0"])                 # the ranks really started apart
    assert np.array_equal(r0["w1"], r1["w1"])
    assert np.array_equal(r0["train_loss"], r1["train_loss"]) and np.array_equal(r0["val_f1"], r1["val_f1"])
    assert np.isfinite(r0["w1"]).all() and not np.array_equal(r0["w1"], r0["w0"])
    assert int(r0["pickled"]) > 0


@pytest.mark.parametrize("mode", ["train", "fwd"])
def test_bench_py_two_ranks_under_torch_distributed_run(tmp_path, mode):
    """The command the driver launches for N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N
    ...`), rehearsed on a one-GPU box: a fresh child (started before it touches the GPU), both ranks on cuda:0
    (LOB_SHARE_GPU=1), collectives through gloo.  Checks the contract line: n_gpus, global batch, the collective the
    step ends with, a finite whole-job value, weak scaling."""
    import json
    import math
    port = _free_port()
    env = dict(os.environ, LOB_DIST_BACKEND="gloo", LOB_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--mode", mode]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 prints ONE JSON line
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["warmup"] == 1
    assert res["scaling"] == "weak" and res["higher_is_better"] is True and res["unit"] == "windows/s"
    cfg = res["config"]
    assert cfg["global_batch"] == 2 * cfg["batch_per_gpu"] == 2 * 4096
    assert cfg["collective"] == ("all_reduce(grads 4.55MB)" if mode == "train" else "all_gather(logits)")
    assert math.isfinite(res["value"]) and res["value"] > 0
    # whole-job value: both ranks' windows over the slowest rank's time
    assert abs(res["value"] - 2 * 4096 * 2 / (res["ms_per_step"] * 2 * 1e-3)) <= 1e-6 * res["value"]
    assert "cpu_baseline" not in res and "roofline" in res


@pytest.mark.parametrize("mode", ["train", "fwd"])
def test_bench_py_two_ranks_started_plainly(tmp_path, mode):
    """`python3 bench.py --gpus 2 ...` WITHOUT a launcher around it (the way the driver starts the N = 1
    bench) must start its own ranks as child processes, relay rank 0's ONE JSON line and return the children's exit code."""
    import json
    import math
    env = dict(os.environ, LOB_DIST_BACKEND="gloo", LOB_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--mode", mode]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["warmup"] == 1 and res["scaling"] == "weak"
    cfg = res["config"]
    assert cfg["global_batch"] == 2 * cfg["batch_per_gpu"] == 2 * 4096
    assert cfg["collective"] == ("all_reduce(grads 4.55MB)" if mode == "train" else "all_gather(logits)")
    assert math.isfinite(res["value"]) and res["value"] > 0
    assert abs(res["value"] - 2 * 4096 * 2 / (res["ms_per_step"] * 2 * 1e-3)) <= 1e-6 * res["value"]


def test_bench_py_plain_launch_relays_a_failing_child(tmp_path):
    """The launcher's exit code is the children's: a bad flag must not read as success."""
    env = dict(os.environ, LOB_DIST_BACKEND="no_such_backend", LOB_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
