"""CPU, world_size 2, gloo: the N > 1 path (shard -> compute -> all-gather collate) returns
exactly what one process returns.  The per-shard compute is a deterministic stand-in (the real
one needs a GPU); everything else is the product code path of lstm_ode_bci_amd.sharding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_compute(X):
    s = X.reshape(len(X), -1).double().sum(1)
    probs = torch.stack([torch.sigmoid(s), 1 - torch.sigmoid(s)], 1).float()
    traj = (s[:, None, None] * torch.arange(1, 7, dtype=torch.float64).reshape(1, 2, 3))
    pred = (s > 0).long()
    return traj, probs, pred


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lstm_ode_bci_amd import sharding
    X = torch.from_numpy(np.random.default_rng(0).standard_normal((n, 4, 3)).astype(np.float32))
    out = sharding.sharded_apply(X, _fake_compute)
    ref = _fake_compute(X)
    ok = all(torch.equal(a, b) for a, b in zip(out, ref))
    lo, hi = sharding.shard_bounds(n, world, rank)
    q.put((rank, ok, hi - lo))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [10, 7])
def test_two_ranks_collate_bit_identical(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sum(cnt for _, _, cnt in res) == n


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lstm_ode_bci_amd import sharding
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    out, scale = sharding.all_reduce_flat_grad_(g)
    ok = out.data_ptr() == g.data_ptr() and torch.equal(out, torch.arange(1000, dtype=torch.float32) * 3) and scale == 0.5
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_all_reduce_two_ranks():
    """Data-parallel training step: the flat gradient buffer is summed in place, the 1/world factor is returned for
    the optimizer launch."""
    from lstm_ode_bci_amd import sharding
    g = torch.ones(5)
    out, scale = sharding.all_reduce_flat_grad_(g)          # no process group: no-op
    assert out is g and scale == 1.0
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


class _FakeInteg:
    """Stand-in for LSTMODEIntegration on CPU tensors: same predict_batch_device contract, deterministic arithmetic."""
    min_device_chunk = 3

    def _chunk(self, batch_size, respect):
        return max(int(batch_size), self.min_device_chunk)

    seen_amp = ()

    def predict_batch_device(self, X, forecast_steps=20, batch_size=512, want_traj=True, use_amp=None):
        self.seen_amp = self.seen_amp + (use_amp,)
        s = X.reshape(len(X), -1).double().sum(1)
        traj = s[:, None, None] * torch.arange(1, forecast_steps * 3 + 1, dtype=torch.float64).reshape(1, forecast_steps, 3)
        probs = torch.stack([torch.sigmoid(s), 1 - torch.sigmoid(s)], 1).float()
        return (traj if want_traj else None), probs, (s > 0).long()


def _pbs_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lstm_ode_bci_amd import sharding
    X = torch.from_numpy(np.random.default_rng(1).standard_normal((n, 4, 3)).astype(np.float32))
    integ = _FakeInteg()
    ref = integ.predict_batch_device(X, 5, n)
    ok = True
    for want_traj in (True, False):
        out = sharding.predict_batch_sharded(integ, X, forecast_steps=5, batch_size=2, gather_trajectories=want_traj)
        ok &= (out[0] is None) == (not want_traj)
        ok &= all(a is None or torch.equal(a, b) for a, b in zip(out, ref))
    # local_shard=True: every rank passes ONLY its own windows (unequal counts, possibly none): the result is the
    # concatenation in rank order -- the same rows as one process over the concatenated array
    cut = (2 * n) // 3                                  # rank 0 holds [0, cut), rank 1 [cut, n)
    mine = X[:cut] if rank == 0 else X[cut:]
    for want_traj in (True, False):
        out = sharding.predict_batch_sharded(integ, mine, forecast_steps=5, batch_size=2, gather_trajectories=want_traj,
                                             local_shard=True)
        ok &= all(a is None or torch.equal(a, b) for a, b in zip(out, ref))
    ok &= sharding.gather_shard_sizes(len(mine)) == [(0, cut), (cut, n)]
    # gather_to = r: the trajectories are collated on rank r only (dist.gather); probabilities / decisions everywhere
    for dst in (0, 1):
        for local in (False, True):
            out = sharding.predict_batch_sharded(integ, mine if local else X, forecast_steps=5, batch_size=2,
                                                 local_shard=local, gather_to=dst)
            ok &= (out[0] is None) == (rank != dst)
            ok &= all(a is None or torch.equal(a, b) for a, b in zip(out, ref))
    try:
        sharding.predict_batch_sharded(integ, X, forecast_steps=5, gather_to=2)
        ok = False
    except ValueError:
        pass
    # use_amp reaches the per-shard call; unset, the object's own setting decides
    integ.seen_amp = ()
    sharding.predict_batch_sharded(integ, X, forecast_steps=5, batch_size=2, use_amp=True)
    lo_, hi_ = sharding.shard_bounds(n, world, rank)
    ok &= (len(integ.seen_amp) > 0) == (hi_ > lo_) and all(a is True for a in integ.seen_amp)
    integ.seen_amp = ()
    sharding.predict_batch_sharded(integ, X, forecast_steps=5, batch_size=2)
    ok &= all(a is None for a in integ.seen_amp)
    # nothing on any rank (local_shard with empty shards everywhere): empty outputs, no 0-row tensor through the chunk loop
    integ.seen_amp = ()
    e = sharding.predict_batch_sharded(integ, X[:0], forecast_steps=5, batch_size=2, local_shard=True)
    ok &= integ.seen_amp == () and tuple(e[0].shape) == (0, 5, 3) and tuple(e[1].shape) == (0, 2) and tuple(e[2].shape) == (0,)
    e = sharding.predict_batch_sharded(integ, X[:0], forecast_steps=5, batch_size=2)
    ok &= integ.seen_amp == () and len(e[1]) == 0
    t0 = sharding.dp_broadcast_(torch.tensor([float(rank + 5)]), 0)
    ok &= float(t0) == 5.0
    # control-flow helpers of the data-parallel train loop
    try:
        sharding.dp_assert_equal((4 + rank, 2), None, "batches")
        raised = False
    except ValueError:
        raised = True
    same = sharding.dp_assert_equal((4, 2), None, "batches") == (4, 2)
    t = torch.tensor([1.0 + rank, 10.0], dtype=torch.float64)
    sharding.dp_sum_(t)
    ok &= raised and same and torch.equal(t, torch.tensor([3.0, 20.0], dtype=torch.float64))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [11, 4, 1])
def test_predict_batch_sharded_chunked_gather_and_dp_helpers(n):
    """predict_batch_sharded (chunked trajectory gather, ragged shards, a rank with an exhausted / empty shard) is
    bit-identical to one process; dp_assert_equal raises on EVERY rank when batch counts differ (so no rank is left
    in a collective), dp_sum_ sums the epoch metrics that drive early stopping."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pbs_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)
