"""Multi-GPU execution of the coupled path: one process per GPU, windows sharded
contiguously over ranks, no data-path collective (every window is independent through
04_lstm_model.py:206-222 and 06_lstm_ode_integration.py:372-401), ONE all-gather at the end
to collate per-window outputs (RCCL over xGMI when the backend is "nccl").

Data-parallel training adds one more: the all-reduce of the FLAT gradient buffer (4.55 MB, one
message per step) before the fused clip + AdamW launch -- `all_reduce_flat_grad_`.

The reference is single-process; these are the only collectives the build adds.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n, world, rank):
    """Contiguous balanced split of n windows: the first n % world ranks get one extra."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(local, n_total, group=None, bounds=None):
    """Collate per-rank row blocks into the full (n_total, ...) tensor on every rank.  The blocks are split as in
    ``shard_bounds`` unless ``bounds`` ([(lo, hi)] per rank, contiguous, in rank order) says otherwise.  Uneven shards
    are padded to the largest shard for the collective."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    if bounds is None:
        bounds = [shard_bounds(n_total, world, r) for r in range(world)]
    per = max(h - l for l, h in bounds)
    pad = per - local.shape[0]
    buf = local if pad == 0 else torch.cat([local, local.new_zeros((pad,) + tuple(local.shape[1:]))], 0)
    out = local.new_empty((world * per,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
    if all(h - l == per for l, h in bounds):
        return out
    return torch.cat([out[r * per:r * per + (h - l)] for r, (l, h) in enumerate(bounds)], 0)


def gather_shard_sizes(n_local, group=None, device=None):
    """[(lo, hi)] of every rank's shard in the concatenation of all shards (rank order), from each rank's own count."""
    world = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    sizes = [int(o.item()) for o in out]
    bounds, lo = [], 0
    for sz in sizes:
        bounds.append((lo, lo + sz))
        lo += sz
    return bounds


def sharded_apply(X, fn, group=None):
    """Run ``fn(X[lo:hi]) -> tuple of per-window tensors`` on this rank's shard and collate
    every output over the ranks.  With one rank this is just ``fn(X)``."""
    if not (dist.is_available() and dist.is_initialized()):
        return fn(X)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = len(X)
    lo, hi = shard_bounds(n, world, rank)
    outs = fn(X[lo:hi])
    return tuple(None if o is None else all_gather_rows(o, n, group) for o in outs)


def predict_batch_sharded(integ, X_batch, forecast_steps=20, batch_size=512, gather_trajectories=True, group=None,
                          overlap=True, local_shard=False, use_amp=None, gather_to=None):
    """``LSTMODEIntegration.predict_batch`` over all ranks of the default process group.
    Returns device tensors (trajectories (N,steps,3) f64 | None, probs (N,2) f32, predictions (N,) i64),
    identical on every rank and bit-identical to the single-GPU result.

    ``local_shard=False``: every rank passes ALL N windows and takes its slice (``shard_bounds``) -- simple, but each
    rank's host holds the whole array (4 GB at N = 65,536).  ``local_shard=True``: every rank passes ONLY ITS OWN windows
    (any count, also 0); the result rows are the concatenation of the shards in rank order; per-rank host memory is the
    shard alone (512 MB at 8,192 windows per rank).

    What overlaps what (``overlap=True``, device tensors): the shard is processed in device chunks; the all-gather of
    chunk c's trajectories -- the only message that is bandwidth-relevant (29.5 MB per rank and 4096 windows at 300
    points; probabilities / decisions are 64 + 32 KB) -- is issued on a side stream behind an event and runs while
    the LSTM kernels of chunk c+1 occupy the compute stream; the compute stream joins the side stream once, at the
    end.  Probabilities and decisions are collated with one small all-gather each after the last chunk.

    ``use_amp``: precision of the LSTM pass, as in ``predict_batch`` (None = the object's setting).  ``gather_to=r``: the
    trajectories -- the only large output (472 MB for 65,536 windows x 300 points) -- are collated on rank r ONLY
    (``dist.gather``; the reference's consumer runs on one host, 06:409-440) and every other rank returns None for them;
    probabilities and decisions (96 KB) still reach every rank."""
    amp = {} if use_amp is None else {"use_amp": use_amp}
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return integ.predict_batch_device(X_batch, forecast_steps, batch_size, want_traj=gather_trajectories, **amp)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if gather_to is not None and not (0 <= int(gather_to) < world):
        raise ValueError(f"predict_batch_sharded: gather_to={gather_to} is not a rank of this group (world size {world})")
    keep_traj = gather_to is None or int(gather_to) == rank          # does this rank hold the collated trajectories?
    if local_shard:
        bounds = gather_shard_sizes(len(X_batch), group)
        n = bounds[-1][1]
        x_off = bounds[rank][0]                                  # X_batch holds global rows [x_off, x_off + len)
    else:
        n = len(X_batch)
        bounds = [shard_bounds(n, world, r) for r in range(world)]
        x_off = 0
    lo, hi = bounds[rank]
    if n == 0:                                                   # nothing anywhere (every rank sees the same n): no collective
        dev = _traj_proto(integ, X_batch, forecast_steps).device
        traj0 = torch.empty((0, forecast_steps, 3), dtype=torch.float64, device=dev) if (gather_trajectories and keep_traj) else None
        return (traj0, torch.empty((0, 2), dtype=torch.float32, device=dev), torch.empty((0,), dtype=torch.int64, device=dev))
    per = max(h - l for l, h in bounds)                          # largest shard: every rank walks the same chunk grid
    chunk = integ._chunk(batch_size, False) if hasattr(integ, "_chunk") else max(int(batch_size), 1)
    traj_full = None
    comm = None
    probs_loc, pred_loc = [], []
    pending = []
    for c in range(0, per, chunk):
        len_c = min(chunk, per - c)
        a, b = min(lo + c, hi), min(lo + c + len_c, hi)
        if b > a:
            traj, probs, pred = integ.predict_batch_device(X_batch[a - x_off:b - x_off], forecast_steps,
                                                           max(len_c, int(batch_size)), want_traj=gather_trajectories, **amp)
            probs_loc.append(probs)
            pred_loc.append(pred)
        else:
            traj = None
        if not gather_trajectories:
            continue
        if traj_full is None:
            proto = traj if traj is not None else None
            if proto is None:            # a rank whose shard is exhausted still needs shape/dtype/device of the message
                proto = _traj_proto(integ, X_batch, forecast_steps)
            # the collated result lives on every rank (all-gather) or on rank `gather_to` alone
            traj_full = proto.new_empty(((n if keep_traj else 0),) + tuple(proto.shape[1:]))
            use_stream = bool(overlap) and traj_full.is_cuda
            if use_stream:
                comm = torch.cuda.Stream(device=traj_full.device)
        send = traj_full.new_zeros((len_c,) + tuple(traj_full.shape[1:]))
        if traj is not None:
            send[:b - a] = traj
        recv = traj_full.new_empty(((world * len_c if keep_traj else 0),) + tuple(traj_full.shape[1:]))

        def exchange(send=send, recv=recv, c=c, len_c=len_c):
            if gather_to is None:
                dist.all_gather_into_tensor(recv, send, group=group)
            else:
                dst = dist.get_global_rank(group, int(gather_to)) if group is not None else int(gather_to)
                parts = [recv[r * len_c:(r + 1) * len_c] for r in range(world)] if keep_traj else None
                dist.gather(send, gather_list=parts, dst=dst, group=group)
            if not keep_traj:
                return
            for r, (l, h) in enumerate(bounds):
                valid = min(len_c, max(0, (h - l) - c))
                if valid:
                    traj_full[l + c:l + c + valid] = recv[r * len_c:r * len_c + valid]
        if comm is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(traj_full.device))
            with torch.cuda.stream(comm):
                comm.wait_event(ev)
                exchange()
            for t in (send, recv):
                t.record_stream(comm)
            pending.append((send, recv))
        else:
            exchange()
    if comm is not None:
        torch.cuda.current_stream(traj_full.device).wait_stream(comm)
    if traj_full is not None and not keep_traj:
        traj_full = None
    if probs_loc:
        probs_l, pred_l = torch.cat(probs_loc, 0), torch.cat(pred_loc, 0)
    else:                                 # fewer windows than ranks: this rank still takes part in the collation
        dev = _traj_proto(integ, X_batch, forecast_steps).device
        probs_l = torch.empty((0, 2), dtype=torch.float32, device=dev)
        pred_l = torch.empty((0,), dtype=torch.int64, device=dev)
    return traj_full, all_gather_rows(probs_l, n, group, bounds), all_gather_rows(pred_l, n, group, bounds)


def _traj_proto(integ, X_batch, forecast_steps):
    """An empty trajectory tensor with the right trailing shape / dtype / device (for ranks with no rows in a chunk)."""
    dev = integ._device() if hasattr(integ, "_device") else (X_batch.device if torch.is_tensor(X_batch) else "cpu")
    return torch.empty((0, forecast_steps, 3), dtype=torch.float64, device=dev)


def dp_assert_equal(values, group=None, what="value"):
    """Every rank must hold the same tuple of integers (batches per epoch, ...): a data-parallel loop whose ranks run
    different numbers of collectives hangs.  Gathers the tuples and raises the SAME ValueError on every rank if they
    differ, so that no rank is left waiting in a collective."""
    world = dist.get_world_size(group)
    mine = [int(v) for v in values]
    # device of the collective: the backend's (nccl needs device tensors)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor(mine, dtype=torch.int64, device=dev)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    allv = [tuple(int(x) for x in o.tolist()) for o in out]
    if any(v != allv[0] for v in allv):
        raise ValueError(f"data-parallel ranks disagree on {what}: {allv} (shard the loader evenly, "
                         "e.g. with sharding.shard_bounds and drop the remainder)")
    return allv[0]


def dp_sum_(t, group=None):
    """In-place sum of a small metrics tensor over the ranks (host tensors are routed through the backend's device)."""
    if dist.get_backend(group) == "nccl" and not t.is_cuda:
        d = t.to(torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(d, group=group)
        t.copy_(d)
    else:
        dist.all_reduce(t, group=group)
    return t


def dp_broadcast_(t, src=0, group=None):
    """In-place broadcast of rank `src`'s tensor (the flat parameter buffer at the start of a data-parallel run)."""
    if dist.get_world_size(group) > 1:
        dist.broadcast(t, src=src, group=group)
    return t


def all_reduce_flat_grad_(flat_grad, group=None):
    """Sum the flat gradient buffer over the ranks in place and return ``(flat_grad, 1 / world)``: the factor is
    handed to ``FusedAdamW.step(grad_scale=...)`` so that the mean is taken inside the optimizer launch instead of
    in a separate pass over the buffer.  A no-op (factor 1) without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat_grad, 1.0
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat_grad, group=group)
    return flat_grad, 1.0 / world
