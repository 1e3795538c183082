"""Child process of tests/test_gpu_multiprocess.py (one rank).  Started fresh -- before anything in it touches the GPU --
with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOB_MP_BACKEND / LOB_MP_OUT in the environment.  All ranks use
cuda:0 (a one-GPU box); with gloo the collectives go through host memory, the code path above them is the product's.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def train_mode(rank, world, dev, out):
    """train_model(data_parallel=True) with ranks that were built from DIFFERENT seeds and hold shards with different
    class balance: the run must start from rank 0's weights and weight the loss by the GLOBAL class counts."""
    import pickle
    from lstm_ode_bci_amd import EnhancedLSTMModel
    from lstm_ode_bci_amd import synthetic as syn
    from lstm_ode_bci_amd.training import DeviceWindowLoader, train_model
    C, H, T = 13, 32, 16
    torch.manual_seed(100 + rank)                   # different initial weights per rank
    m = EnhancedLSTMModel(C, H, 2, 2, 0.0, True).to(dev)
    w0 = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu().numpy()
    x, _ = syn.make_windows(48, T, C, seed=31 + rank)
    y = np.zeros(48, dtype=np.int64)
    y[: (8 if rank == 0 else 30)] = 1               # 8 / 48 positives on rank 0, 30 / 48 on rank 1
    xv, yv = syn.make_windows(16, T, C, seed=41 + rank)
    tl = DeviceWindowLoader(x, y, 12, "sequential", dev)
    vl = DeviceWindowLoader(xv, yv, 16, "sequential", dev)
    m, hist = train_model(m, tl, vl, y, epochs=2, learning_rate=1e-3, patience=5, warmup_epochs=1,
                          gradient_accumulation_steps=2, use_amp=False, verbose=False, data_parallel=True)
    w1 = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu().numpy()
    blob = pickle.dumps(m)                          # the trained model stays picklable (no weakref attribute on it)
    np.savez(os.path.join(out, f"rank{rank}.npz"), w0=w0, w1=w1, train_loss=np.array(hist["train_loss"]),
             val_f1=np.array(hist["val_f1"]), pickled=np.int64(len(blob)))
    dist.barrier()
    dist.destroy_process_group()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("LOB_MP_BACKEND", "gloo")
    out = os.environ["LOB_MP_OUT"]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    if os.environ.get("LOB_MP_MODE") == "train":
        return train_mode(rank, world, dev, out)
    from lstm_ode_bci_amd import CognitiveStateODE, EnhancedLSTMModel, LSTMODEIntegration, sharding
    from lstm_ode_bci_amd import synthetic as syn
    from lstm_ode_bci_amd.training import FusedAdamW, WeightedCrossEntropy

    C, H, T, N = 61, 128, 32, 44
    sd = syn.make_state_dict(C, H, 3, 2, True)
    m = EnhancedLSTMModel(C, H, 3, 2, 0.0, True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    x, y = syn.make_windows(N, T, C, seed=21)
    X = torch.from_numpy(x).to(dev)
    integ = LSTMODEIntegration(m, CognitiveStateODE(), 0.5)
    integ.min_device_chunk = 8                      # several chunks per shard: exercises the side-stream gather
    res = {}
    traj, probs, pred = sharding.predict_batch_sharded(integ, X, forecast_steps=12, batch_size=8)
    res.update(traj=traj.cpu().numpy(), probs=probs.cpu().numpy(), pred=pred.cpu().numpy())
    _, probs2, _ = sharding.predict_batch_sharded(integ, X, forecast_steps=12, batch_size=8, gather_trajectories=False,
                                                 overlap=False)
    res["probs_no_traj"] = probs2.cpu().numpy()

    # one data-parallel training step (fp32 path, dropout 0): this rank's shard of a 40-window global batch
    B = 40
    lo, hi = sharding.shard_bounds(B, world, rank)
    m.train()
    opt = FusedAdamW(m.parameters(), lr=3e-4, weight_decay=1e-4)
    crit = WeightedCrossEntropy(torch.tensor([0.7, 1.3])).to(dev)
    opt.zero_grad()
    loss = crit(m(X[lo:hi]), torch.from_numpy(y[lo:hi]).to(dev))
    loss.backward()
    _, gscale = sharding.all_reduce_flat_grad_(opt.flat_grad)
    res["flat_grad_mean"] = (opt.flat_grad * gscale).cpu().numpy()
    opt.step(clip_grad_norm=1.0, grad_scale=gscale)
    res["flat_param"] = opt.flat_param.cpu().numpy()
    res["gscale"] = np.float64(gscale)
    if backend == "nccl":                           # RCCL code path: the collectives of bench.py, on device buffers
        logits = torch.full((4, 2), float(rank + 1), device=dev)
        gb = torch.empty((world * 4, 2), device=dev)
        dist.all_gather_into_tensor(gb, logits)
        res["nccl_gather"] = gb.cpu().numpy()
    np.savez(os.path.join(out, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
