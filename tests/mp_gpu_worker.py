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
