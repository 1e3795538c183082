#!/usr/bin/env python3
"""Headline benchmark: EEG windows/s through the MI355X-native LSTM-ODE inner loop.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|fwd|coupled] [--batch B]

One "step" = one pass of the hot path over one batch of B synthetic (256 x 61) windows per GPU
(weak scaling: B per GPU is fixed, default 4096 = BASELINE.json configs[2]).  Inputs and
weights are resident in HBM before the timed region.  For N > 1 the driver launches this file
under torch.distributed.run, one rank per GPU (RCCL); windows are sharded over ranks with no
data-path collective; the step ends with the one collective the path needs (all-gather of the
logits for inference, gradient all-reduce for training).

Prints ONE JSON line on rank 0 (contract in the task statement), carrying `roofline` (live HIP
event timing of the dominant kernel vs the fp32 MFMA peak) and `cpu_baseline` (the oracle's
torch-CPU layer stack -- the reference's CPU path semantically -- timed on this host's cores on a
bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T, C, H, L, D = 256, 61, 128, 3, 2
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
GATE_FLOP_FWD = 2 ** 29            # per window, SURVEY.md §8d
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default=None, choices=["train", "fwd", "coupled"])
    ap.add_argument("--batch", type=int, default=4096, help="windows per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hidden", type=int, default=128, help="hidden size (128 = BASELINE configs; 256 = real checkpoints)")
    ap.add_argument("--forecast-steps", type=int, default=300)
    ap.add_argument("--precision", default=None, choices=["fp32", "mixed"],
                    help="mixed = bf16 MFMA inputs for the gate GEMMs under autocast, fp32 recurrence/accumulate "
                         "(BASELINE.json configs[2]); default: mixed for train, fp32 otherwise")
    return ap.parse_args()


def have_backward():
    try:
        from lstm_ode_bci_amd import backward  # noqa: F401
        return True
    except Exception:
        return False


def build_model(dev):
    from lstm_ode_bci_amd import EnhancedLSTMModel
    from lstm_ode_bci_amd import synthetic as syn
    sd = syn.make_state_dict(C, H, L, 2, True)
    m = EnhancedLSTMModel(input_size=C, hidden_size=H, num_layers=L, num_classes=2, dropout=0.4,
                          bidirectional=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev), sd


def kernel_roofline(dev, B, mode, precision):
    """Per-launch durations of the hot kernels of THIS workload, HIP events on the launch stream
    (torch's current stream is the stream every lob_* call is launched on).  Returns
    {name: {sec, flop, bytes, per_step}}; `bytes` = algorithmic HBM bytes of one launch."""
    from lstm_ode_bci_amd import ops
    Bp = ops.ceil32(B)
    g = torch.Generator(device="cpu").manual_seed(1)
    mixed = precision == "mixed"
    train = mode == "train"
    out = {}

    def timeit(fn, n=4):
        fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for s, e in evs:
            s.record()
            fn()
            e.record()
        torch.cuda.synchronize()
        return float(np.mean([s.elapsed_time(e) for s, e in evs])) * 1e-3

    rows, N = T * Bp, D * 4 * H
    K = D * H
    bf16_rec = mixed and ops.bf16_rec(H, ops.PG_BF16)   # bf16-MFMA recurrent kernels: H = 128 and 256
    pe = 2.0 if (bf16_rec and ops.PG_BF16) else 4.0     # bytes per stored pre-activation / saved gate
    de = 2.0 if mixed else 4.0                          # bytes per dP element
    # operand storage types as the step itself uses them: in mixed mode at H = 128 the layer below hands
    # over bf16 activations and the weights are cast once per step -> the LDS-DMA GEMM kernels
    act16 = bf16_rec and ops.dma_ok(K, N, rows)
    xe = 2.0 if act16 else 4.0                          # bytes per inter-layer activation element
    x = torch.randn((rows, K), generator=g).to(dev)
    wih = (torch.rand((N, K), generator=g) * 0.17 - 0.085).to(dev)
    if act16:
        x, w_in = x.to(torch.bfloat16), wih.to(torch.bfloat16)
    else:
        w_in = wih
    bias = torch.zeros(N, device=dev)
    whh = (torch.rand((D, 4 * H, H), generator=g) * 0.17 - 0.085).to(dev)
    sec = timeit(lambda: ops.gate_gemm_x(x, w_in, bias, T, Bp, H, D, True, mixed=mixed))
    out["gate_gemm_x(K=256)"] = {"sec": sec, "flop": 2.0 * rows * N * K, "bytes": rows * (xe * K + pe * N) + xe * N * K,
                                 "per_step": L - 1, "mfma": "bf16" if mixed else "f32"}
    P = ops.gate_gemm_x(x, w_in, bias, T, Bp, H, D, True, mixed=mixed)
    if train:
        Pk = P.clone()

        def rec_fwd():
            Pk.copy_(P)                       # the save-mode kernel overwrites P with the activated gates
            return ops.lstm_rec_fwd(Pk, whh, T, Bp, H, D, True, mixed=mixed)
        t_copy = timeit(lambda: Pk.copy_(P))
        sec = timeit(rec_fwd, n=3) - t_copy
        Y, Cs, _, _ = rec_fwd()
        out["lstm_rec_fwd(save)"] = {"sec": sec, "flop": 2.0 * rows * N * H, "per_step": L,
                                     "mfma": "bf16" if bf16_rec else "f32",
                                     "bytes": rows * (2 * pe * N + 8.0 * K)}   # P in, gates out, c out, Y out
        dY = torch.randn((rows, K), generator=g).to(dev) * 1e-3
        sec = timeit(lambda: ops.lstm_rec_bwd(Pk, Cs, whh, dY, T, Bp, H, D, dp_bf16=mixed), n=3)
        out["lstm_rec_bwd"] = {"sec": sec, "flop": 2.0 * rows * N * H, "per_step": L,
                               "mfma": "bf16" if bf16_rec else "f32",
                               "bytes": rows * (pe * N + 8.0 * K + de * N)}   # gates in, c in, dY in, dP out
        dP, _ = ops.lstm_rec_bwd(Pk, Cs, whh, dY, T, Bp, H, D, dp_bf16=mixed)
        Y16 = Y.to(torch.bfloat16) if (Y is not None and dP.dtype == torch.bfloat16) else None
        if Y16 is not None and ops.can_fuse_dw(dP, x, Y16, T, Bp, H, D):
            # what the step runs: dW_ih and dW_hh of a layer from one pass over dP
            sec = timeit(lambda: ops.lstm_dw(dP, x, Y16, T, Bp, H, D))
            out["lstm_dw(dW_ih+dW_hh)"] = {"sec": sec, "flop": 2.0 * rows * N * (K + H), "per_step": L - 1,
                                           "mfma": "bf16",
                                           "bytes": de * rows * N + xe * rows * K + 2.0 * rows * D * H}
        else:
            dw = torch.zeros((N, K), device=dev)
            sec = timeit(lambda: ops.gemm_tn(dP, x, dw, mixed=mixed))
            out["gemm_tn(dW_ih)"] = {"sec": sec, "flop": 2.0 * rows * N * K, "per_step": L - 1,
                                     "mfma": "bf16" if mixed else "f32",
                                     "bytes": de * rows * N + xe * rows * K}
        wt = wih.t().contiguous()
        if dP.dtype == torch.bfloat16 and ops.dma_ok(N, K, rows):
            wt = wt.to(torch.bfloat16)
        sec = timeit(lambda: ops.gemm_nt(dP, wt, mixed=mixed))
        out["gemm_nt(dX)"] = {"sec": sec, "flop": 2.0 * rows * N * K, "per_step": L - 1,
                              "mfma": "bf16" if mixed else "f32",
                              "bytes": de * rows * N + 4.0 * rows * K}
    else:
        sec = timeit(lambda: ops.lstm_rec_fwd(P, whh, T, Bp, H, D, False, mixed=mixed), n=3)
        out["lstm_rec_fwd"] = {"sec": sec, "flop": 2.0 * rows * N * H, "per_step": L,
                               "mfma": "bf16" if bf16_rec else "f32",
                               "bytes": rows * (pe * N + 4.0 * K)}
    return out


def roofline_of(kr):
    """The dominant kernel (largest time per step) priced against the roofline that bounds it."""
    dom = max(kr, key=lambda k: kr[k]["sec"] * kr[k]["per_step"])
    v = kr[dom]
    allk = {k: {"ms": round(x["sec"] * 1e3, 3), "tflops": round(x["flop"] / x["sec"] / 1e12, 1),
                "GBps": round(x["bytes"] / x["sec"] / 1e9, 0), "launches_per_step": x["per_step"], "mfma": x["mfma"]}
            for k, x in kr.items()}
    if v["mfma"] == "f32":
        a = v["flop"] / v["sec"] / 1e12
        return {"bound": "mfma", "kernel": dom, "achieved": a, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": a / FP32_MFMA_PEAK_TFLOPS, "traffic": None, "flop_per_launch": v["flop"],
                "sec_per_launch": v["sec"], "all": allk}
    a = v["bytes"] / v["sec"] / 1e9              # bf16 GEMMs at K <= 1024 sit under the HBM roof
    return {"bound": "hbm", "kernel": dom, "achieved": a, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": a / HBM_PEAK_GBPS, "traffic": None, "bytes_per_launch": v["bytes"],
            "sec_per_launch": v["sec"], "all": allk}


def _cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(mode, sd, forecast_steps=300):
    """The reference's CPU path (same torch layer stack -> aten::lstm -> oneDNN), bounded sample.  Coupled mode adds
    the reference's step 2 as it runs it: one scipy odeint (LSODA) solve per window in a Python loop, single-threaded
    by construction (06_lstm_ode_integration.py:372-401)."""
    from oracle import torch_cpu_path as TP
    from lstm_ode_bci_amd import synthetic as syn
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    m = TP.build(sd, C, H)
    Bc = 128
    x, y = syn.make_windows(Bc)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    t0 = time.perf_counter()
    best, med, nthreads = 0.0, 0.0, 1
    # oneDNN's RNN primitive does not scale to all host threads; keep the best thread count
    for nt in sorted({min(ncores, n) for n in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        if mode == "train":
            b, md = TP.time_train_step(m, xt, yt, iters=2, warmup=1)
        else:
            b, md = TP.time_forward(m, xt, iters=3, warmup=1)
        if b > best:
            best, med, nthreads = b, md, nt
    what = (f"fwd+bwd train-mode (dropout on, weighted CE), B={Bc}" if mode == "train"
            else f"fwd eval-mode no_grad, B={Bc}") + ", best over 8/16/32/64 threads"
    out = {"value": best, "median": med, "unit": "windows/s", "cores": nthreads, "kind": "port", "host_cpus": os.cpu_count(),
           "cpu_model": _cpu_model_name()}
    if mode == "coupled":
        from oracle import restatement as R
        n = 256
        probs = syn.make_probs(n, seed=7)
        t1 = time.perf_counter()
        R.predict_from_probs(probs, syn.DEFAULT_RATES, 0.5, forecast_steps)          # scipy.integrate.odeint per window
        ode_rate = n / (time.perf_counter() - t1)
        out["lstm_windows_per_s"] = best
        out["ode_solves_per_s_1thread"] = ode_rate
        out["value"] = 1.0 / (1.0 / best + 1.0 / ode_rate)          # LSTM chunk loop, then the per-window ODE loop
        what += f" + {n} odeint solves of {forecast_steps} points on 1 thread"
    out["sample"] = what + f"; torch {torch.__version__} CPU (oneDNN), {time.perf_counter() - t0:.1f}s wall"
    return out


def main():
    global H, GATE_FLOP_FWD
    a = parse()
    if a.hidden != H:
        GATE_FLOP_FWD = GATE_FLOP_FWD * (a.hidden // 128) ** 2 if a.hidden % 128 == 0 else int(GATE_FLOP_FWD * (a.hidden / 128) ** 2)
        H = a.hidden
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with python -m torch.distributed.run --nproc-per-node N")
    # LOB_DIST_BACKEND=gloo + LOB_SHARE_GPU=1: rehearsal of the N > 1 code path on a one-GPU box (all ranks on
    # cuda:0, collectives through host memory); the real multi-GPU run uses nccl (= RCCL over xGMI), one GPU each
    backend = os.environ.get("LOB_DIST_BACKEND", "nccl")
    if os.environ.get("LOB_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from lstm_ode_bci_amd import CognitiveStateODE, LSTMODEIntegration
    from lstm_ode_bci_amd import synthetic as syn

    mode = a.mode or ("train" if have_backward() else "fwd")
    precision = a.precision or ("mixed" if mode == "train" else "fp32")
    B = a.batch
    model, sd = build_model(dev)
    # rank r owns global windows [r*B, (r+1)*B): independent shards, no data-path collective
    x_np, y_np = syn.make_windows(B, T, C, seed=syn.INPUT_SEED + rank)
    x = torch.from_numpy(x_np).to(dev)
    y = torch.from_numpy(y_np).to(dev)
    class_w = torch.tensor([1.0, 1.0], device=dev)
    integ = LSTMODEIntegration(model, CognitiveStateODE(), 0.5)
    gather_buf = torch.empty((world * B, 2), device=dev) if world > 1 else None
    criterion = opt = None
    if mode == "train":
        # the reference's training-step body (04_lstm_model.py:482-512): fwd -> weighted CE -> bwd ->
        # [data-parallel: all-reduce of the flat gradient] -> clip 1.0 + AdamW, all inside the timed step
        from lstm_ode_bci_amd.sharding import all_reduce_flat_grad_
        from lstm_ode_bci_amd.training import FusedAdamW, WeightedCrossEntropy
        criterion = WeightedCrossEntropy(class_w).to(dev)
        opt = FusedAdamW(model.parameters(), lr=3e-4, weight_decay=1e-4)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(precision == "mixed")):
            _step()

    def _step():
        if mode == "train":
            model.train()
            opt.zero_grad()
            loss = criterion(model(x), y)
            loss.backward()
            _, gscale = all_reduce_flat_grad_(opt.flat_grad)        # one 4.55 MB message; no-op at N = 1
            opt.step(clip_grad_norm=1.0, grad_scale=gscale)
        elif mode == "fwd":
            model.eval()
            with torch.no_grad():
                logits = model(x)
                if world > 1:
                    dist.all_gather_into_tensor(gather_buf, logits.contiguous())
        else:
            traj, probs, pred = integ.predict_batch_device(x, forecast_steps=a.forecast_steps, batch_size=B)
            if world > 1:
                dist.all_gather_into_tensor(gather_buf, probs.contiguous())

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        value = world * B * a.steps / dt
        kr = kernel_roofline(dev, B, mode, precision)
        roof = roofline_of(kr)
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # HBM bytes per launch from rocprofv3 --pmc
        if os.path.exists(tpath) and H == 128:           # the committed PMC table is for the H = 128 kernels
            try:
                roof["traffic"] = json.load(open(tpath)).get(f"{roof['kernel']}|{precision}|B{B}")
            except Exception:
                pass
        flop_per_window = GATE_FLOP_FWD * (3 if mode == "train" else 1)
        res = {
            "metric": {"train": "eeg_windows_per_sec_fwd_bwd", "fwd": "eeg_windows_per_sec_fwd",
                       "coupled": "eeg_windows_per_sec_fwd_ode"}[mode],
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": f"BiLSTM(3x128)+attn {mode}, (T=256,C=61) windows, B={B}/GPU, " + ("fp32 exact-MFMA" if precision == "fp32" else
                                                       "bf16 gate GEMMs + fp32 recurrence/accumulate (autocast)")
                                   + (f", RK4 ODE {a.forecast_steps} points" if mode == "coupled" else ""),
                       "batch_per_gpu": B, "global_batch": world * B, "seq_len": T, "channels": C,
                       "hidden": H, "layers": L, "mode": mode, "precision": precision,
                       **({"step": "fwd + weighted CE + bwd + clip 1.0 + AdamW (04_lstm_model.py:482-512)"}
                          if mode == "train" else {}),
                       "collective": ("none" if world == 1 else
                                      ("all_reduce(grads 4.55MB)" if mode == "train" else "all_gather(logits)"))},
            "gate_gemm_tflops_effective": value * flop_per_window / 1e12,
            "gate_gemm_frac_of_fp32_mfma_peak": value * flop_per_window / 1e12 / FP32_MFMA_PEAK_TFLOPS / world,
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(mode, sd, a.forecast_steps)
            res["speedup_vs_cpu_baseline"] = value / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
