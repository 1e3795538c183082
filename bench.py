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
event timing of the dominant kernel against the roof that bounds it) and `cpu_baseline` (the oracle's
torch-CPU layer stack -- the reference's CPU path semantically -- timed on this host's cores on a
bounded sample; rank 0, N = 1 only).  At N = 1 with the default mode the same run also times the other
BASELINE.json configurations in short legs and embeds them under `extra_configs` (--no-extra skips them):
configs[1] fp32 forward at B = 1024, configs[3] the coupled LSTM -> ODE path at B = 4096 / 300 points
(device-resident AND numpy-in / numpy-out through `predict_batch`), the H = 256 mixed training step (the
reference's real checkpoint size, 04_lstm_model.py:877) and the B = 8192 forward (configs[4]'s per-rank shard).
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T, C, L, D = 256, 61, 3, 2
# MI355X_MICROARCH.md: dense MFMA peaks per arithmetic dtype, HBM3E spec peak
# "f16x2": the fp32 path's default arithmetic at H = 128 -- every fp32 product as three fp16 MFMAs (two-way operand
# split, 22-bit products, fp32 accumulate): the peak in fp32-equivalent FLOPs is a third of the dense fp16 peak
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16x2": 2500.0 / 3}


def fp32_mfma_kind(H):
    """Matrix arithmetic of the fp32 path: 'f16x2' where the split kernels run (H = 128, LOB_VAR_F32_SPLIT), else the
    exact-fp32 MFMA."""
    from lstm_ode_bci_amd import _lib
    return "f16x2" if (H == 128 and _lib.get_variant("F32_SPLIT") != 0) else "f32"
HBM_PEAK_GBPS = 8000.0


def gate_flop_fwd(H):
    """Gate-GEMM FLOPs per window, forward (2^29 at H = 128, 2^31 at H = 256)."""
    return 2.0 * T * D * 4 * H * (2 * H + (L - 1) * 3 * H)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default=None, choices=["train", "fwd", "coupled"])
    ap.add_argument("--batch", type=int, default=4096, help="windows per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra BASELINE-config legs")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the per-kernel HIP-event timings (profiler passes that must contain the steps only)")
    ap.add_argument("--detail", default=None, help="also write the verbose result (every kernel, every leg) to this file")
    ap.add_argument("--strict-storage", action="store_true",
                    help="mixed path with fp32 saved cell states / gradient carries / last-layer output (configs[2] read strictly)")
    ap.add_argument("--hidden", type=int, default=128, help="hidden size (128 = BASELINE configs; 256 = real checkpoints)")
    ap.add_argument("--forecast-steps", type=int, default=300)
    ap.add_argument("--precision", default=None, choices=["fp32", "mixed"],
                    help="mixed = bf16 MFMA inputs for the gate GEMMs under autocast, fp32 recurrence/accumulate "
                         "(BASELINE.json configs[2]); default: mixed for train, fp32 otherwise")
    return ap.parse_args()


def build_model(dev, H):
    from lstm_ode_bci_amd import EnhancedLSTMModel
    from lstm_ode_bci_amd import synthetic as syn
    sd = syn.make_state_dict(C, H, L, 2, True)
    m = EnhancedLSTMModel(input_size=C, hidden_size=H, num_layers=L, num_classes=2, dropout=0.4,
                          bidirectional=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev), sd


def kernel_roofline(dev, B, mode, precision, H):
    """Per-launch durations of the hot kernels of THIS workload, HIP events on the launch stream
    (torch's current stream is the stream every lob_* call is launched on).  Returns
    {name: {sec, flop, bytes, per_step, mfma}}; `bytes` = algorithmic HBM bytes of one launch."""
    from lstm_ode_bci_amd import ops
    Bp = ops.ceil32(B)
    g = torch.Generator(device="cpu").manual_seed(1)
    mixed = precision == "mixed"
    train = mode == "train"
    out = {}

    def timeit(fn, n=4, before=None):
        """Mean duration of fn() between two HIP events on the launch stream; `before` (untimed) runs ahead of the start
        event of every repetition (restores an operand the kernel overwrites)."""
        if before is not None:
            before()
        fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for s, e in evs:
            if before is not None:
                before()
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
    ce = 2.0 if ops.c_bf16_ok(H, mixed, bf16_rec and ops.PG_BF16) else 4.0     # bytes per saved cell state
    ye = 2.0 if ops.dy_bf16_ok(H, mixed) else 4.0       # bytes per element of the gradient carried between layers
    # operand storage types as the step itself uses them: in mixed mode the layer below hands over bf16
    # activations and the weights are cast once per step -> the bf16 x bf16 GEMM kernels
    act16 = bf16_rec and (ops.gate_ws_ok(K, H) or ops.dma_ok(K, N, rows))
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
    f32k = fp32_mfma_kind(H)
    out[f"gate_gemm_x(K={K})"] = {"sec": sec, "flop": 2.0 * rows * N * K, "bytes": rows * (xe * K + pe * N) + xe * N * K,
                                  "per_step": L - 1, "mfma": "bf16" if mixed else f32k}
    P = ops.gate_gemm_x(x, w_in, bias, T, Bp, H, D, True, mixed=mixed)
    if train:
        Pk = P.clone()

        def rec_fwd():
            return ops.lstm_rec_fwd(Pk, whh, T, Bp, H, D, True, mixed=mixed)
        # the save-mode kernel overwrites P with the activated gates: P is restored AHEAD of the start event (round 2
        # timed copy + kernel and subtracted a separately timed copy: noise of the order of the difference)
        sec = timeit(rec_fwd, n=4, before=lambda: Pk.copy_(P))
        Pk.copy_(P)
        Y, Cs, _, _ = rec_fwd()
        out["lstm_rec_fwd(save)"] = {"sec": sec, "flop": 2.0 * rows * N * H, "per_step": L,
                                     "mfma": "bf16" if bf16_rec else f32k,
                                     "bytes": rows * (2 * pe * N + (ce + 4.0) * K)}   # P in, gates out, c out, Y out
        dY = (torch.randn((rows, K), generator=g).to(dev) * 1e-3).to(torch.bfloat16 if ye == 2.0 else torch.float32)
        sec = timeit(lambda: ops.lstm_rec_bwd(Pk, Cs, whh, dY, T, Bp, H, D, dp_bf16=mixed), n=3)
        out["lstm_rec_bwd"] = {"sec": sec, "flop": 2.0 * rows * N * H, "per_step": L,
                               "mfma": "bf16" if bf16_rec else "f32",
                               "bytes": rows * (pe * N + (ce + ye) * K + de * N)}   # gates in, c in, dY in, dP out
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
        dx16 = ye == 2.0 and wt.dtype == torch.bfloat16
        sec = timeit(lambda: ops.gemm_nt(dP, wt, mixed=mixed, out_bf16=dx16))
        out["gemm_nt(dX)"] = {"sec": sec, "flop": 2.0 * rows * N * K, "per_step": L - 1,
                              "mfma": "bf16" if mixed else "f32",
                              "bytes": de * rows * N + (2.0 if dx16 else 4.0) * rows * K}
    else:
        sec = timeit(lambda: ops.lstm_rec_fwd(P, whh, T, Bp, H, D, False, mixed=mixed), n=3)
        out["lstm_rec_fwd"] = {"sec": sec, "flop": 2.0 * rows * N * H, "per_step": L,
                               "mfma": "bf16" if bf16_rec else f32k,
                               "bytes": rows * (pe * N + 4.0 * K)}
    return out


def instep_kernel_times(leg, kr, nsteps=2):
    """Durations of the hot kernels INSIDE real steps of this leg: for `nsteps` further steps every call of the five
    operators that matches a row of `kr` is bracketed by two HIP events on the launch stream (torch's current stream is
    the stream the C-ABI launches on); returns {row: mean seconds per launch}.  This is the operating point rocprofv3's
    per-kernel average of the same command describes (profiles/rNN_*_kernel_stats.csv); the isolated back-to-back
    timing of kernel_roofline() stays beside it as `ms_isolated` (same kernel, random operands, nothing between
    launches: a few per cent to 10 % apart from the in-step figure, depending on the box)."""
    from lstm_ode_bci_amd import ops
    H, Bp = leg.H, ops.ceil32(leg.B)
    rows, N, K = T * Bp, D * 4 * H, D * H
    rec = []

    def key_gate(inp, *a, **k):
        return f"gate_gemm_x(K={inp.shape[1]})"

    def key_fwd(P, whh, T_, Bp_, H_, D_, save, *a, **k):
        return "lstm_rec_fwd(save)" if save else "lstm_rec_fwd"

    def key_dw(dP, inp, *a, **k):
        return "lstm_dw(dW_ih+dW_hh)" if inp.shape[1] == K else None

    def key_nt(a_, w, *a, **k):
        return "gemm_nt(dX)" if (tuple(a_.shape) == (rows, N) and tuple(w.shape) == (K, N)) else None

    keyfns = {"gate_gemm_x": key_gate, "lstm_rec_fwd": key_fwd, "lstm_rec_bwd": lambda *a, **k: "lstm_rec_bwd",
              "lstm_dw": key_dw, "gemm_nt": key_nt}
    saved = {}

    def wrap(name):
        orig, keyfn = getattr(ops, name), keyfns[name]

        def f(*a, **k):
            key = keyfn(*a, **k)
            if key not in kr:
                return orig(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(*a, **k)
            e.record()
            # algorithmic bytes of THIS launch from the tensors it really read and wrote (the recurrent kernels: the
            # step's forward writes bf16 Y (+ a bf16 dropped copy) where the isolated launch writes one fp32 Y)
            nb = None
            if name == "lstm_rec_fwd":
                nb = a[0].nbytes * (2 if key.endswith("(save)") else 1) + sum(t.nbytes for t in r if t is not None)
            elif name == "lstm_rec_bwd":
                nb = a[0].nbytes + a[1].nbytes + a[3].nbytes + r[0].nbytes
            rec.append((key, s, e, nb))
            return r
        saved[name] = orig
        setattr(ops, name, f)

    for name in keyfns:
        wrap(name)
    try:
        for _ in range(nsteps):
            leg.step()
        torch.cuda.synchronize()
    finally:
        for name, orig in saved.items():
            setattr(ops, name, orig)
    out, nbytes = {}, {}
    for key, s, e, nb in rec:
        out.setdefault(key, []).append(s.elapsed_time(e) * 1e-3)
        if nb is not None:
            nbytes.setdefault(key, []).append(nb)
    return {k: (float(np.mean(v)), len(v) // nsteps, float(np.mean(nbytes[k])) if k in nbytes else None)
            for k, v in out.items()}


def roofline_of(kr):
    """The dominant kernel (largest time per step) priced against the roofline that bounds it: the exact-fp32 MFMA
    kernels sit under the fp32 MFMA roof; every 16-bit-MFMA kernel of this path (bf16, and the fp16-split fp32 kernels;
    K <= 1024) sits under the HBM roof.  `all` lists every hot kernel with BOTH fractions against its own arithmetic
    dtype's peaks."""
    # dominant = largest time per step; the recurrent forward and BPTT are within a few per cent of each other (same
    # bytes, same launches; rocprofv3 lists the forward as two instantiations and BPTT as the top symbol): inside
    # 10 % the fixed preference below decides, so that every box names the same kernel
    pref = ["lstm_rec_bwd", "lstm_rec_fwd(save)", "lstm_rec_fwd"]
    tmax = max(kr[k]["sec"] * kr[k]["per_step"] for k in kr)
    near = [k for k in kr if kr[k]["sec"] * kr[k]["per_step"] >= 0.90 * tmax]
    dom = next((k for k in pref if k in near), max(near, key=lambda k: kr[k]["sec"] * kr[k]["per_step"]))
    v = kr[dom]
    allk = {k: {"ms": round(x["sec"] * 1e3, 3), **({"ms_isolated": round(x["sec_isolated"] * 1e3, 3)} if "sec_isolated" in x else {}),
                "tflops": round(x["flop"] / x["sec"] / 1e12, 1),
                "GBps": round(x["bytes"] / x["sec"] / 1e9, 0), "launches_per_step": x["per_step"], "mfma": x["mfma"],
                "frac_of_mfma_peak": round(x["flop"] / x["sec"] / 1e12 / MFMA_PEAK_TFLOPS[x["mfma"]], 4),
                "frac_of_hbm_peak": round(x["bytes"] / x["sec"] / 1e9 / HBM_PEAK_GBPS, 4)}
            for k, x in kr.items()}
    if v["mfma"] == "f32":
        a = v["flop"] / v["sec"] / 1e12
        return {"bound": "mfma", "kernel": dom, "achieved": a, "peak": MFMA_PEAK_TFLOPS["f32"], "unit": "TFLOP/s",
                "frac": a / MFMA_PEAK_TFLOPS["f32"], "traffic": None, "flop_per_launch": v["flop"],
                "sec_per_launch": v["sec"], "all": allk}
    a = v["bytes"] / v["sec"] / 1e9
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


def _time_cpu(fn, budget_s, min_iters=5, warmup=2):
    """SURVEY.md §8d / BASELINE.md §3 protocol: `warmup` untimed calls, then >= `min_iters` timed calls (fewer only
    when the budget is exhausted, never fewer than 2); returns the list of durations."""
    t_start = time.perf_counter()
    for _ in range(warmup):
        fn()
    ts = []
    while len(ts) < min_iters:
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        if len(ts) >= 2 and time.perf_counter() - t_start + ts[-1] > budget_s:
            break
    return ts


def cpu_baseline(mode, sd, forecast_steps=300, H=128, budget_s=100.0):
    """The reference's CPU path (same torch layer stack -> aten::lstm -> oneDNN), bounded sample, SURVEY.md §8d
    protocol: 2 warm-ups, then min / median of >= 5 timed iterations (fewer only if the time budget runs out; the
    count is recorded), at n = 1 thread and at the best of 8/16/32/64 threads (oneDNN's RNN primitive does not scale
    to all host threads), for B = 32 and B = 256.  `value` = the best rate found (the number the GPU must beat).
    Coupled mode adds the reference's step 2 as it runs it: one scipy odeint (LSODA) solve per window in a Python
    loop, single-threaded by construction (06_lstm_ode_integration.py:372-401)."""
    from oracle import torch_cpu_path as TP
    from lstm_ode_bci_amd import synthetic as syn
    import torch.nn as nn
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    m = TP.build(sd, C, H)
    t_all = time.perf_counter()
    data = {Bc: syn.make_windows(Bc) for Bc in (32, 256)}
    w = torch.tensor([1.0, 1.0])

    def one(Bc):
        xt, yt = torch.from_numpy(data[Bc][0]), torch.from_numpy(data[Bc][1])
        if mode == "train":
            def fn():
                m.train()
                m.zero_grad(set_to_none=True)
                nn.functional.cross_entropy(m(xt), yt, weight=w).backward()
        else:
            def fn():
                m.eval()
                with torch.no_grad():
                    m(xt, return_attention=True)
        return fn

    # thread count: probed at EACH batch size that is reported (the B = 32 optimum was reused at B = 256)
    probe, best_nb = {}, {}
    for Bc in (32, 256):
        probe[Bc] = {}
        for nt in sorted({min(ncores, n) for n in (8, 16, 32, 64)}):
            torch.set_num_threads(nt)
            ts = _time_cpu(one(Bc), budget_s=budget_s / (12 if Bc == 32 else 8), min_iters=2, warmup=1)
            probe[Bc][nt] = Bc / min(ts)
        best_nb[Bc] = max(probe[Bc], key=probe[Bc].get)
    table = {}
    share = budget_s * 0.6 / 4
    for tag in ("best_n", "n1"):
        for Bc in (32, 256):
            nt = best_nb[Bc] if tag == "best_n" else 1
            torch.set_num_threads(nt)
            ts = _time_cpu(one(Bc), budget_s=share)
            table[f"{tag}_B{Bc}"] = {"threads": nt, "batch": Bc, "iters": len(ts), "windows_per_s_min_time": Bc / min(ts),
                                     "windows_per_s_median": Bc / float(np.median(ts))}
    bestk = max((k for k in table if k.startswith("best_n")), key=lambda k: table[k]["windows_per_s_min_time"])
    best = table[bestk]["windows_per_s_min_time"]
    best_n = table[bestk]["threads"]
    torch.set_num_threads(best_n)
    what = (("fwd+bwd train-mode (dropout on, weighted CE)" if mode == "train" else "fwd eval-mode no_grad") +
            f", H={H}, B=32 and 256, n=1 and best-n threads (best of 8/16/32/64 probed at each B: {best_nb[32]} at B=32, "
            f"{best_nb[256]} at B=256), 2 warm-ups + "
            f"{'/'.join(str(table[k]['iters']) for k in table)} timed iterations; value = best (min-time) rate, {bestk}")
    out = {"value": best, "median": table[bestk]["windows_per_s_median"], "unit": "windows/s", "cores": best_n,
           "kind": "port", "host_cpus": os.cpu_count(), "cpu_model": _cpu_model_name(),
           "n1_value": max(table["n1_B32"]["windows_per_s_min_time"], table["n1_B256"]["windows_per_s_min_time"]),
           "table": table, "thread_probe": {f"B{b}": v for b, v in probe.items()}}
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
    out["sample"] = what + f"; torch {torch.__version__} CPU (oneDNN), {time.perf_counter() - t_all:.1f}s wall"
    return out


class Leg:
    """One workload: model + inputs resident in HBM, a step() closure, and its JSON description."""

    def __init__(self, dev, mode, precision, B, H, forecast_steps, world=1, rank=0, dist=None, api_level=False,
                 strict_storage=False, exact_fp32=False):
        from lstm_ode_bci_amd import CognitiveStateODE, LSTMODEIntegration
        from lstm_ode_bci_amd import synthetic as syn
        self.mode, self.precision, self.B, self.H, self.world, self.dist = mode, precision, B, H, world, dist
        # strict_storage: the three storage choices beyond "bf16 gate GEMMs + fp32 recurrence" switched back to fp32;
        # exact_fp32: the exact-fp32 MFMA kernels instead of the two-way fp16 split (LOB_VAR_F32_SPLIT = 0)
        self.strict_storage, self.exact_fp32 = bool(strict_storage), bool(exact_fp32)
        self.forecast_steps, self.api_level, self.dev = forecast_steps, api_level, dev
        self.model, self.sd = build_model(dev, H)
        # rank r owns global windows [r*B, (r+1)*B): independent shards, no data-path collective
        self.x_np, y_np = syn.make_windows(B, T, C, seed=syn.INPUT_SEED + rank)
        self.x = torch.from_numpy(self.x_np).to(dev)
        self.y = torch.from_numpy(y_np).to(dev)
        self.integ = LSTMODEIntegration(self.model, CognitiveStateODE(), 0.5)
        self.api_chunks = 3 if api_level else 1
        if api_level:
            self.x_api = np.concatenate([self.x_np] * self.api_chunks, 0)
        self.gather_buf = torch.empty((world * B, 2), device=dev) if world > 1 else None
        if mode == "train":
            # the reference's training-step body (04_lstm_model.py:482-512): fwd -> weighted CE -> bwd ->
            # [data-parallel: all-reduce of the flat gradient] -> clip 1.0 + AdamW, all inside the timed step
            from lstm_ode_bci_amd.training import FusedAdamW, WeightedCrossEntropy
            self.criterion = WeightedCrossEntropy(torch.tensor([1.0, 1.0], device=dev)).to(dev)
            self.opt = FusedAdamW(self.model.parameters(), lr=3e-4, weight_decay=1e-4, model=self.model)

    def _switches(self):
        """Context of one leg: storage switches of ops.py and the test-only variant table, restored afterwards."""
        import contextlib
        from lstm_ode_bci_amd import _lib, ops
        leg = self

        @contextlib.contextmanager
        def cm():
            old = (ops.C_BF16, ops.DY_BF16_CARRY, ops.LN_X_BF16, ops.DPRE_BF16)
            if leg.strict_storage:
                ops.C_BF16 = ops.DY_BF16_CARRY = ops.LN_X_BF16 = ops.DPRE_BF16 = False
            try:
                if leg.exact_fp32:
                    with _lib.variant(F32_SPLIT=0):
                        yield
                else:
                    yield
            finally:
                ops.C_BF16, ops.DY_BF16_CARRY, ops.LN_X_BF16, ops.DPRE_BF16 = old
        return cm()

    def step(self):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(self.precision == "mixed")):
            self._step()

    def _step(self):
        from lstm_ode_bci_amd.sharding import all_reduce_flat_grad_
        m = self.model
        if self.mode == "train":
            m.train()
            self.opt.zero_grad()
            loss = self.criterion(m(self.x), self.y)
            loss.backward()
            gscale = 1.0
            if self.world > 1:
                _, gscale = all_reduce_flat_grad_(self.opt.flat_grad)        # one 4.55 MB message
            self.opt.step(clip_grad_norm=1.0, grad_scale=gscale)
        elif self.mode == "fwd":
            m.eval()
            with torch.no_grad():
                logits = m(self.x)
                if self.world > 1:
                    self.dist.all_gather_into_tensor(self.gather_buf, logits.contiguous())
        elif self.api_level:
            # the reference's contract: numpy in -> numpy out (06:346, 406), PCIe both ways inside the timed step; the
            # host array holds API_CHUNKS device passes, so the upload of pass i+1 overlaps the kernels of pass i
            self.integ.predict_batch(self.x_api, forecast_steps=self.forecast_steps, batch_size=self.B, show_progress=False,
                                     use_amp=self.precision == "mixed")
        else:
            traj, probs, pred = self.integ.predict_batch_device(self.x, forecast_steps=self.forecast_steps, batch_size=self.B,
                                                                use_amp=self.precision == "mixed")
            if self.world > 1:
                self.dist.all_gather_into_tensor(self.gather_buf, probs.contiguous())

    def run(self, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        with self._switches():
            return self._run(steps, warmup)

    def _run(self, steps, warmup):
        for _ in range(warmup):
            self.step()
        if self.world > 1:
            self.dist.barrier()
        torch.cuda.synchronize()
        # a full Python garbage collection that has become due (imports and set-up leave ~10^6 tracked objects) must not
        # land inside a timed region a few hundred milliseconds long: measured 30-40 ms, ONCE per process, in whichever leg ran
        # first (round 4: a forward-only leg of 12 steps read 5.9 instead of 2.9 ms per step).  Collect now; the collector stays on
        gc.collect()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        dt = time.perf_counter() - t0
        if self.world > 1:
            tt = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    def describe(self, steps, warmup, dt, with_roofline=True):
        with self._switches():
            return self._describe(steps, warmup, dt, with_roofline)

    def _describe(self, steps, warmup, dt, with_roofline=True):
        mode, precision, B, H, world = self.mode, self.precision, self.B, self.H, self.world
        value = world * B * steps * self.api_chunks / dt
        mf = fp32_mfma_kind(H) if precision == "fp32" else "bf16"
        if precision == "mixed":
            storage = ("; storage: bf16 P / saved gates / dP / inter-layer activations" +
                       ("; saved cell states, gradient carries between layers and the last layer's output fp32 (strict)"
                        if self.strict_storage else
                        ", and ALSO bf16 saved cell states, bf16 gradient carries between layers / LayerNorms and a bf16-only "
                        "last-layer output (the carried state, accumulation and all parameter gradients stay fp32)")
                       if mode == "train" else "; storage: bf16 P and inter-layer activations")
        else:
            storage = ""
        flop_per_window = gate_flop_fwd(H) * (3 if mode == "train" else 1)
        res = {
            "metric": {"train": "eeg_windows_per_sec_fwd_bwd", "fwd": "eeg_windows_per_sec_fwd",
                       "coupled": "eeg_windows_per_sec_fwd_ode"}[mode],
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps / self.api_chunks * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": f"BiLSTM({L}x{H})+attn {mode}, (T={T},C={C}) windows, B={B}/GPU, " +
                                   (("fp32 (matrix products as two-way fp16 splits on the 16-bit MFMA pipe: 22-bit products, "
                                     "fp32 accumulate / state; parity <= 1e-5)" if mf == "f16x2" else "fp32 exact-MFMA")
                                    if precision == "fp32" else
                                    "bf16 gate GEMMs + fp32 recurrence/accumulate (autocast)" + storage)
                                   + (f", RK4 ODE {self.forecast_steps} points" if mode == "coupled" else "")
                                   + (f", numpy in -> numpy out through predict_batch ({self.api_chunks} x {B} windows per call, PCIe "
                                      "inside the step)" if self.api_level else ""),
                       "batch_per_gpu": B, "global_batch": world * B, "seq_len": T, "channels": C,
                       "hidden": H, "layers": L, "mode": mode, "precision": precision,
                       **({"step": "fwd + weighted CE + bwd + clip 1.0 + AdamW (04_lstm_model.py:482-512)"}
                          if mode == "train" else {}),
                       "collective": ("none" if world == 1 else
                                      ("all_reduce(grads 4.55MB)" if mode == "train" else "all_gather(logits)"))},
            # whole-step gate-GEMM rate against the dense MFMA peak of the arithmetic dtype the GEMMs run in
            "gate_gemm_tflops_effective": value * flop_per_window / 1e12 / world,
            "gate_gemm_mfma_dtype": mf,
            "gate_gemm_frac_of_mfma_peak": value * flop_per_window / 1e12 / MFMA_PEAK_TFLOPS[mf] / world,
        }
        if with_roofline:
            kr = kernel_roofline(self.dev, B, mode, precision, H)
            timing = "isolated"
            if world == 1 and not self.api_level:
                # the judged durations: HIP events around every launch inside real steps (rank-local: no collective
                # may sit inside these extra steps, so N > 1 keeps the isolated timings)
                for k, (sec, n_per_step, nb) in instep_kernel_times(self, kr).items():
                    kr[k]["sec_isolated"], kr[k]["sec"] = kr[k]["sec"], sec
                    kr[k]["per_step"] = n_per_step
                    if nb is not None:
                        kr[k]["bytes"] = nb
                timing = "in-step"
            roof = roofline_of(kr)
            roof["timing"] = timing
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # HBM bytes per launch from rocprofv3 --pmc
            if os.path.exists(tpath):
                try:
                    from lstm_ode_bci_amd import _lib
                    table = json.load(open(tpath))
                    stamps = table.get("_build_id", {})
                    have = _lib.lib().lob_build_id().decode()
                    sfx = f"|{precision}|B{B}" + ("" if H == 128 else f"|H{H}")
                    roof["traffic"] = table.get(roof["kernel"] + sfx)
                    # the table is a separate profiler pass (rocprofv3 --pmc cannot run inside this process): every
                    # entry carries the build id it was measured on; one from another build is reported, but marked
                    if roof["traffic"] is not None:
                        roof["traffic_source"] = "profiles/pmc_traffic.json"
                        roof["traffic_stale"] = stamps.get(roof["kernel"] + sfx) != have
                    # whole step: HBM bytes of ALL kernels of one step from the same PMC passes (tools/pmc_traffic.py),
                    # over this run's measured step time; only when measured on THIS build.  An upper bound: the 2x
                    # FETCH_SIZE correction (validated on wide coalesced reads) is applied to every kernel of the step
                    sb = table.get("step" + sfx) if (mode == "train" and not self.strict_storage) else None
                    if sb and stamps.get("step" + sfx) == have:
                        gbps = sb / (dt / steps) / 1e9
                        roof["step"] = {"bytes": sb, "achieved": gbps, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                                        "bytes_per_window": sb / B, "note": "upper bound (2x read correction on every kernel)"}
                except Exception:
                    pass
            res["roofline"] = roof
        return res

    def free(self):
        for a in ("model", "x", "y", "integ", "opt", "criterion", "gather_buf"):
            if hasattr(self, a):
                delattr(self, a)
        torch.cuda.empty_cache()


def extra_legs(dev, forecast_steps, cpu=True):
    """Short legs over the BASELINE.json configurations the headline does not cover (N = 1 only)."""
    specs = [
        ("configs[2] strict storage: training step, mixed, B=4096, fp32 cell states / gradient carries / last-layer output",
         dict(mode="train", precision="mixed", B=4096, H=128, strict_storage=True), 5, 2),
        ("configs[1] fp32 forward, B=1024", dict(mode="fwd", precision="fp32", B=1024, H=128), 8, 3),
        ("configs[1] fp32 forward, B=1024, exact-fp32 MFMA kernels (LOB_VAR_F32_SPLIT=0)",
         dict(mode="fwd", precision="fp32", B=1024, H=128, exact_fp32=True), 5, 2),
        ("configs[3] coupled LSTM->ODE, B=4096, device-resident", dict(mode="coupled", precision="fp32", B=4096, H=128), 5, 2),
        ("configs[3] coupled LSTM->ODE, B=4096, numpy in -> numpy out (predict_batch)",
         dict(mode="coupled", precision="fp32", B=4096, H=128, api_level=True), 3, 1),
        ("H=256 mixed training step, B=4096 (the reference's checkpoint size, 04:877)",
         dict(mode="train", precision="mixed", B=4096, H=256), 5, 2),
        ("H=256 mixed training step, B=512 (the reference's own training batch and hidden size, 04:866-877)",
         dict(mode="train", precision="mixed", B=512, H=256), 8, 3),
        ("configs[4] per-rank shard: forward, B=8192, fp32", dict(mode="fwd", precision="fp32", B=8192, H=128), 4, 2),
        ("configs[4] per-rank shard: forward, B=8192, mixed", dict(mode="fwd", precision="mixed", B=8192, H=128), 5, 2),
        ("H=256 mixed forward, B=4096 (a real checkpoint under the reference's inference autocast, 06:349)",
         dict(mode="fwd", precision="mixed", B=4096, H=256), 5, 2),
    ]
    out = {}
    for name, kw, steps, warmup in specs:
        t0 = time.perf_counter()
        leg = Leg(dev, forecast_steps=forecast_steps, **kw)
        dt = leg.run(steps, warmup)
        r = leg.describe(steps, warmup, dt, with_roofline=not (kw.get("api_level", False) or kw.get("strict_storage", False)
                                                                 or kw.get("exact_fp32", False)))
        if cpu and kw.get("mode") == "coupled" and not kw.get("api_level", False):
            # the reference's coupled path on this host: LSTM chunk loop + one odeint per window (06:343-401)
            r["cpu_baseline"] = cpu_baseline("coupled", leg.sd, forecast_steps, kw["H"], budget_s=25.0)
            r["speedup_vs_cpu_baseline"] = r["value"] / r["cpu_baseline"]["value"]
        r["leg_wall_s"] = round(time.perf_counter() - t0, 1)
        leg.free()
        out[name] = r
    return out


def compact(res):
    """The ONE line the contract asks for, small enough to survive a log tail: every leg keeps its rate, step time,
    workload and dominant-kernel roofline; the per-kernel tables go to --detail."""
    def roof(r):
        if not r:
            return None
        out = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch",
                                     "flop_per_launch", "sec_per_launch", "timing", "step", "traffic_stale") if r.get(k) is not None or k == "traffic"}
        if "all" in r:      # [ms per launch, launches per step, fraction of HBM peak, fraction of its MFMA peak]
            out["all"] = {k: [v["ms"], v["launches_per_step"], v["frac_of_hbm_peak"], v["frac_of_mfma_peak"]]
                          for k, v in r["all"].items()}
        return out

    def cpu(c):
        if not c:
            return None
        return {k: c[k] for k in ("value", "median", "unit", "cores", "kind", "host_cpus", "cpu_model", "n1_value",
                                  "lstm_windows_per_s", "ode_solves_per_s_1thread", "sample") if k in c}
    out = {k: v for k, v in res.items() if k not in ("roofline", "cpu_baseline", "extra_configs")}
    if res.get("roofline"):
        out["roofline"] = roof(res.get("roofline"))
    if "cpu_baseline" in res:
        out["cpu_baseline"] = cpu(res["cpu_baseline"])
    if "extra_configs" in res:
        legs = {}
        for name, r in res["extra_configs"].items():
            e = {"value": round(r["value"], 1), "ms_per_step": round(r["ms_per_step"], 3), "workload": r["config"]["workload"]}
            if r.get("roofline"):
                e["roofline"] = {k: r["roofline"][k] for k in ("bound", "kernel", "frac", "sec_per_launch") if k in r["roofline"]}
            if r.get("cpu_baseline"):
                e["cpu_baseline"] = cpu(r["cpu_baseline"])
                e["speedup_vs_cpu_baseline"] = r["speedup_vs_cpu_baseline"]
            legs[name] = e
        out["extra_configs"] = legs
        # the same numbers where a key whitelist cannot drop them
        short = {}
        for n, r in res["extra_configs"].items():
            c = r["config"]
            key = f"{c['mode']} {c['precision']} H{c['hidden']} B{c['batch_per_gpu']}"
            if "exact-fp32" in n:
                key += " exact-fp32-mfma"
            if "strict" in n:
                key += " strict-storage"
            if "numpy in" in n:
                key += " numpy-in-out"
            short[key] = round(r["value"])
        out["config"]["also_measured_windows_per_s"] = short
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: run the same command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (one rank per GPU, rendezvous on 127.0.0.1 at a
    free port) as a CHILD process, pass its stdout (rank 0's one JSON line) and stderr through, return its exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")               # torch.distributed.run would set 1 and say so on stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    a = parse()
    H = a.hidden
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (fresh child processes; this process has made no
        # GPU call and never replaces its own image), relay their output and exit with their code
        raise SystemExit(launch_ranks(a.gpus))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} does not match WORLD_SIZE={world} of the launcher")
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

    default_run = a.mode is None and a.precision is None and a.batch == 4096 and H == 128
    mode = a.mode or "train"
    precision = a.precision or ("mixed" if mode == "train" else "fp32")
    leg = Leg(dev, mode, precision, a.batch, H, a.forecast_steps, world, rank, dist, strict_storage=a.strict_storage)
    dt = leg.run(a.steps, a.warmup)

    if rank == 0:
        res = leg.describe(a.steps, a.warmup, dt, with_roofline=not a.no_roofline)
        sd = leg.sd
        leg.free()
        if world == 1 and default_run and not a.no_extra:
            res["extra_configs"] = extra_legs(dev, a.forecast_steps, cpu=not a.no_cpu_baseline)
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(mode, sd, a.forecast_steps, H, budget_s=60.0)
            res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
        if a.detail:
            with open(a.detail, "w") as f:
                json.dump(res, f, indent=1)
        print(json.dumps(compact(res)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
