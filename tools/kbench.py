#!/usr/bin/env python3
"""Per-kernel timing on one GPU (HIP events, B=4096 shapes of the bench workload).
    python tools/kbench.py [names...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import ops  # noqa: E402

T, H, D, B = 256, 128, 2, int(os.environ.get("KB_B", "4096"))
Bp = ops.ceil32(B)
rows, N = T * Bp, D * 4 * H
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)


def timeit(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = [s.elapsed_time(e) for s, e in evs]
    return float(np.median(ts)), float(np.min(ts))


def report(name, ms, flop, byts):
    print(f"{name:34s} {ms[0]:8.3f} ms (min {ms[1]:7.3f})  {flop / ms[0] / 1e9:8.1f} TFLOP/s  {byts / ms[0] / 1e6:8.0f} GB/s",
          flush=True)


want = set(sys.argv[1:])
def on(n):
    return not want or any(w in n for w in want)


for K in (128, 256):
    x = torch.randn((rows, K), generator=g).to(dev)
    wih = (torch.rand((N, K), generator=g) * 0.17 - 0.085).to(dev)
    bias = torch.zeros(N, device=dev)
    for mixed in (False, True):
        nm = f"gate_gemm K={K} {'bf16' if mixed else 'f32'}"
        if on(nm):
            report(nm, timeit(lambda: ops.gate_gemm_x(x, wih, bias, T, Bp, H, D, True, mixed=mixed)),
                   2.0 * rows * N * K, 4.0 * rows * (K + N))
whh = (torch.rand((D, 4 * H, H), generator=g) * 0.17 - 0.085).to(dev)
P = ops.gate_gemm_x(x, wih, bias, T, Bp, H, D, True)
for mixed in (False, True):
    tag = 'bf16' if mixed else 'f32'
    if on(f"rec_fwd {tag}"):
        report(f"rec_fwd {tag}", timeit(lambda: ops.lstm_rec_fwd(P, whh, T, Bp, H, D, False, mixed=mixed), 3),
               2.0 * rows * N * H, 4.0 * rows * (N + D * H))
    if on(f"rec_fwd_save {tag}") or on(f"rec_bwd {tag}") or on("tn") or on("dX"):
        Pk = P.clone()
        Y, Cs, _, _ = ops.lstm_rec_fwd(Pk, whh, T, Bp, H, D, True, mixed=mixed)
        if on(f"rec_fwd_save {tag}"):
            report(f"rec_fwd_save {tag} (on gates)", timeit(lambda: ops.lstm_rec_fwd(Pk, whh, T, Bp, H, D, True, mixed=mixed), 3),
                   2.0 * rows * N * H, 4.0 * rows * (2 * N + 2 * D * H))
            Pk.copy_(P); Y, Cs, _, _ = ops.lstm_rec_fwd(Pk, whh, T, Bp, H, D, True, mixed=mixed)
        dY = (torch.randn((rows, D * H), generator=g) * 1e-3).to(dev)
        if on(f"rec_bwd {tag}"):
            report(f"rec_bwd {tag}", timeit(lambda: ops.lstm_rec_bwd(Pk, Cs, whh, dY, T, Bp, H, D, dp_bf16=mixed), 3),
                   2.0 * rows * N * H, 4.0 * rows * (N + 3 * D * H) + (2.0 if mixed else 4.0) * rows * N)
        dP, _ = ops.lstm_rec_bwd(Pk, Cs, whh, dY, T, Bp, H, D, dp_bf16=mixed)
        K = D * H
        dw = torch.zeros((N, K), device=dev)
        eb = 2.0 if mixed else 4.0
        if on(f"tn dWih {tag}"):
            report(f"tn dWih {tag}", timeit(lambda: ops.gemm_tn(dP, x, dw, mixed=mixed)), 2.0 * rows * N * K,
                   eb * rows * N + 4.0 * rows * K)
        if on(f"tn dWhh {tag}"):
            dwh = torch.zeros((4 * H, H), device=dev)
            report(f"tn dWhh {tag}", timeit(lambda: ops.gemm_tn(dP[Bp:, :4 * H], Y[:rows - Bp, :H], dwh, mixed=mixed)),
                   2.0 * rows * 4 * H * H, eb * rows * 4 * H + 4.0 * rows * H)
        if on(f"dX {tag}"):
            wt = wih.t().contiguous()
            report(f"dX {tag}", timeit(lambda: ops.gemm_nt(dP, wt, mixed=mixed)), 2.0 * rows * N * K,
                   eb * rows * N + 4.0 * rows * K)
        del Pk, Y, Cs, dP

# ---- LDS-DMA NT GEMM (bf16 x bf16) ----------------------------------------------------------
if on("dma"):
    for K in (128, 256):
        xb = torch.randn((rows, K), generator=g).to(dev).to(torch.bfloat16)
        wb = (torch.rand((N, K), generator=g) * 0.17 - 0.085).to(dev)
        bias = torch.zeros(N, device=dev)
        for wdt, nm in ((torch.float32, "regstage"), (torch.bfloat16, "dma")):
            report(f"gate_gemm K={K} bf16X {nm}", timeit(lambda: ops.gate_gemm_x(xb, wb.to(wdt), bias, T, Bp, H, D, True, mixed=True)),
                   2.0 * rows * N * K, rows * (2.0 * K + 2.0 * N))
    dPb = torch.randn((rows, N), generator=g).to(dev).to(torch.bfloat16)
    for Kout in (128, 256):
        wt = (torch.rand((Kout, N), generator=g) * 0.1).to(dev)
        for wdt, nm in ((torch.float32, "regstage"), (torch.bfloat16, "dma")):
            report(f"dX N={Kout} {nm}", timeit(lambda: ops.gemm_nt(dPb, wt.to(wdt), mixed=True)), 2.0 * rows * N * Kout,
                   rows * (2.0 * N + 4.0 * Kout))

if on("ws"):
    from lstm_ode_bci_amd import _lib
    for K in (128, 256):
        xb = torch.randn((rows, K), generator=g).to(dev).to(torch.bfloat16)
        wb = (torch.rand((N, K), generator=g) * 0.17 - 0.085).to(dev).to(torch.bfloat16)
        bias = torch.zeros(N, device=dev)
        for v, nm in ((0, "tiled"), (1, "weight-stationary")):
            with _lib.variant(GATE_WS=v):
                report(f"gate_gemm K={K} {nm}", timeit(lambda: ops.gate_gemm_x(xb, wb, bias, T, Bp, H, D, True, mixed=True), 8),
                       2.0 * rows * N * K, rows * (2.0 * K + 2.0 * N))

if on("split"):
    from lstm_ode_bci_amd import _lib
    for K in (128, 256):
        xf = torch.randn((rows, K), generator=g).to(dev)
        wf = (torch.rand((N, K), generator=g) * 0.17 - 0.085).to(dev)
        bias = torch.zeros(N, device=dev)
        for v, nm in ((0, "exact fp32 MFMA"), (1, "fp16x2 split")):
            with _lib.variant(F32_SPLIT=v):
                report(f"gate_gemm f32 K={K} {nm}", timeit(lambda: ops.gate_gemm_x(xf, wf, bias, T, Bp, H, D, True), 4),
                       2.0 * rows * N * K, rows * (4.0 * K + 4.0 * N))
    whh_ = (torch.rand((D, 4 * H, H), generator=g) * 0.17 - 0.085).to(dev)
    with _lib.variant(F32_SPLIT=0):
        Pf = ops.gate_gemm_x(xf, wf, bias, T, Bp, H, D, True)
    for v, nm in ((0, "exact fp32 MFMA"), (1, "fp16x2 split")):
        with _lib.variant(F32_SPLIT=v):
            report(f"rec_fwd f32 {nm}", timeit(lambda: ops.lstm_rec_fwd(Pf, whh_, T, Bp, H, D, False), 3),
                   2.0 * rows * N * H, 4.0 * rows * (N + D * H))

if on("tndma"):
    dPb = torch.randn((rows, N), generator=g).to(dev).to(torch.bfloat16)
    xb = torch.randn((rows, 256), generator=g).to(dev).to(torch.bfloat16)
    dw = torch.zeros((N, 256), device=dev)
    report("tn dWih bf16xbf16", timeit(lambda: ops.gemm_tn(dPb, xb, dw)), 2.0 * rows * N * 256, 2.0 * rows * (N + 256))
    dwh = torch.zeros((512, 128), device=dev)
    report("tn dWhh bf16xbf16", timeit(lambda: ops.gemm_tn(dPb[Bp:, :512], xb[:rows - Bp, :128], dwh)),
           2.0 * rows * 512 * 128, 2.0 * rows * (512 + 128))

if on("f32dma"):
    for K in (128, 256):
        x = torch.randn((rows, K), generator=g).to(dev)
        wih = (torch.rand((N, K), generator=g) * 0.17 - 0.085).to(dev)
        bias = torch.zeros(N, device=dev)
        report(f"gate_gemm K={K} f32", timeit(lambda: ops.gate_gemm_x(x, wih, bias, T, Bp, H, D, True)),
               2.0 * rows * N * K, 4.0 * rows * (K + N))
    dPf = torch.randn((rows, N), generator=g).to(dev)
    wt = (torch.rand((256, N), generator=g) * 0.1).to(dev)
    report("dX f32 N=256", timeit(lambda: ops.gemm_nt(dPf, wt)), 2.0 * rows * N * 256, 4.0 * rows * (N + 256))
