#!/usr/bin/env python3
"""Same-process A/B of the H = 256 step's matrix-bound GEMMs at B = 4096: ping-pong kernels (csrc/gemm_pp.hip) against
the tiled LDS-DMA / weight-stationary twins, interleaved rounds, HIP-event times (median / min)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import _lib, ops

dev = torch.device("cuda:0")
T, H, D, B = 256, 256, 2, int(os.environ.get("KB_B", "4096"))
Bp = ops.ceil32(B)
rows = T * Bp
g = torch.Generator(device=dev).manual_seed(1)


def rnd(shape, scale=1.0, dtype=torch.float32):
    return (torch.randn(shape, generator=g, device=dev) * scale).to(dtype)


def ab(name, fn, flop, variants, rounds=6):
    ts = {k: [] for k in variants}
    for k, kw in variants.items():
        with _lib.variant(**kw):
            fn(); fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, kw in variants.items():
            with _lib.variant(**kw):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); fn(); e.record()
                torch.cuda.synchronize()
                ts[k].append(s.elapsed_time(e))
    for k in variants:
        med, mn = float(np.median(ts[k])), float(np.min(ts[k]))
        print(f"{name:34s} {k:10s} {med:7.3f} ms (min {mn:7.3f})  {flop / med / 1e9:7.0f} TFLOP/s", flush=True)


V = {"pp 32x32x16": dict(GEMM_PP=7), "pp 16x16x32": dict(GEMM_PP=7 | 512), "pp ring": dict(GEMM_PP=15), "twin": dict(GEMM_PP=0)}
want = set(sys.argv[1:])
dP = rnd((rows, D * 4 * H), 1e-2, torch.bfloat16)
if not want or "dx" in want:
    for N in (512, 256):
        wt = rnd((N, D * 4 * H), 0.05, torch.bfloat16)
        ab(f"dX K=2048 N={N}", lambda: ops.gemm_nt(dP, wt, mixed=True, out_bf16=True, drop_p=0.4, seed=3),
           2.0 * rows * N * D * 4 * H, V)
if not want or "gate" in want:
    bias = rnd((D * 4 * H,), 0.1)
    for K in (512, 256):
        x = rnd((rows, K), 1.0, torch.bfloat16)
        w = rnd((D * 4 * H, K), 0.05, torch.bfloat16)
        ab(f"gate K={K}", lambda: ops.gate_gemm_x(x, w, bias, T, Bp, H, D, True, mixed=True), 2.0 * rows * K * D * 4 * H, V)
if not want or "dw" in want:
    for K in (512, 256):
        x = rnd((rows, K), 1.0, torch.bfloat16)
        out = torch.zeros((D * 4 * H, K), device=dev)
        ab(f"dW_ih X width {K}", lambda: ops.gemm_tn(dP, x, out), 2.0 * rows * K * D * 4 * H, V)
    y = rnd((rows, D * H), 1.0, torch.bfloat16)
    out = torch.zeros((4 * H, H), device=dev)
    ab("dW_hh one direction", lambda: ops.gemm_tn(dP[Bp:, :4 * H], y[:(T - 1) * Bp, :H], out), 2.0 * rows * H * 4 * H, V)

if "abl" in want:     # needs a -DLOB_PP_DIAG build of gemm_pp.hip (ABL_SRC=gemm_pp ABL_DEF=LOB_PP_DIAG tools/h256_ablate.sh 1; LOB_LIB_PATH=ab/liblob_abl1.so)
    wt = rnd((512, D * 4 * H), 0.05, torch.bfloat16)
    VA = {"full": dict(GEMM_PP=7)}
    for a, nm in ((1, "no DMA"), (2, "no ds_read"), (4, "no MFMA"), (3, "no DMA+read"), (5, "no DMA+MFMA"), (6, "no read+MFMA"),
                  (7, "barriers only")):
        VA[nm] = dict(GEMM_PP=7 | (a << 4))
    ab("dX N=512 ablation", lambda: ops.gemm_nt(dP, wt, mixed=True, out_bf16=True), 2.0 * rows * 512 * D * 4 * H, VA, rounds=4)

if "prio" in want:
    wt = rnd((512, D * 4 * H), 0.05, torch.bfloat16)
    VP = {"mfma prio1": dict(GEMM_PP=7), "no setprio": dict(GEMM_PP=7 | (1 << 7)), "load prio1": dict(GEMM_PP=7 | (2 << 7))}
    ab("dX N=512 priority", lambda: ops.gemm_nt(dP, wt, mixed=True, out_bf16=True), 2.0 * rows * 512 * D * 4 * H, VP, rounds=6)

if "dma" in want:     # -DLOB_PP_DIAG build as well
    for N in (512, 256):
        wt = rnd((N, D * 4 * H), 0.05, torch.bfloat16)
        VD = {"full kernel": dict(GEMM_PP=7), "DMA only, 8 in flight/wave": dict(GEMM_PP=7 | 1024),
              "DMA only, 24 in flight/wave": dict(GEMM_PP=7 | 1024 | 2048)}
        ab(f"dX N={N} operand DMA", lambda: ops.gemm_nt(dP, wt, mixed=True, out_bf16=True), 2.0 * rows * N * D * 4 * H, VD, rounds=5)

if "fused" in want or not want:
    Y = rnd((rows, D * H), 1.0, torch.bfloat16)
    for nx in (512, 256):
        X = rnd((rows, nx), 1.0, torch.bfloat16)
        o1 = torch.zeros((D * 4 * H, nx), device=dev)
        o2 = torch.zeros((D, 4 * H, H), device=dev)

        def separate():
            ops.gemm_tn(dP, X, o1)
            ops.gemm_tn(dP[Bp:, :4 * H], Y[:(T - 1) * Bp, :H], o2[0])
            ops.gemm_tn(dP[:(T - 1) * Bp, 4 * H:], Y[Bp:, H:], o2[1])
        fl = 2.0 * rows * D * 4 * H * (nx + H)
        ab(f"layer dW nx={nx}: fused", lambda: ops.lstm_dw(dP, X, Y, T, Bp, H, D, out=(o1, o2)), fl, {"fused 1 launch": dict(GEMM_PP=5)})
        ab(f"layer dW nx={nx}: separate", separate, fl, {"3 launches pp": dict(GEMM_PP=5), "3 launches twin": dict(GEMM_PP=0)})

if "h128" in want:      # the H = 128 step's dX shapes (K = 1024): k-split weight-stationary kernel against the ping-pong kernels
    dP1 = rnd((rows, 1024), 1e-2, torch.bfloat16)
    for N in (256, 128):
        wt = rnd((N, 1024), 0.05, torch.bfloat16)
        VH = {"k-split": dict(DX_KSPLIT=1), "pp 32x32x16": dict(DX_KSPLIT=0, GEMM_PP=7), "pp 16x16x32": dict(DX_KSPLIT=0, GEMM_PP=7 | 512),
              "tiled": dict(DX_KSPLIT=0, GEMM_PP=0)}
        ab(f"H=128 dX K=1024 N={N}", lambda: ops.gemm_nt(dP1, wt, mixed=True, out_bf16=True, drop_p=0.4, seed=3),
           2.0 * rows * N * 1024, VH)
