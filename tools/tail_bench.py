#!/usr/bin/env python3
"""The fused tail kernels of the mixed path alone (post-LSTM LayerNorm + score layer forward, lob_attn_scores_bf16; dV GEMM +
LayerNorm backward, lob_attn_ln_bwd_bf16) at the bench's shapes, against the HBM time of their algorithmic bytes.

    python tools/tail_bench.py [B=4096] [T=256]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import ops                      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
Bp = ops.ceil32(B)
rows = T * Bp


def rnd(shape, scale=1.0, dtype=torch.float32):
    return (torch.randn(shape, generator=g, device=dev) * scale).to(dtype)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ts[n // 2]


for H in (128, 256):
    W, W2 = 2 * H, H
    y = rnd((rows, W), 1.0, torch.bfloat16)
    gamma, beta = 1 + rnd((W,), 0.1), rnd((W,), 0.1)
    w1 = rnd((W2, W), 0.05, torch.bfloat16)
    b1, w2, b2 = rnd((W2,), 0.1), rnd((W2,), 0.1), rnd((1,), 0.1)
    for save in (True, False):
        ms = timeit(lambda: ops.attn_scores(y, gamma, beta, w1, b1, w2, b2, T, B, Bp, H, 2, save=save))
        byts = rows * (W * 2 * 2 + (W2 * 4 if save else 0) + 4)
        print(f"H={H} attn_scores save={int(save)}: {ms:7.3f} ms  {byts / ms / 1e9:6.2f} TB/s ({byts / 1e9:.2f} GB)", flush=True)
    dU = rnd((rows, W2), 1e-3, torch.bfloat16)
    w1t = w1.t().contiguous()
    attn = torch.softmax(rnd((B, T)), dim=1)
    dctx = rnd((B, W), 1e-3)
    ms = timeit(lambda: ops.attn_ln_bwd(y, gamma, beta, dU, w1t, attn, dctx, T, B, Bp, H, 2))
    byts = rows * (W * 2 * 2 + W2 * 2)
    print(f"H={H} attn_ln_bwd         : {ms:7.3f} ms  {byts / ms / 1e9:6.2f} TB/s ({byts / 1e9:.2f} GB)", flush=True)
