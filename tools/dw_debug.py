#!/usr/bin/env python3
"""Debug helper: per-block error map of the fused dW kernel against fp64 products."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lstm_ode_bci_amd import ops

dev = torch.device("cuda:0")
H = 128
for (T, Bp, nx, D) in [(5, 32, 256, 2), (3, 64, 128, 2), (40, 96, 256, 2)]:
    rng = np.random.default_rng(T * Bp + nx)
    bf = torch.bfloat16
    dP = torch.from_numpy(rng.standard_normal((T * Bp, D * 4 * H), dtype=np.float32)).to(dev).to(bf)
    X = torch.from_numpy(rng.standard_normal((T * Bp, nx), dtype=np.float32)).to(dev).to(bf)
    Y = torch.from_numpy(rng.standard_normal((T * Bp, D * H), dtype=np.float32)).to(dev).to(bf)
    dwih, dwhh = ops.lstm_dw(dP, X, Y, T, Bp, H, D)
    p64, x64, y64 = dP.double(), X.double(), Y.double()
    e = (dwih.double() - p64.T @ x64).abs()
    print(T, Bp, nx, "dwih blocks bad:", [(i, j) for i in range(e.shape[0] // 32) for j in range(e.shape[1] // 32)
                                          if e[32 * i:32 * i + 32, 32 * j:32 * j + 32].max() > 1e-2][:40])
    for d in range(D):
        a = p64[:, d * 512:(d + 1) * 512]
        y = y64[:, d * H:(d + 1) * H]
        ref = a[Bp:].T @ y[:(T - 1) * Bp] if d == 0 else a[:(T - 1) * Bp].T @ y[Bp:]
        e = (dwhh[d].double() - ref).abs()
        bad = [(i, j) for i in range(16) for j in range(4) if e[32 * i:32 * i + 32, 32 * j:32 * j + 32].max() > 1e-2]
        print("   dwhh", d, "bad blocks", bad)
        if bad:
            i, j = bad[0]
            blk = e[32 * i:32 * i + 32, 32 * j:32 * j + 32]
            print("   rows bad:", (blk.max(1).values > 1e-2).nonzero().flatten().tolist(), "cols bad:",
                  (blk.max(0).values > 1e-2).nonzero().flatten().tolist())
            # is the wrong value equal to the product WITHOUT the exclusion / with another shift?
            alt = a.T @ y
            print("   matches unshifted product:", (dwhh[d].double() - alt).abs()[32 * i:32 * i + 32, 32 * j:32 * j + 32].max().item())

# ---- timing at the step's size
T, Bp, D = 256, 4096, 2
for nx in (256, 128):
    bf = torch.bfloat16
    dP = (torch.randn((T * Bp, D * 4 * H), device=dev) * 0.1).to(bf)
    X = torch.randn((T * Bp, nx), device=dev).to(bf)
    Y = torch.randn((T * Bp, D * H), device=dev).to(bf)
    for _ in range(3):
        ops.lstm_dw(dP, X, Y, T, Bp, H, D)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.lstm_dw(dP, X, Y, T, Bp, H, D)
    e1.record()
    torch.cuda.synchronize()
    print("lstm_dw nx=%d: %.3f ms" % (nx, e0.elapsed_time(e1) / 10))
