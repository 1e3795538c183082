#!/usr/bin/env python3
"""Micro-benchmark of the row-wise kernels at the training-step shapes (B=4096, T=256)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import ops

dev = torch.device("cuda:0")
T, B = 256, 4096
Bp = B


def timeit(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in evs]))


for W in (128, 256):
    x = torch.randn((T * B, W), device=dev)
    g = torch.ones(W, device=dev); b = torch.zeros(W, device=dev)
    dy = torch.randn((T * Bp, W), device=dev)
    for name, kw in (("norm", dict()), ("norm+gelu+drop", dict(act=ops.ACT_GELU, drop_p=0.2, seed=5)),
                     ("norm+remap", dict(remap=(T, B, Bp))), ("norm+gelu+drop+remap+bf16", dict(act=ops.ACT_GELU, drop_p=0.2, seed=5, remap=(T, B, Bp), out_bf16=True))):
        ms = timeit(lambda: ops.layernorm_act(x, g, b, **kw))
        ob = 2 if kw.get("out_bf16") else 4
        print(f"LN fwd W={W} {name:28s} {ms:7.3f} ms  {T*B*W*(4+ob)/ms/1e6:7.0f} GB/s")
    ms = timeit(lambda: ops.layernorm_act(x, None, None))
    print(f"LN fwd W={W} {'identity (copy)':28s} {ms:7.3f} ms  {T*B*W*8/ms/1e6:7.0f} GB/s")
    for name, kw in (("norm", dict()), ("norm+gelu+drop+remap", dict(act=ops.ACT_GELU, drop_p=0.2, seed=5, remap=(T, B, Bp)))):
        ms = timeit(lambda: ops.layernorm_act_bwd(x, g, b, dy, **kw))
        print(f"LN bwd W={W} {name:28s} {ms:7.3f} ms  {T*B*W*12/ms/1e6:7.0f} GB/s")
    ms = timeit(lambda: ops.layernorm_act_bwd(x, None, None, dy))
    print(f"LN bwd W={W} {'identity':28s} {ms:7.3f} ms  {T*B*W*12/ms/1e6:7.0f} GB/s")
    ms = timeit(lambda: torch.add(x, dy, out=dy))
    print(f"torch add W={W} (2R:1W)              {ms:7.3f} ms  {T*B*W*12/ms/1e6:7.0f} GB/s")

# mixed path, width 256, every stream bf16 (LOB_X_BF16): 32 lanes per row against 64 (LOB_VAR_LN_LPR = 64)
from lstm_ode_bci_amd import _lib
W = 256
x16 = torch.randn((T * B, W), device=dev).to(torch.bfloat16)
dy16 = (torch.randn((T * B, W), device=dev) * 1e-3).to(torch.bfloat16)
g = torch.ones(W, device=dev); b = torch.zeros(W, device=dev)
for lpr in (16, 64):
    with _lib.variant(LN_LPR=lpr):
        msf = timeit(lambda: ops.layernorm_act(x16, g, b, out_bf16=True))
        msb = timeit(lambda: ops.layernorm_act_bwd(x16, g, b, dy16, dx_bf16=True))
    print(f"LN W=256 bf16 streams, LN_LPR={lpr}: fwd {msf:.3f} ms ({T*B*W*4/msf/1e6:.0f} GB/s), bwd {msb:.3f} ms ({T*B*W*6/msb/1e6:.0f} GB/s)")
