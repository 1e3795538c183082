#!/usr/bin/env python3
"""Input projection of the mixed path alone (B = 4096, T = 256, C = 61): column-decomposed kernel, the first fused kernel
(one wave per 32-row tile; H = 128 only) and the unfused three-kernel sequence, training (saving) and inference."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import ops

dev = torch.device("cuda:0")
B, T, C = 4096, 256, 61
g = torch.Generator(device=dev).manual_seed(1)
x2d = torch.randn((B * T, C), generator=g, device=dev)


def timeit(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2]


for H in (128, 256):
    w = torch.randn((H, C), generator=g, device=dev) * 0.2
    b = torch.randn((H,), generator=g, device=dev) * 0.1
    gam = torch.rand((H,), generator=g, device=dev) + 0.5
    bet = torch.randn((H,), generator=g, device=dev) * 0.1
    Bp = B
    for save in (True, False):
        kw = dict(act=ops.ACT_GELU, drop_p=0.3, seed=5, save=save)
        t_new = timeit(lambda: ops.input_proj_ln(x2d, w, b, gam, bet, B, T, Bp, H, colwave=True, **kw))
        t_old = timeit(lambda: ops.input_proj_ln(x2d, w, b, gam, bet, B, T, Bp, H, **kw)) if H == 128 else float("nan")

        def unfused():
            xb = ops.pad_cast_bf16(x2d, 64)
            wpad = torch.zeros((H, 64), device=dev)
            wpad[:, :C] = w
            pre = ops.gemm_nt(xb, wpad, b, mixed=True)
            return ops.layernorm_act(pre, gam, bet, act=ops.ACT_GELU, remap=(T, B, Bp), drop_p=0.3, seed=5, out_bf16=True)
        t_un = timeit(unfused)
        print(f"H={H} save={int(save)}: column-decomposed {t_new:.3f} ms | wave-per-tile {t_old:.3f} | unfused sequence {t_un:.3f}", flush=True)
