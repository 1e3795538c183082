#!/usr/bin/env python3
"""Does an H2D copy from page-locked memory overlap the recurrent kernels, and at what rate?  (predict_batch uploads chunk
k+1 on a side stream while chunk k is in the LSTM kernels.)  Prints copy and kernel durations alone and overlapped, for
several piece sizes."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import EnhancedLSTMModel, synthetic as syn

dev = torch.device("cuda:0")
B, T, C, H = 4096, 256, 61, 128
sd = syn.make_state_dict(C, H, 3, 2, True)
m = EnhancedLSTMModel(C, H, 3, 2, 0.4, True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.to(dev).eval()
x = torch.randn(B, T, C, device=dev)
stage = torch.empty((B, T, C), dtype=torch.float32).pin_memory()
dst = torch.empty((B, T, C), device=dev)
side = torch.cuda.Stream()
mb = stage.numel() * 4 / 1e6


def ev():
    return torch.cuda.Event(enable_timing=True)


def run(copy, kernels, pieces=1):
    torch.cuda.synchronize()
    k0, k1, c0, c1 = ev(), ev(), ev(), ev()
    main = torch.cuda.current_stream()
    k0.record(main)
    if kernels:
        with torch.no_grad():
            m(x)
    k1.record(main)
    if copy:
        with torch.cuda.stream(side):
            c0.record(side)
            n = B // pieces
            for p in range(pieces):
                dst[p * n:(p + 1) * n].copy_(stage[p * n:(p + 1) * n], non_blocking=True)
            c1.record(side)
    torch.cuda.synchronize()
    return (k0.elapsed_time(k1) if kernels else 0.0), (c0.elapsed_time(c1) if copy else 0.0), (k0.elapsed_time(c1) if copy else 0.0)


for _ in range(2):
    run(True, True)
print(f"{mb:.0f} MB page-locked -> device; fp32 forward of {B} windows")
print("kernels alone: %.2f ms" % run(False, True)[0])
for pieces in (1, 4, 16):
    c = run(True, False, pieces)[1]
    print(f"copy alone, {pieces:2d} piece(s): {c:.2f} ms = {mb / c:.1f} GB/s")
for pieces in (1, 4, 16):
    k, c, end = run(True, True, pieces)
    print(f"overlapped, {pieces:2d} piece(s): kernels {k:.2f} ms, copy {c:.2f} ms = {mb / c:.1f} GB/s, copy finished {end:.2f} ms after the kernels started")
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"))
