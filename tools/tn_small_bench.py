#!/usr/bin/env python3
"""The two small TN GEMMs of the mixed step (score layer dW1 = dU^T v, input projection dW = dpre^T xb) alone."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import ops
dev = torch.device("cuda:0")
rows = 256 * 4096
g = torch.Generator(device=dev).manual_seed(1)
def rnd(shape, dtype=torch.bfloat16): return (torch.randn(shape, generator=g, device=dev) * 0.1).to(dtype)
def timeit(fn, n=20):
    for _ in range(3): fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2]
for M, N in ((128, 256), (128, 64), (256, 512), (256, 64)):
    a, b = rnd((rows, M)), rnd((rows, N))
    out = torch.zeros((M, N), device=dev)
    ms = timeit(lambda: ops.gemm_tn(a, b, out, mixed=True))
    byts = rows * (M + N) * 2
    print(f"TN {M}x{N} over {rows} rows: {ms:7.3f} ms  {byts / ms / 1e9:5.2f} TB/s", flush=True)
