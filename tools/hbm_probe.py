#!/usr/bin/env python3
"""Achievable HBM bandwidth on this GPU for simple streaming patterns (torch elementwise kernels, 1 GiB tensors):
copy (1R:1W), add (2R:1W), sum (read only), fill (write only).  Context for the roofline fractions in DESIGN.md."""
import torch

dev = torch.device("cuda:0")
n = 256 * 1024 * 1024
a = torch.randn(n, device=dev)
b = torch.randn(n, device=dev)
c = torch.empty(n, device=dev)


def t(fn, it=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e-3


for name, fn, byts in (("copy 1R:1W", lambda: c.copy_(a), 8 * n), ("add 2R:1W", lambda: torch.add(a, b, out=c), 12 * n),
                       ("sum read-only", lambda: a.sum(), 4 * n), ("fill write-only", lambda: c.fill_(1.0), 4 * n)):
    s = t(fn)
    print(f"{name:16s} {byts / s / 1e9:8.0f} GB/s  ({s * 1e3:.3f} ms)")
