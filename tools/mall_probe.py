#!/usr/bin/env python3
"""Does a consumer kernel read a producer's fresh output out of the 256-MiB Infinity Cache?  Chains of torch streaming
kernels over a working set of S bytes: y = x * a (reads S, writes S) then z = y * b (reads the fresh y, writes S) ...
The achieved bytes/s of the chain as a function of S says how much of a pass a time-chunked pipeline could keep on-die
(the P / G / dP streams of the training step)."""
import sys
import torch

dev = torch.device("cuda:0")


def chain(nbytes, links=8, reps=5):
    n = nbytes // 2
    bufs = [torch.randn(n, device=dev).to(torch.bfloat16) for _ in range(3)]
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(links):
            torch.mul(bufs[i % 3], 1.0001, out=bufs[(i + 1) % 3])
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e-3)
    return 2.0 * nbytes * links / best / 1e12


for mb in (16, 32, 64, 96, 128, 192, 256, 512, 1024, 2048):
    print(f"working set {mb:5d} MB per buffer: {chain(mb << 20):6.2f} TB/s (read + write)", flush=True)
