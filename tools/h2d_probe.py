#!/usr/bin/env python3
"""Host side of predict_batch's upload: how fast the page-locked staging copy (numpy -> pinned) and the H2D DMA run on
this box, by thread count."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
n, shape = 4096, (256, 61)
x = np.random.default_rng(0).standard_normal((n,) + shape, dtype=np.float32)
stage = torch.empty((n,) + shape, dtype=torch.float32).pin_memory()
dst = stage.numpy()
mb = x.nbytes / 1e6
for nt in (1, 2, 4, 8, 16, 32):
    pool = ThreadPoolExecutor(max_workers=nt)
    step = (n + nt - 1) // nt
    def run():
        list(pool.map(lambda s: np.copyto(dst[s:s + step], x[s:s + step]), range(0, n, step)))
    run()
    t = time.perf_counter()
    for _ in range(5):
        run()
    dt = (time.perf_counter() - t) / 5
    print(f"numpy copyto, {nt:2d} threads: {dt * 1e3:6.2f} ms  {mb / dt / 1e3:6.1f} GB/s", flush=True)
    pool.shutdown()
for nt in (1, 8, 16, 32):
    torch.set_num_threads(nt)
    src = torch.from_numpy(x)
    stage.copy_(src)
    t = time.perf_counter()
    for _ in range(5):
        stage.copy_(src)
    dt = (time.perf_counter() - t) / 5
    print(f"torch copy_,  {nt:2d} threads: {dt * 1e3:6.2f} ms  {mb / dt / 1e3:6.1f} GB/s", flush=True)
d = torch.empty((n,) + shape, device="cuda")
d.copy_(stage, non_blocking=True); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    d.copy_(stage, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(f"H2D from pinned: {dt * 1e3:6.2f} ms  {mb / dt / 1e3:6.1f} GB/s")
t = time.perf_counter()
for _ in range(3):
    d.copy_(torch.from_numpy(x))
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 3
print(f"H2D from pageable: {dt * 1e3:6.2f} ms  {mb / dt / 1e3:6.1f} GB/s")
