#!/usr/bin/env python3
"""Small-batch latency of the forward / coupled path (serving-style calls): ms per call (best and median of five
batches of --iters back-to-back calls, after a 30-ms warm-up) and windows/s."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import EnhancedLSTMModel, CognitiveStateODE, LSTMODEIntegration, synthetic as syn

dev = torch.device("cuda:0")
sd = syn.make_state_dict(61, 128, 3, 2, True)
m = EnhancedLSTMModel(61, 128, 3, 2, 0.4, True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.to(dev).eval()
integ = LSTMODEIntegration(m, CognitiveStateODE(), 0.5)
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="1,8,32,128,512")
ap.add_argument("--only", default="", help="substring filter on the leg name (e.g. 'mixed')")
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
for B in [int(b) for b in args.batches.split(",")]:
    x = torch.randn(B, 256, 61, device=dev)
    for name, fn in (("fwd fp32", lambda: m(x)),
                     ("fwd mixed", lambda: torch.autocast("cuda", dtype=torch.bfloat16).__enter__() or m(x)),
                     ("coupled fp32 (20 pts)", lambda: integ.predict_batch_device(x, forecast_steps=20, batch_size=max(B, 1)))):
        if args.only and args.only not in name:
            continue
        with torch.no_grad():
            if name == "fwd mixed":
                def fn():
                    with torch.autocast("cuda", dtype=torch.bfloat16):
                        return m(x)
            # warm-up: the first calls of a new shape / precision load code objects and grow the caching allocator, and an
            # idle GPU ramps its clocks over tens of milliseconds -- run until 30 ms of work have passed
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.03:
                fn()
                torch.cuda.synchronize()
            n = args.iters
            rates = []
            for _ in range(5):                 # five batches of n calls: min and median of the per-call time
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                rates.append((time.perf_counter() - t) / n)
            rates.sort()
            dt, med = rates[0], rates[2]
        print(f"B={B:4d} {name:24s} {dt * 1e3:8.3f} ms/call (median {med * 1e3:6.3f})  {B / dt:10.0f} windows/s", flush=True)
