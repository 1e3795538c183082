#!/usr/bin/env python3
"""Small-batch latency of the forward / coupled path (serving-style calls): ms per call and windows/s."""
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
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            n = args.iters
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / n
        print(f"B={B:4d} {name:24s} {dt * 1e3:8.3f} ms/call  {B / dt:10.0f} windows/s", flush=True)
