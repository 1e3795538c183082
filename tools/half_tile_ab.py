#!/usr/bin/env python3
"""Same-process A/B of LOB_VAR_REC_HALF on the fp32 forward (configs[1]: B = 1024) and smaller batches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lstm_ode_bci_amd import _lib
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
for B in (1024, 512):
    for prec in ("fp32", "mixed"):
        leg = bench.Leg(dev, "fwd", prec, B, 128, 300)
        res = {}
        for rep in range(3):
            for v in (4, 2, 0):
                with _lib.variant(REC_HALF=v):
                    dt = leg.run(20, 5)
                res.setdefault(v, []).append(dt / 20 * 1e3)
        print(f"fwd {prec} B={B}: quarter tiles {min(res[4]):.3f} ms | half tiles {min(res[2]):.3f} ms | full tiles {min(res[0]):.3f} ms", flush=True)
        leg.free()
