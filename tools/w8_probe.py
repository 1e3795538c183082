#!/usr/bin/env python3
"""Mixed inference forward at several batch sizes with the eight-wave (LOB_VAR_REC_W8 = 2) and the four-wave (0)
recurrent kernels: where the cross-over is (the default uses the eight-wave kernel up to 256 workgroups)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import EnhancedLSTMModel, _lib, synthetic as syn

dev = torch.device("cuda:0")
sd = syn.make_state_dict(61, 128, 3, 2, True)
m = EnhancedLSTMModel(61, 128, 3, 2, 0.4, True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.to(dev).eval()
for B in (1, 32, 512, 1024, 2048, 4096):
    x = torch.randn(B, 256, 61, device=dev)
    line = f"B={B:5d}"
    for v in (0, 2):
        with _lib.variant(REC_W8=v), torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                m(x)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
        line += f"   {'8-wave' if v else '4-wave'} {ms:7.3f} ms"
    print(line, flush=True)
