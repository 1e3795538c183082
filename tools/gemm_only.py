#!/usr/bin/env python3
"""Runs only the bf16 NT DMA GEMMs of the training step (gate GEMM K=256, dX N=256) a few times: a small target for
rocprofv3 --pmc passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import ops
dev = torch.device("cuda:0")
T, B, H, D = 256, 4096, 128, 2
rows, N, K = T * B, D * 4 * H, D * H
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.randn((rows, K), generator=g).to(dev).to(torch.bfloat16)
w = (torch.rand((N, K), generator=g) * 0.17 - 0.085).to(dev).to(torch.bfloat16)
bias = torch.zeros(N, device=dev)
dP = torch.randn((rows, N), generator=g).to(dev).to(torch.bfloat16)
wt = (torch.rand((K, N), generator=g) * 0.1).to(dev).to(torch.bfloat16)
for _ in range(4):
    ops.gate_gemm_x(x, w, bias, T, B, H, D, True, mixed=True)
    ops.gemm_nt(dP, wt, mixed=True)
torch.cuda.synchronize()
