#!/usr/bin/env python3
"""(Run by hand on a GPU box: python tests/mixed_error_report.py.)  Per-tensor relative gradient error of the mixed path against the oracle's fp32 autograd (max|diff| / max|ref|)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import EnhancedLSTMModel, synthetic as syn
from oracle import torch_cpu_path as TP

from lstm_ode_bci_amd import ops

dev = torch.device("cuda:0")
# PG_BF16 False: P / saved gates stay fp32 (only the GEMM inputs and dP are bf16) -- what bf16 STORAGE adds to the error
for pg in (True, False):
  ops.PG_BF16 = pg
  print(f"==== ops.PG_BF16 = {pg}")
  for (C, H, L, T, B) in ((61, 128, 3, 32, 6), (61, 128, 3, 256, 64)):
      sd = syn.make_state_dict(C, H, L, 2, True)
      x, y = syn.make_windows(B, T, C)
      ref = TP.build(sd, C, H)
      rl, rg, rgx = TP.loss_and_grads(ref, torch.from_numpy(x), torch.from_numpy(y))
      m = EnhancedLSTMModel(C, H, L, 2, 0.4, True)
      m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
      m = m.to(dev).eval()
      xg = torch.from_numpy(x).to(dev).requires_grad_(True)
      with torch.autocast("cuda", dtype=torch.bfloat16):
          loss = torch.nn.functional.cross_entropy(m(xg), torch.from_numpy(y).to(dev))
      loss.backward()
      print(f"T={T} B={B}: loss err {abs(loss.item() - rl):.2e}; grad_x rel {np.abs(xg.grad.cpu().numpy() - rgx).max() / np.abs(rgx).max():.2e}")
      errs = sorted(((float(np.abs(p.grad.cpu().numpy() - rg[k]).max() / max(np.abs(rg[k]).max(), 1e-30)), k, float(np.abs(rg[k]).max()))
                     for k, p in m.named_parameters()), reverse=True)
      for e, k, mx in errs[:8]:
          print(f"   {e:.2e}  {k:34s} max|ref| {mx:.2e}")

ops.PG_BF16 = True
