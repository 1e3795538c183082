#!/usr/bin/env python3
"""Same-box A/B of a host-side switch of ops.py on the bench's own workloads: python tools/ab_switch.py FUSE_INPUT_PROJ
[--hidden 128] [--mode train|fwd|coupled] [--precision mixed|fp32] [--batch 4096] [--off-first].  Alternates the two settings three
times (12 + 4 steps each); --off-first starts with the switch off (whatever runs first in a process may carry one-time costs)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                     # noqa: E402
from lstm_ode_bci_amd import ops                 # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("switch")
ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--mode", default="train")
ap.add_argument("--precision", default="mixed")
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--off-first", action="store_true", help="start with the switch OFF (is a slow first repetition the switch's or the process's?)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
leg = bench.Leg(dev, a.mode, a.precision, a.batch, a.hidden, 300)
old = getattr(ops, a.switch)
res = {True: [], False: []}
try:
    for rep in range(3):
        for v in ((False, True) if a.off_first else (True, False)):
            setattr(ops, a.switch, v)
            dt = leg.run(12, 4)
            res[v].append(dt / 12 * 1e3)
finally:
    setattr(ops, a.switch, old)
for v in (True, False):
    print(f"{a.switch}={v}: " + "  ".join(f"{t:.3f}" for t in res[v]) + f"  ms/step (min {min(res[v]):.3f})", flush=True)
