#!/usr/bin/env python3
"""predict_batch (numpy in -> numpy out, 06:308-406) against its own roofs.

    python tools/api_probe.py [n_chunks=3] [steps=300]

Prints, for n_chunks x 4096 windows per call: the device-resident time of the same work, the API-level time, and the
pieces the difference is made of -- bytes over PCIe each way and the link rate they reach, the pageable -> page-locked
staging copy, the first touch of the freshly allocated result arrays, the host copy of the results."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import CognitiveStateODE, EnhancedLSTMModel, LSTMODEIntegration        # noqa: E402
from lstm_ode_bci_amd import integration as I                                                # noqa: E402
from lstm_ode_bci_amd import synthetic as syn                                                # noqa: E402

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
B, T, C, H = 4096, 256, 61, 128
dev = torch.device("cuda:0")
sd = syn.make_state_dict(C, H, 3, 2, True)
m = EnhancedLSTMModel(C, H, 3, 2, 0.4, True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.to(dev).eval()
integ = LSTMODEIntegration(m, CognitiveStateODE(), coupling_strength=0.5)
integ.use_amp = False
integ.ramp_chunk = int(os.environ.get("RAMP", integ.ramp_chunk))
integ.stage_piece = int(os.environ.get("PIECE", integ.stage_piece))
print("ramp_chunk", integ.ramp_chunk, "stage_piece", integ.stage_piece)
n = nch * B
x = np.random.default_rng(1).standard_normal((n, T, C), dtype=np.float32)
print(f"host: {os.cpu_count()} cpus, affinity {len(os.sched_getaffinity(0))}, copy threads {I._COPY_THREADS}; "
      f"{nch} x {B} windows, {steps} points", flush=True)


def wall(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return min(ts)


xd = torch.from_numpy(x[:B]).to(dev)
t_dev = wall(lambda: integ.predict_batch_device(xd, steps, B))
t_api = wall(lambda: integ.predict_batch(x, steps, B, show_progress=False))
up, down = x.nbytes, n * (steps * 24 + 8 + 8)
print(f"device-resident: {t_dev * 1e3:7.2f} ms per {B} windows = {B / t_dev:9.0f} windows/s")
print(f"predict_batch  : {t_api * 1e3:7.2f} ms per call, {t_api / nch * 1e3:6.2f} per {B} = {n / t_api:9.0f} windows/s "
      f"({t_api / (nch * t_dev):.2f} x the device-resident time)")
print(f"PCIe bytes per call: up {up / 1e6:.0f} MB, down {down / 1e6:.0f} MB; link time at the measured rates below")

# the pieces
stage = torch.empty((B, T, C), dtype=torch.float32).pin_memory()
pool = I._copy_pool()
nt = I._COPY_THREADS
step = (B + nt - 1) // nt
dst = stage.numpy()


def stage_copy():
    list(pool.map(lambda s: np.copyto(dst[s:s + step], x[s:s + step]), range(0, B, step)))


t_stage = wall(stage_copy, 5)
d = torch.empty((B, T, C), device=dev)
t_h2d = wall(lambda: d.copy_(stage, non_blocking=True), 5)
traj_d = torch.empty((B, steps, 3), dtype=torch.float64, device=dev)
stage_o = torch.empty((B, steps, 3), dtype=torch.float64).pin_memory()
t_d2h = wall(lambda: stage_o.copy_(traj_d, non_blocking=True), 5)
so = stage_o.numpy()


def first_touch():
    out = np.empty((B, steps, 3), dtype=np.float64)
    st = (B + nt - 1) // nt
    list(pool.map(lambda s: np.copyto(out[s:s + st], so[s:s + st]), range(0, B, st)))
    return out


out_warm = np.empty((B, steps, 3), dtype=np.float64)
out_warm[:] = 0


def warm_copy():
    st = (B + nt - 1) // nt
    list(pool.map(lambda s: np.copyto(out_warm[s:s + st], so[s:s + st]), range(0, B, st)))


t_ft = wall(first_touch, 5)
t_wc = wall(warm_copy, 5)
mb_in, mb_out = B * T * C * 4 / 1e6, B * steps * 24 / 1e6
print(f"per {B}-window chunk:")
print(f"  pageable -> page-locked staging copy ({nt} threads): {t_stage * 1e3:6.2f} ms = {mb_in / t_stage / 1e3:5.1f} GB/s")
print(f"  H2D {mb_in:.0f} MB from page-locked memory            : {t_h2d * 1e3:6.2f} ms = {mb_in / t_h2d / 1e3:5.1f} GB/s (spec 63)")
print(f"  D2H {mb_out:.1f} MB of trajectories                    : {t_d2h * 1e3:6.2f} ms = {mb_out / t_d2h / 1e3:5.1f} GB/s")
print(f"  results -> freshly allocated arrays (first touch)   : {t_ft * 1e3:6.2f} ms; into touched pages {t_wc * 1e3:6.2f} ms")
fill = t_stage + t_h2d
drain = t_d2h + t_ft
print(f"pipeline model: fill (stage + H2D of chunk 0) {fill * 1e3:.2f} + {nch} x {t_dev * 1e3:.2f} + drain (D2H + host copy of the last "
      f"chunk) {drain * 1e3:.2f} = {(fill + nch * t_dev + drain) * 1e3:.2f} ms; measured {t_api * 1e3:.2f}")
print(f"link occupancy during the call: up {up / t_api / 1e9:.1f} GB/s, down {down / t_api / 1e9:.1f} GB/s of 63 GB/s each way "
      f"-> the link is {'NOT ' if up / t_api / 1e9 < 40 else ''}the bound")

# ---- where inside the call the GPU waits: events around every chunk's kernels (start of the LSTM pass, end of the ODE kernel)
from lstm_ode_bci_amd import ops as _ops                      # noqa: E402
marks = []
_pd, _ode = integ._probs_device, _ops.ode_rk4


def pd(*a, **k):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append(["start", e, time.perf_counter()])
    return _pd(*a, **k)


def ode(*a, **k):
    r = _ode(*a, **k)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append(["end", e, time.perf_counter()])
    return r


integ._probs_device = pd
I.ops.ode_rk4 = ode
t0 = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
h0 = time.perf_counter()
t0.record()
I._probe = []
integ.predict_batch(x, steps, B, show_progress=False)
h1 = time.perf_counter()
torch.cuda.synchronize()
print("host stamps (ms since the call): " + "; ".join(f"{lab} {(t - h0) * 1e3:.2f}" for lab, t in I._probe))
I._probe = None
print(f"instrumented call: {(h1 - h0) * 1e3:.2f} ms host wall; per chunk on the GPU timeline (ms since the call started):")
prev_end = 0.0
for i in range(0, len(marks), 2):
    s, e = marks[i], marks[i + 1]
    ts, te = t0.elapsed_time(s[1]), t0.elapsed_time(e[1])
    print(f"  chunk {i // 2}: kernels {ts:7.2f} -> {te:7.2f} ({te - ts:5.2f} ms), GPU idle before it {ts - prev_end:5.2f} ms; "
          f"host enqueued it at {(s[2] - h0) * 1e3:7.2f} .. {(e[2] - h0) * 1e3:7.2f}")
    prev_end = te
print(f"  after the last kernel: {(h1 - h0) * 1e3 - prev_end:5.2f} ms (download + host copy of the last chunk)")
