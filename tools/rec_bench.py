#!/usr/bin/env python3
"""Recurrent forward (saving) and BPTT of the mixed path alone, B = 4096, T = 256: `python tools/rec_bench.py [H]`.
With LOB_LIB_PATH=ab/liblob_abl<k>.so (tools/h256_ablate.sh) the diagnostic builds of the H = 256 kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import ops

dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, D, B = 256, 2, 4096
Bp = ops.ceil32(B)
rows = T * Bp
g = torch.Generator(device=dev).manual_seed(1)


def rnd(shape, scale=1.0, dtype=torch.float32):
    return (torch.randn(shape, generator=g, device=dev) * scale).to(dtype)


def timeit(fn, n=6, before=None):
    for _ in range(2):
        if before: before()
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        if before: before()
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2]


x = rnd((rows, 2 * H), 1.0, torch.bfloat16)
w = rnd((D * 4 * H, 2 * H), 0.04, torch.bfloat16)
bias = rnd((D * 4 * H,), 0.1)
whh = rnd((D, 4 * H, H), 0.04)
P = ops.gate_gemm_x(x, w, bias, T, Bp, H, D, True, mixed=True)
G = P.clone()
kw = dict(mixed=True, want_f32=False, want_bf16=True)
ms_f = timeit(lambda: ops.lstm_rec_fwd(G, whh, T, Bp, H, D, True, **kw), before=lambda: G.copy_(P))
ms_d = timeit(lambda: ops.lstm_rec_fwd(G, whh, T, Bp, H, D, True, drop_p=0.4, seed=5, **kw), before=lambda: G.copy_(P))
ms_i = timeit(lambda: ops.lstm_rec_fwd(P, whh, T, Bp, H, D, False, **kw))
G.copy_(P)
Y, Cs, Y16, _ = ops.lstm_rec_fwd(G, whh, T, Bp, H, D, True, **kw)
dY = rnd((rows, D * H), 1e-3, torch.bfloat16)
ms_b = timeit(lambda: ops.lstm_rec_bwd(G, Cs, whh, dY, T, Bp, H, D, dp_bf16=True))
print(f"H={H} lib={os.environ.get('LOB_LIB_PATH', 'default')}: fwd save {ms_f:.3f} ms | save+dropout {ms_d:.3f} | "
      f"inference {ms_i:.3f} | BPTT {ms_b:.3f}", flush=True)
