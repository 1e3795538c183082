#!/usr/bin/env python3
"""fp32 path, H = 128, B = 4096: the backward kernels on the fp16-split arithmetic against their exact-fp32 twins."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import _lib, ops
dev = torch.device("cuda:0")
T, Bp, H, D = 256, 4096, 128, 2
rows = T * Bp
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda shape, s: torch.randn(shape, generator=g, device=dev) * s


def timeit(fn, n=4):
    fn(); fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in ev)[n // 2]


dP = rnd((rows, D * 4 * H), 1e-4)
one = torch.ones(1, device=dev)
amax_dp = dP.abs().max().reshape(1)
for K in (256, 128):
    wt = rnd((K, D * 4 * H), 0.05)
    x = rnd((rows, K), 0.5)
    aw = wt.abs().max().reshape(1)
    dw = torch.zeros((D * 4 * H, K), device=dev)
    s_nt = timeit(lambda: ops.gemm_nt(dP, wt, amax=(amax_dp, aw)))
    s_tn = timeit(lambda: ops.gemm_tn(dP, x, dw, amax=(amax_dp, one)))
    with _lib.variant(F32_SPLIT=0):
        e_nt = timeit(lambda: ops.gemm_nt(dP, wt))
        e_tn = timeit(lambda: ops.gemm_tn(dP, x, dw))
    with _lib.variant(F32_SPLIT=2):
        f_nt = timeit(lambda: ops.gemm_nt(dP, wt, amax=(amax_dp, aw)))
        f_tn = timeit(lambda: ops.gemm_tn(dP, x, dw, amax=(amax_dp, one)))
    print(f"K_in={K}: dX split {s_nt:.3f} ms (split at fragment read {f_nt:.3f}, exact {e_nt:.3f}) | dW_ih split {s_tn:.3f} ms "
          f"(fragment {f_tn:.3f}, exact {e_tn:.3f})", flush=True)
y = rnd((rows, D * H), 0.5)
dwh = torch.zeros((4 * H, H), device=dev)
a_sl, y_sl = dP[:, :4 * H], y[:, :H]
s_hh = timeit(lambda: ops.gemm_tn(a_sl[Bp:], y_sl[:(T - 1) * Bp], dwh, amax=(amax_dp, one)))
with _lib.variant(F32_SPLIT=0):
    e_hh = timeit(lambda: ops.gemm_tn(a_sl[Bp:], y_sl[:(T - 1) * Bp], dwh))
print(f"dW_hh (one direction): split {s_hh:.3f} ms (exact {e_hh:.3f})", flush=True)
del dP, y
torch.cuda.empty_cache()
x = rnd((rows, H), 1.0)
wih, bias, whh = rnd((D * 4 * H, H), 0.08), rnd((D * 4 * H,), 0.1), rnd((D, 4 * H, H), 0.08)
P = ops.gate_gemm_x(x, wih, bias, T, Bp, H, D, True, mixed=False, exact=True)
Y, Cs, _, _ = ops.lstm_rec_fwd(P, whh, T, Bp, H, D, True)
dY = rnd((rows, D * H), 1e-3)
rng = whh.abs().amax(dim=(1, 2)).contiguous()
am = torch.zeros(1, device=dev)
s_b = timeit(lambda: ops.lstm_rec_bwd(P, Cs, whh, dY, T, Bp, H, D, amax_out=am, range=rng))
with _lib.variant(F32_SPLIT=0):
    e_b = timeit(lambda: ops.lstm_rec_bwd(P, Cs, whh, dY, T, Bp, H, D))
print(f"BPTT: split {s_b:.3f} ms (exact {e_b:.3f})", flush=True)
