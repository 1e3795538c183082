#!/usr/bin/env python3
"""Per-kernel timing of the H = 256 mixed step's big kernels at B = 4096 (the reference's real checkpoint size,
04_lstm_model.py:877): gate GEMMs (weight-stationary vs tiled), recurrent forward / BPTT, dX, dW."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import _lib, ops

dev = torch.device("cuda:0")
T, H, D, B = 256, 256, 2, int(os.environ.get("KB_B", "4096"))
Bp = ops.ceil32(B)
rows = T * Bp
g = torch.Generator(device=dev).manual_seed(1)


def rnd(shape, scale=1.0, dtype=torch.float32):
    return (torch.randn(shape, generator=g, device=dev) * scale).to(dtype)


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def report(name, ms, flop, byts):
    print(f"{name:44s} {ms:8.3f} ms  {flop / ms / 1e9:8.1f} TFLOP/s  {byts / ms / 1e6:8.1f} GB/s", flush=True)


bias = rnd((D * 4 * H,), 0.1)
for K in (512, 256):
    x = rnd((rows, K), 1.0, torch.bfloat16)
    w = rnd((D * 4 * H, K), 0.04, torch.bfloat16)
    for v in (1, 0):
        with _lib.variant(GATE_WS=v):
            ms = timeit(lambda: ops.gate_gemm_x(x, w, bias, T, Bp, H, D, True, mixed=True))
        report(f"gate GEMM K={K} {'weight-stationary' if v else 'tiled LDS-DMA'}", ms, 2.0 * rows * K * D * 4 * H,
               2.0 * rows * (K + D * 4 * H))
x = rnd((rows, 512), 1.0, torch.bfloat16)
w = rnd((D * 4 * H, 512), 0.04, torch.bfloat16)
whh = rnd((D, 4 * H, H), 0.04)
P = ops.gate_gemm_x(x, w, bias, T, Bp, H, D, True, mixed=True)
G = P.clone()
ms = timeit(lambda: ops.lstm_rec_fwd(G.copy_(P), whh, T, Bp, H, D, True, mixed=True, want_f32=False, want_bf16=True)) - timeit(lambda: G.copy_(P))
report("rec fwd (save, Y16)", ms, 2.0 * rows * D * 4 * H * H, rows * D * (4 * H * 2 * 2 + H * 4 + H * 2))
ms = timeit(lambda: ops.lstm_rec_fwd(P, whh, T, Bp, H, D, False, mixed=True, want_f32=False, want_bf16=True))
report("rec fwd (inference, Y16)", ms, 2.0 * rows * D * 4 * H * H, rows * D * (4 * H * 2 + H * 2))
Y, Cs, Y16, _ = ops.lstm_rec_fwd(G.copy_(P), whh, T, Bp, H, D, True, mixed=True, want_f32=False, want_bf16=True)
dY = rnd((rows, D * H), 1e-3)
ms = timeit(lambda: ops.lstm_rec_bwd(G, Cs, whh, dY, T, Bp, H, D, dp_bf16=True))
report("rec BPTT", ms, 2.0 * rows * D * 4 * H * H, rows * D * (4 * H * 2 * 2 + H * 4 + H * 4))

# LDS-resident weight fragments (LOB_VAR_H256_LDSW) against every fragment streamed
for v in (1, 0):
    with _lib.variant(H256_LDSW=v):
        ms_f = timeit(lambda: ops.lstm_rec_fwd(G.copy_(P), whh, T, Bp, H, D, True, mixed=True, want_f32=False, want_bf16=True)) - timeit(lambda: G.copy_(P))
        ms_i = timeit(lambda: ops.lstm_rec_fwd(P, whh, T, Bp, H, D, False, mixed=True, want_f32=False, want_bf16=True))
        Y, Cs, Y16, _ = ops.lstm_rec_fwd(G.copy_(P), whh, T, Bp, H, D, True, mixed=True, want_f32=False, want_bf16=True)
        dY16 = dY.to(torch.bfloat16)
        ms_b = timeit(lambda: ops.lstm_rec_bwd(G, Cs, whh, dY16, T, Bp, H, D, dp_bf16=True))
    print(f"H256_LDSW={v}: rec fwd save {ms_f:.3f} ms, inference {ms_i:.3f} ms, BPTT (bf16 dY) {ms_b:.3f} ms", flush=True)

# weight gradients at H = 256 (tiled TN GEMM): dW_ih = dP^T X and, per direction, dW_hh = dP_d^T Y_d
dPb = rnd((rows, D * 4 * H), 1e-2, torch.bfloat16)
for K in (512, 256):
    X = rnd((rows, K), 1.0, torch.bfloat16)
    out = torch.zeros((D * 4 * H, K), device=dev)
    ms = timeit(lambda: ops.gemm_tn(dPb, X, out, mixed=True))
    report(f"dW_ih = dP^T X, X width {K}", ms, 2.0 * rows * K * D * 4 * H, 2.0 * rows * (K + D * 4 * H))
Yb = rnd((rows, D * H), 1.0, torch.bfloat16)
out = torch.zeros((4 * H, H), device=dev)
ms = timeit(lambda: ops.gemm_tn(dPb[Bp:, :4 * H], Yb[:(T - 1) * Bp, :H], out, mixed=True))
report("dW_hh (one direction) = dP_d^T Y_d", ms, 2.0 * rows * H * 4 * H, 2.0 * rows * (H + 4 * H))
