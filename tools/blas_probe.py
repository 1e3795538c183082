#!/usr/bin/env python3
"""What the library GEMM (hipBLASLt / rocBLAS through torch.mm) reaches on the H = 256 step's plain GEMM shapes -- the
yardstick for the hand-written tiled kernels (gemm_nt_dma_kernel / gemm_tn_dma_kernel), not a product path."""
import time, torch
dev = torch.device("cuda:0")
M = 256 * 4096
def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for name, (m, n, k, tn) in {"dX  [M,2048]x[2048,512]": (M, 512, 2048, False), "gate [M,512]x[512,2048]": (M, 2048, 512, False),
                            "dW_ih [2048,M]x[M,512]": (2048, 512, M, True), "dW_hh [1024,M]x[M,256]": (1024, 256, M, True),
                            "H128 gate [M,256]x[256,1024]": (M, 1024, 256, False), "H128 dX [M,1024]x[1024,256]": (M, 256, 1024, False)}.items():
    if tn:
        a = torch.randn(k, m, device=dev, dtype=torch.bfloat16); b = torch.randn(k, n, device=dev, dtype=torch.bfloat16)
        f = lambda: torch.mm(a.t(), b)
    else:
        a = torch.randn(m, k, device=dev, dtype=torch.bfloat16); b = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
        f = lambda: torch.mm(a, b.t())
    dt = bench(f)
    print(f"{name:32s} {dt*1e3:8.3f} ms  {2*m*n*k/dt/1e12:8.1f} TFLOP/s", flush=True)
    del a, b
