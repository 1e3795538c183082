"""Parity of the 8-wave ping-pong GEMMs (csrc/gemm_pp.hip) that carry the matrix-bound GEMMs of the H = 256 mixed
training step (the reference's real checkpoint size, 04_lstm_model.py:877; training step 04:482-512).

* small ragged-in-tiles shapes (several output tiles per workgroup, several workgroups per XCD, one k-tile pair ...)
  against a float64 product of the same bf16 operands: these kernels synchronise their LDS-DMA by hand-counted
  ``s_waitcnt vmcnt(N)`` across barriers, so every schedule edge (first / last k-tile, tile change, tail) gets a case;
* the step's full shapes (rows = 256 * 4096) against the tiled / weight-stationary twins selected through the test-only
  variant table: same MFMA, same k order -> bit-identical (NT kernels); split-k + atomics -> fp32 rounding (TN).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from lstm_ode_bci_amd import _lib
    assert _lib.lib().lob_version() >= 200
    return torch.device("cuda:0")


def _rand(shape, dev, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=dev) * scale).to(dtype)


def _ref_nt(a, w):
    return a.double() @ w.double().t()


@pytest.mark.parametrize("M,N,K", [(256, 256, 2048), (512, 512, 2048), (2304, 512, 1024), (256 * 70, 256, 2048),
                                   (256 * 9, 768, 1152)])
@pytest.mark.parametrize("out_bf16", [True, False])
@pytest.mark.parametrize("PP", [7, 15, 7 | 512])
def test_nt_pp_small_shapes_vs_float64(dev, M, N, K, out_bf16, PP):
    """dX-shaped products through lob_gemm_nt_bf16 (K >= 1024 routes to the ping-pong kernel; PP = 15: the schedule with
    16-MFMA segments and the triple-buffered A half)."""
    from lstm_ode_bci_amd import _lib, ops
    a = _rand((M, K), dev, 1, 1.0, torch.bfloat16)
    w = _rand((N, K), dev, 2, 0.05, torch.bfloat16)
    with _lib.variant(GEMM_PP=PP):
        c = ops.gemm_nt(a, w, mixed=True, out_bf16=out_bf16)
        c2 = ops.gemm_nt(a, w, mixed=True, out_bf16=out_bf16)
    ref = _ref_nt(a, w)
    assert torch.equal(c, c2)
    tol = (1e-2 if out_bf16 else 2e-5) * float(ref.abs().max())
    err = float((c.double() - ref).abs().max())
    assert err <= tol, (err, tol)
    with _lib.variant(GEMM_PP=0):
        ct = ops.gemm_nt(a, w, mixed=True, out_bf16=out_bf16)
    if PP & 512:                         # v_mfma_f32_16x16x32_bf16: 32 products per instruction, another summation tree
        assert (c.double() - ct.double()).abs().max().item() <= tol
    else:
        assert torch.equal(c, ct)        # same MFMA, same k order as the tiled kernel


def test_nt_pp_dropout_epilogue_matches_standalone_mask(dev):
    """The fused dropout-backward epilogue applies the mask of element (row * ldc + col), like lob_dropout_f32."""
    from lstm_ode_bci_amd import _lib, ops
    M, N, K = 1024, 512, 2048
    a = _rand((M, K), dev, 3, 1.0, torch.bfloat16)
    w = _rand((N, K), dev, 4, 0.05, torch.bfloat16)
    with _lib.variant(GEMM_PP=7):
        plain = ops.gemm_nt(a, w, mixed=True)
        dropped = ops.gemm_nt(a, w, mixed=True, drop_p=0.4, seed=1234)
    want = ops.dropout(plain, 0.4, 1234)
    assert torch.equal(dropped, want)
    frac = float((dropped == 0).float().mean())
    assert 0.38 < frac < 0.42


@pytest.mark.parametrize("T,Bp,K", [(8, 32, 512), (8, 64, 256), (24, 96, 512), (256, 32, 512)])
@pytest.mark.parametrize("PP", [7, 15])
def test_gate_pp_small_shapes_vs_weight_stationary(dev, T, Bp, K, PP):
    """Fragment-order P of the H = 256 gate GEMM: ping-pong kernel against the weight-stationary twin (bit-identical:
    same MFMA, same k order, bias added last) and against a float64 product through the recurrent kernel's layout."""
    from lstm_ode_bci_amd import _lib, ops
    H, D = 256, 2
    x = _rand((T * Bp, K), dev, 5, 1.0, torch.bfloat16)
    w = _rand((D * 4 * H, K), dev, 6, 0.05, torch.bfloat16)
    bias = _rand((D * 4 * H,), dev, 7, 0.1)
    with _lib.variant(GEMM_PP=PP):
        p_pp = ops.gate_gemm_x(x, w, bias, T, Bp, H, D, True, mixed=True)
    with _lib.variant(GEMM_PP=0, GATE_WS=1):
        p_ws = ops.gate_gemm_x(x, w, bias, T, Bp, H, D, True, mixed=True)
    assert p_pp.dtype == torch.bfloat16 and torch.equal(p_pp, p_ws)
    # un-permute the fragment order [d][t][bt][w][gate][q pair][lane][8] and compare with the float64 product
    ref = (_ref_nt(x, w) + bias.double()).reshape(T, Bp // 32, 32, D, 4, H // 32, 32)       # t, bt, row, d, gate, w, col
    frag = p_pp.reshape(D, T, Bp // 32, H // 32, 4, 2, 64, 8).double()
    lane = torch.arange(64, device=dev)
    got = torch.empty_like(ref)
    for pq in range(2):
        for e in range(8):
            r = 8 * pq + e
            row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)          # acc_row(r, lane)
            col = lane & 31
            v = frag[:, :, :, :, :, pq, :, e]                        # d, t, bt, w, gate, lane
            got[:, :, row, :, :, :, col] = v.permute(5, 1, 2, 0, 4, 3)     # (lane, t, bt, d, gate, w)
    assert float((got - ref).abs().max()) <= 1e-2 * float(ref.abs().max())


@pytest.mark.parametrize("M,N,Kc", [(256, 256, 1024), (512, 256, 2048 + 128), (1024, 512, 4096), (2048, 512, 16384)])
@pytest.mark.parametrize("PP", [7, 15])
def test_tn_pp_small_shapes_vs_float64(dev, M, N, Kc, PP):
    from lstm_ode_bci_amd import _lib, ops
    a = _rand((Kc, M), dev, 8, 1e-2, torch.bfloat16)
    b = _rand((Kc, N), dev, 9, 1.0, torch.bfloat16)
    with _lib.variant(GEMM_PP=PP):
        c = ops.gemm_tn(a, b, torch.zeros((M, N), device=dev))
    ref = a.double().t() @ b.double()
    err = float((c.double() - ref).abs().max())
    assert err <= 2e-5 * float(ref.abs().max()) + 1e-7, err
    # column slices of wider tensors (the per-direction dW_hh call of backward.py)
    wide_a = _rand((Kc, 2 * M), dev, 10, 1e-2, torch.bfloat16)
    wide_b = _rand((Kc, 2 * N), dev, 11, 1.0, torch.bfloat16)
    with _lib.variant(GEMM_PP=PP):
        c2 = ops.gemm_tn(wide_a[:, M:], wide_b[:, N:], torch.zeros((M, N), device=dev))
    ref2 = wide_a[:, M:].double().t() @ wide_b[:, N:].double()
    assert float((c2.double() - ref2).abs().max()) <= 2e-5 * float(ref2.abs().max()) + 1e-7


@pytest.mark.parametrize("PP", [7, 15])
def test_pp_full_size_against_twins(dev, PP):
    """rows = 256 * 4096 (the bench's H = 256 step): dX bit-identical to the tiled LDS-DMA kernel, the gate GEMM
    bit-identical to the weight-stationary kernel, the weight gradients equal to fp32 rounding of split-k sums; every
    ping-pong result reproduces itself (a stale-LDS read under load would not)."""
    from lstm_ode_bci_amd import _lib, ops
    T, Bp, H, D = 256, 4096, 256, 2
    rows = T * Bp
    dP = _rand((rows, D * 4 * H), dev, 12, 1e-2, torch.bfloat16)
    for N in (512, 256):
        wt = _rand((N, D * 4 * H), dev, 13, 0.05, torch.bfloat16)
        with _lib.variant(GEMM_PP=PP):
            a = ops.gemm_nt(dP, wt, mixed=True, out_bf16=True, drop_p=0.4, seed=77)
            a2 = ops.gemm_nt(dP, wt, mixed=True, out_bf16=True, drop_p=0.4, seed=77)
        with _lib.variant(GEMM_PP=0):
            b = ops.gemm_nt(dP, wt, mixed=True, out_bf16=True, drop_p=0.4, seed=77)
        assert torch.equal(a, a2) and torch.equal(a, b)
        del a, a2, b
    bias = _rand((D * 4 * H,), dev, 14, 0.1)
    for K in (512, 256):
        x = _rand((rows, K), dev, 15, 1.0, torch.bfloat16)
        w = _rand((D * 4 * H, K), dev, 16, 0.05, torch.bfloat16)
        with _lib.variant(GEMM_PP=PP):
            p = ops.gate_gemm_x(x, w, bias, T, Bp, H, D, True, mixed=True)
            p2 = ops.gate_gemm_x(x, w, bias, T, Bp, H, D, True, mixed=True)
        with _lib.variant(GEMM_PP=0, GATE_WS=1):
            q = ops.gate_gemm_x(x, w, bias, T, Bp, H, D, True, mixed=True)
        assert torch.equal(p, p2) and torch.equal(p, q)
        del p, p2, q
        with _lib.variant(GEMM_PP=PP):
            dw = ops.gemm_tn(dP, x, torch.zeros((D * 4 * H, K), device=dev))
        with _lib.variant(GEMM_PP=0):
            dw0 = ops.gemm_tn(dP, x, torch.zeros((D * 4 * H, K), device=dev))
        scale = float(dw0.abs().max())
        assert float((dw - dw0).abs().max()) <= 2e-5 * scale
    y = _rand((rows, D * H), dev, 17, 1.0, torch.bfloat16)
    with _lib.variant(GEMM_PP=PP):
        dwh = ops.gemm_tn(dP[Bp:, :4 * H], y[:(T - 1) * Bp, :H], torch.zeros((4 * H, H), device=dev))
    with _lib.variant(GEMM_PP=0):
        dwh0 = ops.gemm_tn(dP[Bp:, :4 * H], y[:(T - 1) * Bp, :H], torch.zeros((4 * H, H), device=dev))
    assert float((dwh - dwh0).abs().max()) <= 2e-5 * float(dwh0.abs().max())


@pytest.mark.parametrize("T,Bp,nx", [(2, 64, 512), (4, 64, 256), (9, 128, 512), (6, 192, 256), (256, 64, 512), (33, 256, 512)])
def test_fused_h256_layer_weight_gradients_vs_float64(dev, T, Bp, nx):
    """lstm_dw_pp_kernel (256 x 384 / 256 x 256 ping-pong tiles): dW_ih and dW_hh of an H = 256 layer from one pass over
    dP == the exact products of the same bf16 operands, including the time shift of h_prev (t - 1 forward, t + 1
    reverse), the step without a predecessor, operands that are column slices of wider tensors, several contraction
    chunks per tile and a ragged last chunk."""
    from lstm_ode_bci_amd import ops
    H, D = 256, 2
    bf = torch.bfloat16
    dP = _rand((T * Bp, D * 4 * H), dev, T * Bp + nx, 1.0, bf)
    Xw = _rand((T * Bp, nx + 64), dev, 1 + nx, 1.0, bf)
    X = Xw[:, :nx]
    Y = _rand((T * Bp, D * H), dev, 2 + T, 1.0, bf)
    assert ops.can_fuse_dw(dP, X, Y, T, Bp, H, D)
    dwih, dwhh = ops.lstm_dw(dP, X, Y, T, Bp, H, D)
    dwih2, dwhh2 = ops.lstm_dw(dP, X, Y, T, Bp, H, D)
    p64, x64, y64 = dP.double(), X.double(), Y.double()
    tol = 2e-5 * (T * Bp) ** 0.5 * 4
    assert (dwih.double() - p64.T @ x64).abs().max().item() < tol
    for d in range(D):
        a = p64[:, d * 4 * H:(d + 1) * 4 * H]
        y = y64[:, d * H:(d + 1) * H]
        ref = a[Bp:].T @ y[:(T - 1) * Bp] if d == 0 else a[:(T - 1) * Bp].T @ y[Bp:]
        assert (dwhh[d].double() - ref).abs().max().item() < tol, d
    assert (dwih - dwih2).abs().max().item() <= 1e-4 * dwih.abs().max().item()      # atomics: order only
    assert not ops.can_fuse_dw(dP, X, Y, T, 32, H, D) and not ops.can_fuse_dw(dP.float(), X, Y, T, Bp, H, D)


def test_fused_h256_dw_full_size_against_separate_gemms(dev):
    """rows = 256 * 4096: the fused kernel against three launches of the 256 x 256 TN kernel (split-k + atomics on both
    sides: fp32 rounding of 1M-term sums)."""
    from lstm_ode_bci_amd import _lib, ops
    T, Bp, H, D = 256, 4096, 256, 2
    rows = T * Bp
    dP = _rand((rows, D * 4 * H), dev, 41, 1e-2, torch.bfloat16)
    Y = _rand((rows, D * H), dev, 42, 1.0, torch.bfloat16)
    for nx in (512, 256):
        X = _rand((rows, nx), dev, 43, 1.0, torch.bfloat16)
        dwih, dwhh = ops.lstm_dw(dP, X, Y, T, Bp, H, D)
        ref_ih = ops.gemm_tn(dP, X, torch.zeros_like(dwih))
        ref_hh = torch.zeros_like(dwhh)
        ops.gemm_tn(dP[Bp:, :4 * H], Y[:(T - 1) * Bp, :H], ref_hh[0])
        ops.gemm_tn(dP[:(T - 1) * Bp, 4 * H:], Y[Bp:, H:], ref_hh[1])
        for got, ref, nm in ((dwih, ref_ih, "dW_ih"), (dwhh, ref_hh, "dW_hh")):
            err = (got - ref).abs().max().item()
            assert err <= 2e-4 * ref.abs().max().item(), (nx, nm, err, ref.abs().max().item())
