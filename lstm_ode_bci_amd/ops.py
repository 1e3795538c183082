"""Thin host wrappers: torch tensors in, C-ABI calls out (raw device pointers + the current
HIP stream).  torch is plumbing here (allocator, streams); all arithmetic is in liblob.so."""
from __future__ import annotations

import contextlib
import ctypes as C

import numpy as np

import torch

from . import _lib

ACT_NONE, ACT_TANH, ACT_GELU = 0, 1, 2
LN_IDENTITY = 0x200          # LOB_LN_IDENTITY: OR into `act` of the LayerNorm entry points
OUT_BF16, DY_BF16, X_BF16 = 0x400, 0x800, 0x1000      # LOB_OUT_BF16 / LOB_DY_BF16 / LOB_X_BF16 (include/lob.h)

#: mixed mode, H == 128: carry the gradient between the LSTM layers (dX of the layer above = dY of the layer below), out
#: of the post-LSTM LayerNorm and into the input-projection LayerNorm as bf16 -- like dP, it is only consumed after
#: widening to fp32 (half the bytes of three streams of the backward).  Set to False to keep those carries fp32.
DY_BF16_CARRY = True


#: mixed mode, H == 128: the cell states saved for BPTT are stored as bf16 (the state carried through time stays fp32).
C_BF16 = True


def _rec16(H):
    """Hidden sizes whose bf16 recurrent kernels take bf16 cell states / a bf16 dY: the 16-row H == 128 kernels and the
    H == 256 kernels."""
    return (H == 128 and _lib.get_variant("REC_BF16_ROWS") != 32) or H == 256


def c_bf16_ok(H, mixed, p16):
    """bf16 storage of the saved cell states (with bf16 saved gates)."""
    return bool(mixed) and C_BF16 and bool(p16) and _rec16(H)


def dy_bf16_ok(H, mixed):
    """The bf16 BPTT kernels read a bf16 dY."""
    return bool(mixed) and DY_BF16_CARRY and PG_BF16 and _rec16(H)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_device(device):
    """Context manager for the high-level entry points: every lob_* launch goes to the CURRENT device on its current
    stream, so a model that lives on cuda:1 while cuda:0 is current must be run under a device guard.  A no-op when
    `device` is already current."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx == torch.cuda.current_device():
        return contextlib.nullcontext()
    return torch.cuda.device(idx)


def same_device(tensors, what):
    """All of `tensors` (None entries skipped) on one CUDA device, else LobError; returns that device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.LobError(f"{what}: expected device tensors (the product path has no CPU fallback)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise _lib.LobError(f"{what}: operands on different devices ({dev} and {t.device})")
    return dev


def _ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _chk(t, name, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.LobError(f"{name}: expected a device tensor (the product path has no CPU fallback)")
    if t.device.index != torch.cuda.current_device():
        # the launch would go to the current device with another device's pointers: a memory fault or silent peer access
        raise _lib.LobError(f"{name}: tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}"
                            " (run the call under torch.cuda.device(...): the model / integration entry points do)")
    if t.dtype != dtype:
        raise _lib.LobError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.LobError(f"{name}: expected a contiguous tensor")


def uses_frag(H):
    return bool(_lib.lib().lob_lstm_uses_fragment_layout(int(H)))


def ceil32(n):
    return (n + 31) // 32 * 32


def _bf16_ok(a, K, lda):
    return K % 8 == 0 and lda % 8 == 0 and a.data_ptr() % 16 == 0


def f32_split_gemm_ok(M, N, K, nt=True):
    """Shapes the fp16-split fp32 GEMMs take (lob_gemm_nt_f32_split / lob_gemm_tn_f32_split) when LOB_VAR_F32_SPLIT is on."""
    if _lib.get_variant("F32_SPLIT") == 0:
        return False
    return (K % 32 == 0 and K >= 128 and N <= 2048) if nt else (M % 4 == 0 and N % 4 == 0)


def gemm_nt(a, w, bias=None, act=ACT_NONE, out=None, accumulate=False, mixed=False, drop_p=0.0, seed=0, out_bf16=False,
            amax=None):
    """out[M,N] (+)= act(a[M,K] @ w[N,K]^T + bias).  mixed=True (or a bf16 `a`): bf16 MFMA inputs,
    fp32 accumulate/output; falls back to the exact-fp32 kernel for shapes the bf16 kernel refuses.
    out_bf16 (bf16 a AND w only): the result is stored as bf16.
    amax = (device float >= max|a|, device float >= max|w|), fp32 operands, no bias / act / accumulate: the products run
    as two-way fp16 splits on the 16-bit matrix pipe (22-bit products: the fp32 path's backward GEMMs)."""
    if (amax is not None and not mixed and a.dtype == torch.float32 and w.dtype == torch.float32 and bias is None
            and act == ACT_NONE and not accumulate and not out_bf16 and (drop_p == 0 or _lib.get_variant("F32_SPLIT") != 2)
            and f32_split_gemm_ok(a.shape[0], w.shape[0], a.shape[1]) and a.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and a.shape[1] % 4 == 0):
        _chk(a, "a"); _chk(w, "w"); _chk(amax[0], "amax_a"); _chk(amax[1], "amax_w")
        M, K = a.shape
        N = w.shape[0]
        if out is None:
            out = torch.empty((M, N), device=a.device, dtype=torch.float32)
        _chk(out, "out")
        rc = _lib.lib().lob_gemm_nt_f32_split(_ptr(a), K, _ptr(w), K, _ptr(out), N, M, N, K, _ptr(amax[0]), _ptr(amax[1]),
                                              float(drop_p), C.c_uint64(seed), _stream())
        _lib.check(rc, "lob_gemm_nt_f32_split")
        return out
    a_bf16 = a.dtype == torch.bfloat16
    w_bf16 = w.dtype == torch.bfloat16
    _chk(a, "a", a.dtype if a_bf16 else torch.float32); _chk(w, "w", w.dtype if w_bf16 else torch.float32)
    _chk(bias, "bias")
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out_bf16 and not (a_bf16 and w_bf16):
        raise _lib.LobError("gemm_nt: out_bf16 needs bf16 operands (the LDS-DMA kernel's epilogue)")
    caller_out = out is not None
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    _chk(out, "out", torch.bfloat16 if out_bf16 else torch.float32)
    if accumulate:
        act = act | 0x100
    if out_bf16:
        act = act | OUT_BF16
    if (mixed or a_bf16) and _bf16_ok(a, K, K) and w.data_ptr() % 16 == 0:
        rc = _lib.lib().lob_gemm_nt_bf16(_ptr(a), 1 if a_bf16 else 0, K, _ptr(w), int(w_bf16), K, _ptr(bias),
                                         _ptr(out), N, M, N, K, act, float(drop_p), C.c_uint64(seed), _stream())
        _lib.check(rc, "lob_gemm_nt_bf16")
        return out
    if out_bf16:
        raise _lib.LobError("gemm_nt: out_bf16 with operands the bf16 LDS-DMA kernel does not take (alignment / K % 8)")
    if a_bf16:
        raise _lib.LobError("gemm_nt: bf16 operand with a shape the bf16 kernel does not support")
    if drop_p > 0 and (accumulate or not out.is_contiguous()):
        raise _lib.LobError("gemm_nt: the dropout epilogue on the exact-fp32 kernel is a second pass over a fresh, contiguous result")
    rc = _lib.lib().lob_gemm_nt_f32(_ptr(a), K, _ptr(w), K, _ptr(bias), _ptr(out), N, M, N, K, act, _stream())
    _lib.check(rc, "lob_gemm_nt_f32")
    if drop_p > 0:        # the exact-fp32 kernel has no mask epilogue: the same mask as a pass of its own (element row * N + col)
        masked = dropout(out, drop_p, seed)
        return out.copy_(masked) if caller_out else masked
    return out


def gemm_tn(a, b, out, mixed=False, amax=None):
    """out[M,N] += a[Kc,M]^T @ b[Kc,N]; a, b may be column slices of wider row-major tensors
    (fp32 or bf16 storage).  mixed=True or any bf16 operand -> bf16 MFMA, fp32 accumulate.
    amax = (device float >= max|a|, device float >= max|b|) with fp32 operands: two-way fp16 split products (gemm_nt)."""
    Kc, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == Kc and a.stride(1) == 1 and b.stride(1) == 1
    _chk(out, "out")
    a16, b16 = a.dtype == torch.bfloat16, b.dtype == torch.bfloat16
    if (amax is not None and not mixed and not a16 and not b16 and f32_split_gemm_ok(M, N, Kc, nt=False)
            and a.stride(0) % 4 == 0 and b.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0):
        _chk(amax[0], "amax_a"); _chk(amax[1], "amax_b")
        rc = _lib.lib().lob_gemm_tn_f32_split(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), out.stride(0),
                                              M, N, Kc, _ptr(amax[0]), _ptr(amax[1]), _stream())
        _lib.check(rc, "lob_gemm_tn_f32_split")
        return out
    am, bm = (8 if a16 else 4), (8 if b16 else 4)
    ok16 = (M % am == 0 and a.stride(0) % am == 0 and N % bm == 0 and b.stride(0) % bm == 0
            and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)
    if (a16 or b16) and not ok16:
        raise _lib.LobError("gemm_tn: bf16 operand with a shape/alignment the bf16 kernel does not support")
    if (mixed and ok16) or a16 or b16:
        rc = _lib.lib().lob_gemm_tn_bf16(_ptr(a), int(a16), a.stride(0), _ptr(b), int(b16), b.stride(0),
                                         _ptr(out), out.stride(0), M, N, Kc, _stream())
        _lib.check(rc, "lob_gemm_tn_bf16")
        return out
    rc = _lib.lib().lob_gemm_tn_f32(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), out.stride(0),
                                    M, N, Kc, _stream())
    _lib.check(rc, "lob_gemm_tn_f32")
    return out


def fused_dw_enabled():
    """dW_ih and dW_hh of a layer from one pass over dP (LOB_VAR_FUSED_DW = 0: two/three separate TN GEMMs)."""
    return _lib.get_variant("FUSED_DW") != 0


def can_fuse_dw(dP, inp, Y, T, Bp, H, D):
    bf = torch.bfloat16
    shape_ok = ((H == 128 and Bp % 32 == 0 and inp.shape[1] in (128, 256)) or
                (H == 256 and D == 2 and Bp % 64 == 0 and (T * Bp) % 128 == 0 and inp.shape[1] in (256, 512)))
    return (fused_dw_enabled() and shape_ok and T >= 2 and dP.dtype == bf and inp.dtype == bf and Y.dtype == bf
            and dP.shape[1] == D * 4 * H and Y.shape[1] == D * H
            and dP.stride(1) == 1 and inp.stride(1) == 1 and Y.stride(1) == 1
            and dP.stride(0) % 8 == 0 and inp.stride(0) % 8 == 0 and Y.stride(0) % 8 == 0
            and dP.data_ptr() % 16 == 0 and inp.data_ptr() % 16 == 0 and Y.data_ptr() % 16 == 0)


def lstm_dw(dP, inp, Y, T, Bp, H, D, out=None):
    """(dW_ih (D*4H, nx), dW_hh (D, 4H, H)) fp32 from one pass over the bf16 gate gradients dP; out = zeroed (or
    to-be-accumulated-into) destination pair."""
    if out is not None:
        dwih, dwhh = out
        _chk(dwih, "dwih"); _chk(dwhh, "dwhh")
        assert dwih.shape == (D * 4 * H, inp.shape[1]) and dwhh.shape == (D, 4 * H, H)
    else:
        dwih = torch.zeros((D * 4 * H, inp.shape[1]), device=dP.device, dtype=torch.float32)
        dwhh = torch.zeros((D, 4 * H, H), device=dP.device, dtype=torch.float32)
    rc = _lib.lib().lob_lstm_dw_bf16(_ptr(dP), dP.stride(0), _ptr(inp), inp.stride(0), inp.shape[1], _ptr(Y),
                                     Y.stride(0), _ptr(dwih), _ptr(dwhh), T, Bp, H, D, _stream())
    _lib.check(rc, "lob_lstm_dw_bf16")
    return dwih, dwhh


#: bf16 x bf16 NT GEMMs with K >= 128, K % 32 == 0 go through the LDS-DMA kernel (pass bf16 weights)
NT_DMA = True


def dma_ok(K, N, M):
    """Shapes the LDS-DMA NT GEMM accepts (K = contraction, N = output columns, M = rows)."""
    return NT_DMA and K % 32 == 0 and K >= 128 and N % 128 == 0 and N <= 2048 and M % 256 == 0


def gate_ws_ok(K, H):
    """Shapes of the weight-stationary gate GEMM (csrc/gate_gemm_ws.hip): bf16 x bf16 -> bf16 fragment-order P at
    H == 128 (K = 128 / 256) and H == 256 (K = 256 / 512), any number of rows."""
    return PG_BF16 and ((H == 128 and K in (128, 256)) or (H == 256 and K in (256, 512)))


#: mixed mode, H == 128: store the fragment-order pre-activations / saved gates as bf16 (half the HBM
#: bytes of the two largest streams of the step).  Set to False to keep them fp32.
PG_BF16 = True


def bf16_rec(H, p16=True):
    """Hidden sizes with bf16-MFMA recurrent kernels: 128 (W_hh in registers; fp32 or bf16 P) and 256 (W_hh streamed
    from L2; bf16 P / saved gates only)."""
    return H == 128 or (H == 256 and bool(p16))


def gate_gemm_x(x, wih, bias, T, Bp, H, D, frag, mixed=False, range=None, exact=False):
    """P = x[T*Bp,K] @ wih[D*4H,K]^T + bias, fragment order when frag.  In mixed mode at H == 128 / 256 P is
    bf16 when ``ops.PG_BF16`` (it is only ever read by the bf16 recurrent kernel).
    fp32 path, H == 128: ``range`` (D + 1 floats on the device: max |W_ih| per direction, a bound on |x|; see
    ``images.build``) lets the fp16-split kernel choose its operand pre-scales; ``exact=True`` keeps the exact-fp32 MFMA
    kernel (activations without a known bound)."""
    _chk(range, "range")
    x16 = x.dtype == torch.bfloat16
    w16 = wih.dtype == torch.bfloat16
    _chk(x, "x", x.dtype if x16 else torch.float32); _chk(wih, "wih", wih.dtype if w16 else torch.float32)
    _chk(bias, "bias")
    K = x.shape[1]
    assert x.shape[0] == T * Bp and wih.shape == (D * 4 * H, K)
    if (mixed or x16) and not frag:
        return gemm_nt(x, wih, bias, mixed=True)
    if (mixed or x16) and _bf16_ok(x, K, K):
        p16 = PG_BF16 and H in (128, 256)
        P = torch.empty((T * Bp, D * 4 * H), device=x.device, dtype=torch.bfloat16 if p16 else torch.float32)
        rc = _lib.lib().lob_gate_gemm_x_bf16(_ptr(x), int(x16), K, _ptr(wih), int(w16), _ptr(bias), _ptr(P), int(p16),
                                             T, Bp, H, D, K, _stream())
        _lib.check(rc, "lob_gate_gemm_x_bf16")
        return P
    P = torch.empty((T * Bp, D * 4 * H), device=x.device, dtype=torch.float32)
    if x16:
        raise _lib.LobError("gate_gemm_x: bf16 input with a shape the bf16 kernel does not support")
    rc = _lib.lib().lob_gate_gemm_x_f32(_ptr(x), K, _ptr(wih), _ptr(bias), _ptr(P), T, Bp, H, D, K,
                                        (1 if frag else 0) | (2 if exact else 0), _ptr(range), _stream())
    _lib.check(rc, "lob_gate_gemm_x_f32")
    return P


def can_fuse_dropout(H, mixed):
    """The bf16-MFMA recurrent kernels (mixed mode, H == 128; H == 256 with bf16 P) can emit the dropped bf16 copy
    themselves."""
    return bool(mixed) and bf16_rec(H, PG_BF16)


OVERLAP_DW = True            # small batches: the weight-gradient GEMM of layer l on a second stream, next to the BPTT of layer
                             # l - 1 (whose workgroups cover at most half of the CUs there); False: everything on one stream


def rec_underfilled(H, Bp, D):
    """True where the H = 128 recurrent kernels of a training step leave at least half of the 256 CUs idle (16-row tiles, one
    workgroup per CU up to 128 tiles: B <= 1024 with two directions).  H = 256 is NOT in: its part-tile BPTT at B <= 512 also
    covers only half of the CUs, but what bounds it is the W_hh stream out of L2, and a GEMM next to it takes that bandwidth:
    measured, the BPTT ran exactly as much longer as the GEMM it overlapped took (10.75 ms with and without)."""
    return H == 128 and (Bp // 16) * D <= 128


FUSE_F32_DROPOUT = True      # fp32 path, H = 128, split kernels: inter-layer dropout in the recurrent forward's store and the
                             # dX GEMM's epilogue instead of four stand-alone passes per step (A/B and twin tests: False)


def can_fuse_dropout_f32(H, Bp, save):
    """The fp32 path's fp16-split saving forward (H == 128) can write dropout(Y) next to Y (lob_lstm_rec_fwd_f32_drop); its
    backward is the mask epilogue of lob_gemm_nt_f32_split, or the stand-alone kernel where that GEMM does not run."""
    return bool(FUSE_F32_DROPOUT) and bool(save) and H == 128 and Bp % 32 == 0 and _lib.get_variant("F32_SPLIT") != 0


def lstm_rec_fwd(P, whh, T, Bp, H, D, save, mixed=False, drop_p=0.0, seed=0, want_f32=True, want_bf16=False, nvalid=0,
                 range=None):
    """Runs the persistent recurrent kernel; returns (Y fp32 or None, Csave or None, Y16 or None, Yd or None).
    mixed: h W_hh^T on bf16 MFMA (H == 128), everything else fp32.  With can_fuse_dropout: want_bf16 adds
    Y16 = bf16(Y); drop_p > 0 adds Yd = bf16(dropout(Y)); want_f32=False skips the fp32 Y.
    nvalid: how many of the Bp rows carry windows (0 = all): lets a one-window inference call skip the padding rows.
    range (fp32 path, H == 128): D floats on the device, max |W_hh| per direction (the fp16-split kernel's pre-scale)."""
    _chk(range, "range")
    p16 = P.dtype == torch.bfloat16
    _chk(P, "P", P.dtype if p16 else torch.float32); _chk(whh, "whh")
    assert whh.shape == (D, 4 * H, H)
    dev = P.device
    c16 = bool(save) and c_bf16_ok(H, mixed, p16)
    Cs = torch.empty((D * T * Bp * H,), device=dev, dtype=torch.bfloat16 if c16 else torch.float32) if save else None
    Y = Y16 = Yd = None
    if mixed and bf16_rec(H, p16):
        # H == 256 streams the weights from L2: hand them over as bf16 in MFMA fragment order (include/lob.h), so
        # that every wave-load is 1 KB contiguous; one 1-MB permute per call
        whh16 = None
        if H == 256:       # [D, g 4, w 8, l31 32, ks 16, hi 2, j 8] -> [D, w, ks, g, hi, l31, j]
            whh16 = (whh.to(torch.bfloat16).reshape(D, 4, 8, 32, 16, 2, 8).permute(0, 2, 4, 1, 5, 3, 6).contiguous())
        if want_f32 or not want_bf16:
            Y = torch.empty((T * Bp, D * H), device=dev, dtype=torch.float32)
        if want_bf16:
            Y16 = torch.empty((T * Bp, D * H), device=dev, dtype=torch.bfloat16)
        if drop_p > 0:
            Yd = torch.empty((T * Bp, D * H), device=dev, dtype=torch.bfloat16)
        rc = _lib.lib().lob_lstm_rec_fwd_bf16(_ptr(P), int(p16), _ptr(whh), _ptr(whh16), _ptr(Y), _ptr(Cs), int(c16),
                                              _ptr(Y16), _ptr(Yd), float(drop_p), C.c_uint64(seed), T, Bp, H, D,
                                              1 if save else 0, int(nvalid), _stream())
    elif drop_p > 0:      # fp32 path, fp16-split kernel: Yd = dropout(Y) as fp32, the stand-alone kernel's mask
        assert not p16 and not want_bf16 and can_fuse_dropout_f32(H, Bp, save)
        Y = torch.empty((T * Bp, D * H), device=dev, dtype=torch.float32)
        Yd = torch.empty((T * Bp, D * H), device=dev, dtype=torch.float32)
        rc = _lib.lib().lob_lstm_rec_fwd_f32_drop(_ptr(P), _ptr(whh), _ptr(Y), _ptr(Yd), float(drop_p), C.c_uint64(seed),
                                                  _ptr(Cs), T, Bp, H, D, _ptr(range), _stream())
    else:
        assert not p16 and not want_bf16
        Y = torch.empty((T * Bp, D * H), device=dev, dtype=torch.float32)
        rc = _lib.lib().lob_lstm_rec_fwd_f32(_ptr(P), _ptr(whh), _ptr(Y), _ptr(Cs), T, Bp, H, D, 1 if save else 0,
                                             _ptr(range), _stream())
    _lib.check(rc, "lob_lstm_rec_fwd")
    return Y, Cs, Y16, Yd


def layernorm_act(x, gamma, beta, act=ACT_NONE, eps=1e-5, remap=None, drop_p=0.0, seed=0, out=None,
                  out_bf16=False):
    """LN(+act, +dropout) over the last axis of x[rows, width]; remap=(T, B, Bp) relays rows
    (b,t) -> t*Bp + b (out must then have T*Bp rows; pad rows are left untouched)."""
    x16 = x.dtype == torch.bfloat16
    _chk(x, "x", torch.bfloat16 if x16 else torch.float32); _chk(gamma, "gamma"); _chk(beta, "beta")
    if gamma is None:            # nn.Identity in place of the LayerNorm (09_sensitivity_analysis.py:190, 209)
        act = act | LN_IDENTITY
    rows, width = x.shape
    if x16:
        if not ln_x_bf16_ok(width) or remap is not None:
            raise _lib.LobError("layernorm_act: bf16 input rows are read at width 256 / 512 (no relayout) only")
        act = act | X_BF16
    out_bf16 = bool(out_bf16) and width in (128, 256, 512)
    odt = torch.bfloat16 if out_bf16 else torch.float32
    if remap is None:
        rT = rB = rBp = 0
        if out is None:
            out = torch.empty((rows, width), device=x.device, dtype=odt)
    else:
        rT, rB, rBp = remap
        if out is None:
            out = torch.zeros((rT * rBp, width), device=x.device, dtype=odt) if rBp != rB else \
                torch.empty((rT * rBp, width), device=x.device, dtype=odt)
    _chk(out, "out", odt)
    rc = _lib.lib().lob_layernorm_act_f32(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), int(out_bf16), rows, width,
                                          eps, act,
                                          rT, rB, rBp, float(drop_p), C.c_uint64(seed), _stream())
    _lib.check(rc, "lob_layernorm_act_f32")
    return out


#: mixed path, H == 128, C <= 64: Linear -> LayerNorm -> GELU -> Dropout of input_proj in ONE kernel (bit-identical to the
#: three-kernel sequence it replaces; tests switch it off to compare)
FUSE_INPUT_PROJ = True


def input_proj_ok(x2d, H, C):
    return bool(FUSE_INPUT_PROJ) and H in (128, 256) and 0 < C <= 64 and x2d.is_contiguous() and x2d.data_ptr() % 16 == 0


IP_COLWAVE = 0x4000      # include/lob.h LOB_IP_COLWAVE


def input_proj_ln(x2d, w, b, gamma, beta, B, T, Bp, H, act=ACT_NONE, eps=1e-5, drop_p=0.0, seed=0, save=False, colwave=False):
    """input_proj (04_lstm_model.py:173-178) of the mixed path in one launch: x2d fp32 [B*T, C] rows (b,t) -> bf16
    activations [T*Bp, H] time-major.  save: also returns the fp32 pre-activations [B*T, H] and the bf16 padded windows
    [B*T, Cp] that the backward reads.  Returns (a, pre | None, xb | None)."""
    _chk(x2d, "x"); _chk(w, "w"); _chk(b, "b"); _chk(gamma, "gamma"); _chk(beta, "beta")
    rows, Cc = x2d.shape
    assert rows == B * T and w.shape == (H, Cc)
    if gamma is None:
        act = act | LN_IDENTITY
    if colwave and H == 128:        # H = 128: the column-decomposed kernel (H = 256 always runs it) as the test twin
        act = act | IP_COLWAVE
    dev = x2d.device
    a = (torch.zeros if Bp != B else torch.empty)((T * Bp, H), device=dev, dtype=torch.bfloat16)
    Cp = (Cc + 7) // 8 * 8
    pre = torch.empty((rows, H), device=dev, dtype=torch.float32) if save else None
    xb = torch.empty((rows, Cp), device=dev, dtype=torch.bfloat16) if save else None
    rc = _lib.lib().lob_input_proj_ln_bf16(_ptr(x2d), Cc, _ptr(w), w.stride(0), _ptr(b), _ptr(gamma), _ptr(beta), _ptr(pre),
                                           _ptr(xb), Cp, _ptr(a), B, T, Bp, H, eps, act, float(drop_p), C.c_uint64(seed),
                                           _stream())
    _lib.check(rc, "lob_input_proj_ln_bf16")
    return a, pre, xb


#: fp32 path, H == 128: input_proj (Linear + LayerNorm + GELU + dropout) in ONE kernel; tests switch it off to compare
FUSE_INPUT_PROJ_F32 = True


def input_proj_f32_ok(x2d, H, C, w):
    return (bool(FUSE_INPUT_PROJ_F32) and H == 128 and 0 < C <= 64 and x2d.dtype == torch.float32 and x2d.is_contiguous()
            and x2d.data_ptr() % 16 == 0 and w.stride(1) == 1)


def input_proj_ln_f32(x2d, w, b, gamma, beta, B, T, Bp, H, act=ACT_NONE, eps=1e-5, drop_p=0.0, seed=0, save=False):
    """fp32 twin of input_proj_ln: x2d fp32 [B*T, C] rows (b,t) -> fp32 activations [T*Bp, H] time-major (+ the fp32
    pre-activations [B*T, H] with save).  Returns (a, pre | None)."""
    _chk(x2d, "x"); _chk(w, "w"); _chk(b, "b"); _chk(gamma, "gamma"); _chk(beta, "beta")
    rows, Cc = x2d.shape
    assert rows == B * T and w.shape == (H, Cc)
    if gamma is None:
        act = act | LN_IDENTITY
    dev = x2d.device
    a = (torch.zeros if Bp != B else torch.empty)((T * Bp, H), device=dev, dtype=torch.float32)
    pre = torch.empty((rows, H), device=dev, dtype=torch.float32) if save else None
    rc = _lib.lib().lob_input_proj_ln_f32(_ptr(x2d), Cc, _ptr(w), w.stride(0), _ptr(b), _ptr(gamma), _ptr(beta), _ptr(pre),
                                          _ptr(a), B, T, Bp, H, eps, act, float(drop_p), C.c_uint64(seed), _stream())
    _lib.check(rc, "lob_input_proj_ln_f32")
    return a, pre


def dropout(x, p, seed, out=None):
    _chk(x, "x")
    if out is None:
        out = torch.empty_like(x)
    rc = _lib.lib().lob_dropout_f32(_ptr(x), _ptr(out), x.numel(), float(p), C.c_uint64(seed), _stream())
    _lib.check(rc, "lob_dropout_f32")
    return out


def attn_pool_fwd(v, u, w2, b2, T, B, Bp):
    """u=None: uniform weights 1/T (mean pooling over time, 09_sensitivity_analysis.py:236)."""
    v16 = v.dtype == torch.bfloat16
    _chk(v, "v", v.dtype if v16 else torch.float32); _chk(u, "u"); _chk(w2, "w2"); _chk(b2, "b2")
    W, W2 = v.shape[1], (u.shape[1] if u is not None else 0)
    ctx = torch.empty((B, W), device=v.device, dtype=torch.float32)
    attn = torch.empty((B, T), device=v.device, dtype=torch.float32)
    rc = _lib.lib().lob_attn_pool_fwd_f32(_ptr(v), int(v16), _ptr(u), _ptr(w2), _ptr(b2), _ptr(ctx), _ptr(attn),
                                          T, B, Bp, W, W2, _stream())
    _lib.check(rc, "lob_attn_pool_fwd_f32")
    return ctx, attn


#: mixed path, H == 128 bidirectional: post-LSTM LayerNorm + the attention's score layer in ONE kernel (bit-identical to the
#: LayerNorm kernel + the K = 256 GEMM + the pooling kernel's score sums; tests switch it off to compare)
FUSE_ATTN_SCORES = True


def attn_scores_ok(y16, H, D, Bp, w1):
    return (bool(FUSE_ATTN_SCORES) and H in (128, 256) and D == 2 and Bp % 32 == 0 and y16.dtype == torch.bfloat16
            and y16.is_contiguous() and w1 is not None and tuple(w1.shape) == (H, 2 * H))


def attn_scores(y16, gamma, beta, w1_16, b1, w2, b2, T, B, Bp, H, D, eps=1e-5, save=False):
    """v = LN(y16) (bf16 [T*Bp, 256]), u = tanh(W1 v + b1) (fp32 [T*Bp, 128], only with save), scores S [B, T] fp32."""
    _chk(y16, "y16", torch.bfloat16); _chk(gamma, "gamma"); _chk(beta, "beta"); _chk(w1_16, "w1", torch.bfloat16)
    _chk(b1, "b1"); _chk(w2, "w2"); _chk(b2, "b2")
    dev = y16.device
    v = torch.empty((T * Bp, 2 * H), device=dev, dtype=torch.bfloat16)
    u = torch.empty((T * Bp, H), device=dev, dtype=torch.float32) if save else None
    S = torch.empty((B, T), device=dev, dtype=torch.float32)
    rc = _lib.lib().lob_attn_scores_bf16(_ptr(y16), _ptr(gamma), _ptr(beta), _ptr(w1_16), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(v),
                                         _ptr(u), _ptr(S), T, B, Bp, H, D, eps, _stream())
    _lib.check(rc, "lob_attn_scores_bf16")
    return v, u, S


#: fp32 path, H == 128 bidirectional: LayerNorm + score layer in ONE kernel (v bit-identical; u / scores on the fp16-split
#: arithmetic of the fp32 path's gate GEMMs instead of the exact-fp32 MFMA GEMM); tests switch it off to compare
FUSE_ATTN_SCORES_F32 = True


def attn_scores_f32_ok(y, H, D, Bp, w1):
    return (bool(FUSE_ATTN_SCORES_F32) and H == 128 and D == 2 and Bp % 32 == 0 and y.dtype == torch.float32
            and y.is_contiguous() and w1 is not None and w1.dtype == torch.float32 and tuple(w1.shape) == (H, 2 * H)
            and w1.is_contiguous() and _lib.get_variant("F32_SPLIT") != 0)


def attn_scores_f32(y, gamma, beta, w1, b1, w2, b2, T, B, Bp, H, D, eps=1e-5, save=False):
    """fp32 twin of attn_scores: v = LN(y) (fp32 [T*Bp, 256]), u = tanh(W1 v + b1) (fp32 [T*Bp, 128], only with save),
    scores S [B, T]."""
    _chk(y, "y"); _chk(gamma, "gamma"); _chk(beta, "beta"); _chk(w1, "w1"); _chk(b1, "b1"); _chk(w2, "w2"); _chk(b2, "b2")
    dev = y.device
    v = torch.empty((T * Bp, 2 * H), device=dev, dtype=torch.float32)
    u = torch.empty((T * Bp, H), device=dev, dtype=torch.float32) if save else None
    S = torch.empty((B, T), device=dev, dtype=torch.float32)
    rc = _lib.lib().lob_attn_scores_f32(_ptr(y), _ptr(gamma), _ptr(beta), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(v),
                                        _ptr(u), _ptr(S), T, B, Bp, H, D, eps, _stream())
    _lib.check(rc, "lob_attn_scores_f32")
    return v, u, S


def attn_pool_fwd_scores(v, S, T, B, Bp):
    """Softmax over time of finished scores S [B, T] and the context sums (the second half of attn_pool_fwd)."""
    v16 = v.dtype == torch.bfloat16
    _chk(v, "v", torch.bfloat16 if v16 else torch.float32); _chk(S, "S")
    W = v.shape[1]
    ctx = torch.empty((B, W), device=v.device, dtype=torch.float32)
    attn = torch.empty((B, T), device=v.device, dtype=torch.float32)
    rc = _lib.lib().lob_attn_pool_fwd_f32(_ptr(v), int(v16), _ptr(S), None, None, _ptr(ctx), _ptr(attn), T, B, Bp, W, 0, _stream())
    _lib.check(rc, "lob_attn_pool_fwd_f32")
    return ctx, attn


#: mixed path, H == 128 bidirectional: dV = dU W1 and the post-LSTM LayerNorm's backward in ONE kernel (dx bit-identical to
#: the GEMM + LayerNorm-backward pair; tests switch it off to compare)
FUSE_ATTN_LN_BWD = True


def attn_ln_bwd_ok(ylast, dU, w1t, H, D, Bp, gamma):
    bf = torch.bfloat16
    return (bool(FUSE_ATTN_LN_BWD) and H in (128, 256) and D == 2 and Bp % 32 == 0 and gamma is not None
            and ylast.dtype == bf and dU.dtype == bf and w1t is not None and w1t.dtype == bf
            and tuple(w1t.shape) == (2 * H, H) and ylast.is_contiguous() and dU.is_contiguous() and w1t.is_contiguous())


def attn_ln_bwd(ylast, gamma, beta, dU, w1t, attn, dctx, T, B, Bp, H, D, eps=1e-5, dg=None, db=None):
    """dY (bf16 [T*Bp, 256]) = LayerNorm backward of (dU @ W1 + attn[b][t] * dctx[b]); dgamma / dbeta accumulated."""
    _chk(ylast, "ylast", torch.bfloat16); _chk(dU, "dU", torch.bfloat16); _chk(w1t, "w1t", torch.bfloat16)
    _chk(gamma, "gamma"); _chk(beta, "beta"); _chk(attn, "attn"); _chk(dctx, "dctx")
    dg = torch.zeros_like(gamma) if dg is None else dg
    db = torch.zeros_like(beta) if db is None else db
    dx = torch.empty_like(ylast)
    rc = _lib.lib().lob_attn_ln_bwd_bf16(_ptr(ylast), _ptr(gamma), _ptr(beta), _ptr(dU), _ptr(w1t), _ptr(dx), _ptr(dg), _ptr(db),
                                         _ptr(attn), _ptr(dctx), T, B, Bp, H, D, eps, _stream())
    _lib.check(rc, "lob_attn_ln_bwd_bf16")
    return dx, dg, db


#: mixed path, H == 128: backward of input_proj's LayerNorm + GELU + dropout AND the Linear's weight gradient in ONE kernel
#: (dpre never goes to HBM); equal to the LayerNorm backward + TN GEMM pair up to fp32 summation order.  OFF by default:
#: measured 0.07 ms per step SLOWER than the pair (same-box A/B: 14.30 against 14.23 ms) -- the LayerNorm backward with GELU'
#: and the dropout hash is VALU-bound, and next to 128 accumulator registers only two waves per SIMD are left to hide it
FUSE_INPUT_PROJ_BWD = False


def input_proj_bwd_ok(pre, dA, xb, H):
    bf = torch.bfloat16
    return (bool(FUSE_INPUT_PROJ_BWD) and H == 128 and pre.dtype == torch.float32 and pre.shape[1] == 128 and dA.dtype == bf
            and xb is not None and xb.dtype == bf and xb.shape[1] <= 64 and xb.shape[1] % 8 == 0
            and pre.is_contiguous() and dA.is_contiguous() and xb.is_contiguous())


def input_proj_bwd(pre, gamma, beta, dA, xb, dW, B, T, Bp, H, act=ACT_NONE, eps=1e-5, drop_p=0.0, seed=0, dg=None, db=None,
                   dbias=None):
    """dW [128, Cp] += dpre^T xb, dgamma / dbeta / dbias accumulated, where dpre = backward of LayerNorm + act + dropout of
    input_proj applied to dA (bf16, time-major); pre: the LayerNorm's fp32 input rows (b,t)."""
    _chk(pre, "pre"); _chk(dA, "dA", torch.bfloat16); _chk(xb, "xb", torch.bfloat16); _chk(dW, "dW")
    _chk(gamma, "gamma"); _chk(beta, "beta")
    if gamma is None:
        act = act | LN_IDENTITY
        dg = db = None
    else:
        dg = torch.zeros_like(gamma) if dg is None else dg
        db = torch.zeros_like(beta) if db is None else db
    rc = _lib.lib().lob_input_proj_bwd_bf16(_ptr(pre), _ptr(gamma), _ptr(beta), _ptr(dA), _ptr(xb), xb.shape[1], _ptr(dW),
                                            dW.stride(0), _ptr(dg), _ptr(db), _ptr(dbias), B, T, Bp, H, eps, act,
                                            float(drop_p), C.c_uint64(seed), _stream())
    _lib.check(rc, "lob_input_proj_bwd_bf16")
    return dW, dg, db


def softmax_rows(x):
    _chk(x, "x")
    out = torch.empty_like(x)
    rc = _lib.lib().lob_softmax_rows_f32(_ptr(x), _ptr(out), x.shape[0], x.shape[1], _stream())
    _lib.check(rc, "lob_softmax_rows_f32")
    return out


def prob_to_state(probs):
    """y0 (B,3) f64 = prob_to_ode_state(P(closed)) (08_forecasting.py:215-234)."""
    _chk(probs, "probs")
    y0 = torch.empty((probs.shape[0], 3), device=probs.device, dtype=torch.float64)
    rc = _lib.lib().lob_prob_to_state_f64(_ptr(probs), _ptr(y0), probs.shape[0], _stream())
    _lib.check(rc, "lob_prob_to_state_f64")
    return y0


def ode_rk4(base_rates, n_points, t0, t1, substeps, *, probs=None, alpha=0.0, y0=None,
            want_traj=True, want_final=False, want_pred=True, raw=False):
    """Batched ODE solve.  probs (B,2) f32 -> coupled mode; y0 (B,3) f64 -> plain solve."""
    src = probs if probs is not None else y0
    B = src.shape[0]
    dev = src.device
    if probs is not None:
        _chk(probs, "probs")
    else:
        _chk(y0, "y0", torch.float64)
    traj = torch.empty((B, n_points, 3), device=dev, dtype=torch.float64) if want_traj else None
    final = torch.empty((B, 3), device=dev, dtype=torch.float64) if want_final else None
    pred = torch.empty((B,), device=dev, dtype=torch.int64) if want_pred else None
    rates = (C.c_double * 6)(*[float(r) for r in base_rates])
    rc = _lib.lib().lob_ode_rk4_f64(_ptr(probs), _ptr(y0), rates, float(alpha), int(n_points), float(t0),
                                    float(t1), int(substeps), _ptr(traj), _ptr(final), _ptr(pred), B, 1 if raw else 0,
                                    _stream())
    _lib.check(rc, "lob_ode_rk4_f64")
    return traj, final, pred


# ---------------------------------------------------------------------------------------------
# backward-side wrappers
# ---------------------------------------------------------------------------------------------
def f32_split_bwd_ok(H, Bp):
    """The fp32 path's BPTT on the fp16-split arithmetic (lob_lstm_rec_bwd_f32_x)."""
    return H == 128 and Bp % 32 == 0 and _lib.get_variant("F32_SPLIT") != 0


def lstm_rec_bwd(G, Cs, whh, dY, T, Bp, H, D, dp_bf16=False, dbias=None, dbias2=None, amax_out=None, range=None):
    """BPTT through one layer; returns (dP[T*Bp, D*4H] row-major fp32|bf16, dbias[D*4H]); dbias: zeroed destination.
    dbias2 (mixed kernels only, see ``rec_bwd_two_bias_ok``): a second destination that receives the same adds.
    fp32 saved gates at H == 128 (``f32_split_bwd_ok``): the fp16-split kernel; amax_out (a zeroed device float) then
    receives max|dP|, range = D device floats max|W_hh| per direction."""
    g16 = G.dtype == torch.bfloat16
    dy16 = dY.dtype == torch.bfloat16
    c16 = Cs.dtype == torch.bfloat16
    _chk(G, "G", G.dtype if g16 else torch.float32); _chk(Cs, "Csave", torch.bfloat16 if c16 else torch.float32)
    _chk(whh, "whh"); _chk(dY, "dY", torch.bfloat16 if dy16 else torch.float32)
    assert dY.shape == (T * Bp, D * H) and (not g16 or (dp_bf16 and bf16_rec(H, g16)))
    if (dy16 or c16) and not (dp_bf16 and bf16_rec(H, g16) and _rec16(H)):
        raise _lib.LobError("lstm_rec_bwd: bf16 dY / cell states are read by the bf16 BPTT kernels (H = 128 / 256) only")
    dP = torch.empty((T * Bp, D * 4 * H), device=G.device, dtype=torch.bfloat16 if dp_bf16 else torch.float32)
    fused_bias = uses_frag(H)
    if dbias is None:
        dbias = torch.zeros((D * 4 * H,), device=G.device, dtype=torch.float32)
    _chk(dbias, "dbias"); _chk(dbias2, "dbias2")
    if dbias2 is not None and not (dp_bf16 and bf16_rec(H, g16)):
        raise _lib.LobError("lstm_rec_bwd: a second bias-gradient destination is taken by the bf16 BPTT kernels only")
    if dp_bf16 and bf16_rec(H, g16):
        whht16 = None
        if H == 256:       # [D, ks 64, hi 2, j 8, w 8, l31 32] -> [D, w, ks, hi, l31, j]  (fragment order, lob.h)
            whht16 = (whh.to(torch.bfloat16).reshape(D, 64, 2, 8, 8, 32).permute(0, 4, 1, 2, 5, 3).contiguous())
        rc = _lib.lib().lob_lstm_rec_bwd_bf16(_ptr(G), int(g16), _ptr(Cs), int(c16), _ptr(whh), _ptr(whht16), _ptr(dY),
                                              int(dy16), _ptr(dP), _ptr(dbias), _ptr(dbias2), T, Bp, H, D, _stream())
    elif not g16 and not dy16 and not c16 and f32_split_bwd_ok(H, Bp):
        _chk(amax_out, "amax_out"); _chk(range, "range")
        rc = _lib.lib().lob_lstm_rec_bwd_f32_x(_ptr(G), _ptr(Cs), _ptr(whh), _ptr(dY), _ptr(dP), int(dp_bf16), _ptr(dbias),
                                               _ptr(amax_out), _ptr(range), T, Bp, H, D, _stream())
    else:
        if amax_out is not None:
            raise _lib.LobError("lstm_rec_bwd: max|dP| is produced by the fp16-split fp32 kernel only (H = 128, LOB_VAR_F32_SPLIT)")
        rc = _lib.lib().lob_lstm_rec_bwd_f32(_ptr(G), _ptr(Cs), _ptr(whh), _ptr(dY), _ptr(dP), int(dp_bf16),
                                             _ptr(dbias) if fused_bias else _ptr(None), T, Bp, H, D, _stream())
    _lib.check(rc, "lob_lstm_rec_bwd")
    if not fused_bias:
        colsum(dP, dbias)
    return dP, dbias


def colsum(a, out=None):
    """out[N] += column sums of a[M,N] (a may be a column slice of a wider row-major tensor)."""
    M, N = a.shape
    assert a.stride(1) == 1
    if out is None:
        out = torch.zeros((N,), device=a.device, dtype=torch.float32)
    fn = _lib.lib().lob_colsum_bf16 if a.dtype == torch.bfloat16 else _lib.lib().lob_colsum_f32
    rc = fn(_ptr(a), a.stride(0), M, N, _ptr(out), _stream())
    _lib.check(rc, "lob_colsum")
    return out


def act(x, kind):
    _chk(x, "x")
    out = torch.empty_like(x)
    rc = _lib.lib().lob_act_f32(_ptr(x), _ptr(out), x.numel(), kind, _stream())
    _lib.check(rc, "lob_act_f32")
    return out


def act_bwd(dy, pre, kind):
    _chk(dy, "dy"); _chk(pre, "pre")
    dx = torch.empty_like(pre)
    rc = _lib.lib().lob_act_bwd_f32(_ptr(dy), _ptr(pre), _ptr(dx), pre.numel(), kind, _stream())
    _lib.check(rc, "lob_act_bwd_f32")
    return dx


def layernorm_act_bwd(x, gamma, beta, dy, act=ACT_NONE, eps=1e-5, remap=None, drop_p=0.0, seed=0, pool=None,
                      dx_colsum=None, dx_bf16=False, dg=None, db=None):
    """Returns (dx [rows,width] in INPUT row order, dgamma, dbeta).  pool=(attn [B,T], dctx [B,width], T, B, Bp)
    adds attn[b][t] * dctx[b] to dy on the fly (context path of the attention pooling).  dx_colsum [width]
    (widths 128/256/512): += column sums of dx (see ``can_fuse_colsum``).  dy may be bf16 and dx_bf16 stores dx as
    bf16 (widths 128 / 256: the mixed path's gradient carries, ``DY_BF16_CARRY``)."""
    dy16 = dy.dtype == torch.bfloat16
    x16 = x.dtype == torch.bfloat16
    _chk(x, "x", torch.bfloat16 if x16 else torch.float32); _chk(gamma, "gamma"); _chk(beta, "beta")
    _chk(dy, "dy", torch.bfloat16 if dy16 else torch.float32)
    rows, width = x.shape
    if x16:
        if not (ln_x_bf16_ok(width) and dy16 and dx_bf16):
            raise _lib.LobError("layernorm_act_bwd: bf16 input rows need width 256 / 512 and bf16 dy / dx")
        act = act | X_BF16
    rT, rB, rBp = (0, 0, 0) if remap is None else remap
    dx = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16 if dx_bf16 else torch.float32)
    if dy16:
        act = act | DY_BF16
    if dx_bf16:
        act = act | OUT_BF16
    if gamma is None:            # identity "LayerNorm": dx = dy * act' * mask, no affine gradients
        act = act | LN_IDENTITY
        dg = db = None
    else:
        dg = torch.zeros_like(gamma) if dg is None else dg           # accumulation targets (fp32 atomics)
        db = torch.zeros_like(beta) if db is None else db
        _chk(dg, "dgamma"); _chk(db, "dbeta")
    pa, pd, pT, pB, pBp = (None, None, 0, 0, 0) if pool is None else pool
    rc = _lib.lib().lob_layernorm_act_bwd_f32(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(dy), _ptr(dx), _ptr(dg), _ptr(db),
                                              rows, width, eps, act, rT, rB, rBp, float(drop_p), C.c_uint64(seed),
                                              _ptr(pa), _ptr(pd), pT, pB, pBp, _ptr(dx_colsum), _stream())
    _lib.check(rc, "lob_layernorm_act_bwd_f32")
    return dx, dg, db


#: mixed path: the input projection's LayerNorm backward hands dpre to the weight-gradient GEMM as bf16 (that GEMM rounds it
#: to bf16 anyway; the bias gradient is summed from the fp32 values inside the LayerNorm backward)
DPRE_BF16 = True

#: mixed path: the last LSTM layer hands its output to the LayerNorm as bf16 only (no fp32 copy of Y is written or read).
LN_X_BF16 = True


def ln_x_bf16_ok(width):
    """Widths at which the LayerNorm kernels read bf16 input rows (the post-LSTM LayerNorm at H = 128 / 256, D = 2)."""
    return width in (256, 512)


def can_fuse_colsum(width):
    """The vectorised LayerNorm backward (these widths) can emit the column sums of dx itself."""
    return width in (128, 256, 512)


def attn_pool_bwd(v, u, attn, dctx, w2, T, B, Bp, want_dv=True, du_bf16=False, du_colsum=None, dw2=None, dattn=None):
    """Returns (dV [T*Bp,W] fp32 or None, dPreU [T*Bp,W2] fp32|bf16, dw2 [W2]); pad rows are zero.
    want_dv=False: the direct term a[t]*dctx is left to layernorm_act_bwd(pool=...).  dattn [B,T]: gradient w.r.t. the
    weights themselves (stand-alone Attention module)."""
    _chk(dattn, "dattn")
    v16 = v.dtype == torch.bfloat16
    _chk(v, "v", v.dtype if v16 else torch.float32); _chk(u, "u"); _chk(attn, "attn"); _chk(dctx, "dctx"); _chk(w2, "w2")
    W = v.shape[1]
    alloc = torch.zeros if Bp != B else torch.empty
    if u is None:                # mean pooling: dV = dctx / T only
        dV = alloc((T * Bp, W), device=v.device, dtype=torch.float32)
        rc = _lib.lib().lob_attn_pool_bwd_f32(_ptr(v), int(v16), None, _ptr(attn), _ptr(dctx), None, _ptr(dV), None, 0,
                                              None, None, T, B, Bp, W, 0, None, _stream())
        _lib.check(rc, "lob_attn_pool_bwd_f32")
        return dV, None, None
    W2 = u.shape[1]
    dV = alloc((T * Bp, W), device=v.device, dtype=torch.float32) if want_dv else None
    dU = alloc((T * Bp, W2), device=v.device, dtype=torch.bfloat16 if du_bf16 else torch.float32)
    if dw2 is None:
        dw2 = torch.zeros((W2,), device=v.device, dtype=torch.float32)
    rc = _lib.lib().lob_attn_pool_bwd_f32(_ptr(v), int(v16), _ptr(u), _ptr(attn), _ptr(dctx), _ptr(w2), _ptr(dV),
                                          _ptr(dU), int(du_bf16), _ptr(dw2), _ptr(du_colsum), T, B, Bp, W, W2, _ptr(dattn),
                                          _stream())
    _lib.check(rc, "lob_attn_pool_bwd_f32")
    return dV, dU, dw2


def attn_bwd_fuses_colsum(v, u, want_dv, du_bf16):
    """Shapes for which lob_attn_pool_bwd_f32 can accumulate the column sums of dPreU itself."""
    return (v.dtype == torch.bfloat16 and du_bf16 and not want_dv and u is not None and
            (v.shape[1], u.shape[1]) in ((256, 128), (512, 256)))


# ---------------------------------------------------------------------------------------------
# training-step and attribution wrappers (csrc/train_step.hip)
# ---------------------------------------------------------------------------------------------
def weighted_ce(logits, target, class_weight=None, scale=1.0, want_grad=True):
    """nn.CrossEntropyLoss(weight) 'mean' (04_lstm_model.py:435): returns (loss[1], dlogits*scale or None,
    correct[1] int32), all on the device (no sync)."""
    _chk(logits, "logits"); _chk(target, "target", torch.int64); _chk(class_weight, "class_weight")
    B, Cn = logits.shape
    assert target.shape == (B,) and (class_weight is None or class_weight.shape == (Cn,))
    loss = torch.empty((1,), device=logits.device, dtype=torch.float32)
    dl = torch.empty_like(logits) if want_grad else None
    correct = torch.empty((1,), device=logits.device, dtype=torch.int32)
    rc = _lib.lib().lob_weighted_ce_f32(_ptr(logits), _ptr(target), _ptr(class_weight), _ptr(loss), _ptr(dl),
                                        _ptr(correct), B, Cn, float(scale), _stream())
    _lib.check(rc, "lob_weighted_ce_f32")
    return loss, dl, correct


def sumsq(x, out=None, scratch=None):
    """out[0] += sum x^2 over a flat fp32 buffer.  scratch (512 floats): bit-reproducible summation order."""
    _chk(x, "x"); _chk(scratch, "scratch")
    if out is None:
        out = torch.zeros((1,), device=x.device, dtype=torch.float32)
    assert scratch is None or scratch.numel() >= 512
    rc = _lib.lib().lob_sumsq_f32(_ptr(x), x.numel(), _ptr(out), _ptr(scratch), _stream())
    _lib.check(rc, "lob_sumsq_f32")
    return out


def clip_scale_(g, normsq, max_norm):
    _chk(g, "g"); _chk(normsq, "normsq")
    rc = _lib.lib().lob_clip_scale_f32(_ptr(g), g.numel(), _ptr(normsq), float(max_norm), _stream())
    _lib.check(rc, "lob_clip_scale_f32")
    return g


def adamw_(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, normsq=None, max_norm=1.0,
           grad_scale=1.0):
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, n)
    _chk(normsq, "normsq")
    assert p.numel() == g.numel() == m.numel() == v.numel()
    rc = _lib.lib().lob_adamw_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(betas[0]),
                                  float(betas[1]), float(eps), float(weight_decay), int(step), _ptr(normsq),
                                  float(max_norm), float(grad_scale), _stream())
    _lib.check(rc, "lob_adamw_f32")


def abs_colsum(gx2d, out, scale=1.0):
    """out[C] += scale * sum_rows |gx2d[row, :]|."""
    _chk(gx2d, "gx"); _chk(out, "out")
    rows, Cn = gx2d.shape
    assert out.shape == (Cn,)
    rc = _lib.lib().lob_abs_colsum_f32(_ptr(gx2d), rows, Cn, float(scale), _ptr(out), _stream())
    _lib.check(rc, "lob_abs_colsum_f32")
    return out


def pad_cast_bf16(x, Cp):
    """bf16 copy of x[rows, C] with the columns zero-padded to Cp (a multiple of 8)."""
    _chk(x, "x")
    rows, Cn = x.shape
    out = torch.empty((rows, Cp), device=x.device, dtype=torch.bfloat16)
    rc = _lib.lib().lob_pad_cast_bf16(_ptr(x), _ptr(out), rows, Cn, Cp, _stream())
    _lib.check(rc, "lob_pad_cast_bf16")
    return out
