"""Backward pipeline of the drop-in model (the mirror of ``autograd._forward_impl``).

Produces the gradient of every ``state_dict`` tensor and of the input window batch
(04_lstm_model.py:490 ``loss.backward()``; 07_explainability.py:242-257 reads ``X.grad``).
All arithmetic is HIP (liblob.so); torch only slices, transposes the small weight matrices
and allocates.
"""
from __future__ import annotations

import torch

from . import ops
from .autograd import _seed
from .ops import ACT_GELU, ceil32


def _t(w):
    return w.t().contiguous()


def _linear_bwd(dy, x, w, need_dx=True, need_w=True, db=None):
    """y = x w^T + b  ->  (dx, dw, db); db may be handed in when a producer of dy already summed its columns."""
    dw = None
    if need_w:
        dw = torch.zeros_like(w)
        ops.gemm_tn(dy, x, dw)
        if db is None:
            db = ops.colsum(dy)
    dx = ops.gemm_nt(dy, _t(w)) if need_dx else None
    return dx, dw, db


def backward_impl(sv, ps, cfg, x_shape, dlogits, needs_input_grad):
    if sv is None:
        raise RuntimeError("backward called on a forward that ran without grad tracking")
    L, D, H, (p_in, p_lstm, p_cls), seed, mixed = cfg
    B, T, C = x_shape
    Bp = ceil32(B)
    n = len(ps)
    g = [None] * n
    # torch.autograd.grad(outputs, inputs=[x]) (the attribution path, 07:239-263) asks for the input gradient
    # only: the weight-gradient GEMMs (a third of the backward) are skipped then
    need_w = any(needs_input_grad[2:])
    i_ln, i_a0w, i_a0b, i_a2w, i_a2b = n - 12, n - 10, n - 9, n - 8, n - 7
    i_c0w, i_c3w, i_c6w = n - 6, n - 4, n - 2

    # ---- classifier (04:196-204)
    dz2d, g[i_c6w], g[i_c6w + 1] = _linear_bwd(dlogits, sv["z2d"], ps[i_c6w], need_w=need_w)
    dz2 = ops.dropout(dz2d, p_cls, _seed(seed, 21)) if p_cls > 0 else dz2d
    dz2p = ops.act_bwd(dz2, sv["z2p"], ACT_GELU)
    dz1d, g[i_c3w], g[i_c3w + 1] = _linear_bwd(dz2p, sv["z1d"], ps[i_c3w], need_w=need_w)
    dz1 = ops.dropout(dz1d, p_cls, _seed(seed, 20)) if p_cls > 0 else dz1d
    dz1p = ops.act_bwd(dz1, sv["z1p"], ACT_GELU)
    dctx, g[i_c0w], g[i_c0w + 1] = _linear_bwd(dz1p, sv["ctx"], ps[i_c0w], need_w=need_w)

    # ---- attention pooling (04:123-128)
    v, u = sv["v"], sv["u"]
    a0w = ps[i_a0w]
    if a0w is None:          # mean pooling (09:236): dV = dctx / T, no parameters
        dV, _, _ = ops.attn_pool_bwd(v, None, sv["attn"], dctx, None, T, B, Bp)
        pool = None
    else:
        fused = v.dtype == torch.bfloat16        # mixed mode: bf16 v / dU, context path folded into the LN backward
        # the score MLP's first-bias gradient = column sums of dU: emitted by the pooling backward itself where it can
        cs_fused = need_w and ops.attn_bwd_fuses_colsum(v, u, not fused, fused)
        du_cs = torch.zeros((u.shape[1],), device=u.device, dtype=torch.float32) if cs_fused else None
        dV, dU, dw2 = ops.attn_pool_bwd(v, u, sv["attn"], dctx, ps[i_a2w].reshape(-1), T, B, Bp,
                                        want_dv=not fused, du_bf16=fused, du_colsum=du_cs)
        g[i_a2w] = dw2.reshape(1, -1)
        g[i_a2b] = torch.zeros_like(ps[i_a2b])            # b2 cancels in the softmax: exactly 0
        if need_w:
            g[i_a0w] = torch.zeros_like(a0w)
            ops.gemm_tn(dU, v, g[i_a0w], mixed=mixed)
            g[i_a0b] = du_cs if cs_fused else ops.colsum(dU)
        if fused:
            w1t = _t(a0w)
            if ops.dma_ok(dU.shape[1], w1t.shape[0], dU.shape[0]):
                w1t = w1t.to(torch.bfloat16)
            dV = ops.gemm_nt(dU, w1t, mixed=mixed)                       # dU W1; + a[t] dctx is added below
            pool = (sv["attn"], dctx, T, B, Bp)
        else:
            ops.gemm_nt(dU, _t(a0w), out=dV, accumulate=True, mixed=mixed)
            pool = None

    # ---- post-LSTM LayerNorm (04:212)
    dY, g[i_ln], g[i_ln + 1] = ops.layernorm_act_bwd(sv["ylast"], ps[i_ln], ps[i_ln + 1], dV, pool=pool)

    # ---- LSTM stack, last layer first (04:211)
    for layer in reversed(range(L)):
        lay = sv["layers"][layer]
        if layer + 1 < L and p_lstm > 0 and not lay["fused_drop"]:
            dY = ops.dropout(dY, p_lstm, _seed(seed, 10 + layer))
        dP, dbias = ops.lstm_rec_bwd(lay["G"], lay["C"], lay["whh"], dY, T, Bp, H, D, dp_bf16=mixed)
        inp, Y, wih = lay["inp"], lay["Y"], lay["wih"]
        base = 4 + layer * 4 * D
        fused_dw = need_w and ops.can_fuse_dw(dP, inp, Y, T, Bp, H, D)
        if fused_dw:
            dwih, dwhh_all = ops.lstm_dw(dP, inp, Y, T, Bp, H, D)
        elif need_w:
            dwih = torch.zeros_like(wih)
            ops.gemm_tn(dP, inp, dwih, mixed=mixed)
        for d in range(D if need_w else 0):
            dwhh = dwhh_all[d] if fused_dw else torch.zeros_like(ps[base + 4 * d + 1])
            if T > 1 and not fused_dw:
                a_sl = dP[:, d * 4 * H:(d + 1) * 4 * H]
                y_sl = Y[:, d * H:(d + 1) * H]
                if d == 0:      # h_prev(t) = h(t-1)
                    ops.gemm_tn(a_sl[Bp:], y_sl[:(T - 1) * Bp], dwhh, mixed=mixed)
                else:           # reverse direction: h_prev(t) = h(t+1)
                    ops.gemm_tn(a_sl[:(T - 1) * Bp], y_sl[Bp:], dwhh, mixed=mixed)
            g[base + 4 * d + 0] = dwih[d * 4 * H:(d + 1) * 4 * H]
            g[base + 4 * d + 1] = dwhh
            # b_ih and b_hh have the same gradient, but they must not receive the same tensor OBJECT: autograd may
            # then install one tensor as the .grad of both parameters, and an in-place clip_grad_norm_ (04:501) would
            # scale it twice
            g[base + 4 * d + 2] = dbias[d * 4 * H:(d + 1) * 4 * H]
            g[base + 4 * d + 3] = dbias[d * 4 * H:(d + 1) * 4 * H].clone()
        # dX of this layer = dY of the layer below; when that layer's output dropout was fused into its
        # producer, its backward (the same mask) is fused into this GEMM's epilogue
        below_fused = layer > 0 and sv["layers"][layer - 1]["fused_drop"]
        wt = _t(wih)
        if dP.dtype == torch.bfloat16 and ops.dma_ok(dP.shape[1], wt.shape[0], dP.shape[0]):
            wt = wt.to(torch.bfloat16)                # bf16 x bf16 -> LDS-DMA kernel
        dY = ops.gemm_nt(dP, wt, mixed=mixed, drop_p=p_lstm if below_fused else 0.0,
                         seed=_seed(seed, 10 + layer - 1))
        del dP

    # ---- input projection: Linear -> LayerNorm -> GELU -> Dropout (04:173-178)
    # input_proj.0.bias gradient = column sums of dpre: emitted by the LayerNorm backward itself where it can
    db_fused = need_w and ops.can_fuse_colsum(sv["pre"].shape[1])
    db0 = torch.zeros((sv["pre"].shape[1],), device=dY.device, dtype=torch.float32) if db_fused else None
    dpre, g[2], g[3] = ops.layernorm_act_bwd(sv["pre"], ps[2], ps[3], dY, act=ACT_GELU, remap=(T, B, Bp),
                                             drop_p=p_in, seed=_seed(seed, 0), dx_colsum=db0)
    if sv.get("xb") is not None and need_w:      # mixed: dW through the bf16 TN kernel on the padded bf16 windows
        xb = sv["xb"]
        dwp = torch.zeros((ps[0].shape[0], xb.shape[1]), device=dY.device, dtype=torch.float32)
        ops.gemm_tn(dpre, xb, dwp, mixed=True)
        g[0] = dwp[:, :C].contiguous()
        g[1] = db0 if db0 is not None else ops.colsum(dpre)
        gx2d = ops.gemm_nt(dpre, _t(ps[0])) if needs_input_grad[0] else None
    else:
        gx2d, g[0], g[1] = _linear_bwd(dpre, sv["x2d"], ps[0], need_dx=bool(needs_input_grad[0]), need_w=need_w,
                                       db=db0)
    gx = gx2d.reshape(B, T, C) if gx2d is not None else None
    return gx, g
