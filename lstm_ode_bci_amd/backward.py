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


_SIDE = {}


def _side_stream(dev):
    """One second stream per device for work nobody waits for inside the backward (ops.OVERLAP_DW)."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=dev)
    return st


def _t(w):
    return w.t().contiguous()


class _ZeroPool:
    """Accumulation targets of the backward (weight-gradient GEMMs, bias / LayerNorm gradients: fp32 atomics into a
    zeroed buffer) carved out of ONE zero-filled allocation: one fill launch instead of ~25."""
    ALIGN = 64          # floats

    def __init__(self, dev, capacity):
        self.buf = torch.zeros(int(capacity), device=dev, dtype=torch.float32)
        self.off = 0

    def take(self, shape):
        n = 1
        for s in shape:
            n *= int(s)
        if self.off + n > self.buf.numel():
            return torch.zeros(tuple(shape), device=self.buf.device, dtype=torch.float32)
        out = self.buf[self.off:self.off + n].view(tuple(shape))
        self.off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        return out


def _linear_bwd(dy, x, w, need_dx=True, need_w=True, db=None, wt=None, tgt=None, iw=None):
    """y = x w^T + b  ->  (dx, dw, db); db may be handed in when a producer of dy already summed its columns; wt is
    the pre-built transpose of w (images.build); tgt / iw: accumulation targets of parameters iw (weight), iw + 1 (bias)."""
    dw = None
    if need_w:
        dw = tgt.param(iw, w.shape) if tgt is not None else torch.zeros_like(w)
        ops.gemm_tn(dy, x, dw)
        if db is None:      # (a db that is handed in has already been accumulated into its target by the producer of dy)
            db = ops.colsum(dy, tgt.param(iw + 1, (dy.shape[1],)) if tgt is not None else None)
    dx = ops.gemm_nt(dy, wt if wt is not None else _t(w)) if need_dx else None
    return dx, dw, db


class _Targets:
    """Where the backward accumulates parameter gradients.  Every weight-gradient kernel ADDS into its destination (fp32
    atomics), so the destination can be either a zeroed piece of one pool (the gradients are then returned to autograd,
    which adds them to p.grad with one launch per parameter) or -- with a gradient sink, i.e. an attached FusedAdamW --
    the parameter's own slice of the optimizer's flat gradient buffer: no zero fill, no per-parameter accumulation
    launches, and autograd receives None for the parameters."""

    def __init__(self, ps, dev, sink):
        self.sink = sink
        self.pool = _ZeroPool(dev, sum(p.numel() + 64 for p in ps if p is not None) + 8192 if sink is None else 16384)

    def zeros(self, shape):
        return self.pool.take(shape)

    def param(self, i, shape):
        """Accumulation target of parameter i."""
        return self.sink.view(i, shape) if self.sink is not None else self.pool.take(shape)

    def span(self, i, step, count, shape):
        """One contiguous target covering parameters i, i + step, ... (the two directions of an LSTM layer)."""
        if self.sink is not None:
            return self.sink.span(i, step, count, shape)
        return self.pool.take(shape)

    def deliver(self, i, value):
        """A gradient that was computed into a temporary: with a sink it is added to the flat buffer here."""
        if self.sink is None:
            return value
        self.sink.view(i, value.shape).add_(value)
        return None

    def result(self, g):
        """What backward hands to autograd for parameter gradients."""
        return [None] * len(g) if self.sink is not None else g


def width_ok(w):
    """Widths at which the LayerNorm backward reads / writes bf16 gradient streams."""
    return w in (128, 256, 512)


def backward_impl(sv, ps, cfg, x_shape, dlogits, needs_input_grad, sink=None):
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
    wT = sv.get("wT") or {}
    # mixed path, H = 128: the gradient between the LSTM layers / LayerNorms travels as bf16 (ops.DY_BF16_CARRY); it
    # needs the fused dropout-backward epilogue of the dX GEMM (the stand-alone dropout kernel is fp32)
    carry16 = ops.dy_bf16_ok(H, mixed) and all(l["fused_drop"] or p_lstm == 0 or i + 1 == L
                                               for i, l in enumerate(sv["layers"]))
    # accumulation targets: one zero-filled pool per backward call (retained-graph passes never share gradients), or the
    # attached optimizer's flat gradient buffer (sink)
    if not need_w:
        sink = None
    tgt = _Targets(ps, dlogits.device, sink) if need_w else None

    def zeros(shape):
        return tgt.zeros(shape) if tgt is not None else torch.zeros(tuple(shape), device=dlogits.device)

    # ---- classifier (04:196-204)
    dz2d, g[i_c6w], g[i_c6w + 1] = _linear_bwd(dlogits, sv["z2d"], ps[i_c6w], need_w=need_w, wt=wT.get("c6wT"), tgt=tgt, iw=i_c6w)
    dz2 = ops.dropout(dz2d, p_cls, _seed(seed, 21)) if p_cls > 0 else dz2d
    dz2p = ops.act_bwd(dz2, sv["z2p"], ACT_GELU)
    dz1d, g[i_c3w], g[i_c3w + 1] = _linear_bwd(dz2p, sv["z1d"], ps[i_c3w], need_w=need_w, wt=wT.get("c3wT"), tgt=tgt, iw=i_c3w)
    dz1 = ops.dropout(dz1d, p_cls, _seed(seed, 20)) if p_cls > 0 else dz1d
    dz1p = ops.act_bwd(dz1, sv["z1p"], ACT_GELU)
    dctx, g[i_c0w], g[i_c0w + 1] = _linear_bwd(dz1p, sv["ctx"], ps[i_c0w], need_w=need_w, wt=wT.get("c0wT"), tgt=tgt, iw=i_c0w)

    # ---- attention pooling (04:123-128)
    v, u = sv["v"], sv["u"]
    a0w = ps[i_a0w]
    if a0w is None:          # mean pooling (09:236): dV = dctx / T, no parameters
        dV, _, _ = ops.attn_pool_bwd(v, None, sv["attn"], dctx, None, T, B, Bp)
        pool_ctx = None
    else:
        fused = v.dtype == torch.bfloat16        # mixed mode: bf16 v / dU, context path folded into the LN backward
        # the score MLP's first-bias gradient = column sums of dU: emitted by the pooling backward itself where it can
        cs_fused = need_w and ops.attn_bwd_fuses_colsum(v, u, not fused, fused)
        du_cs = tgt.param(i_a0b, (u.shape[1],)) if cs_fused else None
        dw2_t = tgt.param(i_a2w, (u.shape[1],)) if need_w else zeros((u.shape[1],))
        dV, dU, dw2 = ops.attn_pool_bwd(v, u, sv["attn"], dctx, ps[i_a2w].reshape(-1), T, B, Bp,
                                        want_dv=not fused, du_bf16=fused, du_colsum=du_cs, dw2=dw2_t)
        g[i_a2w] = dw2.reshape(1, -1)
        # b2 cancels in the softmax: its gradient is exactly 0 (a zero tensor for autograd; nothing to add to a sink)
        g[i_a2b] = zeros(ps[i_a2b].shape) if (sink is None or not need_w) else None
        if need_w:
            g[i_a0w] = tgt.param(i_a0w, a0w.shape)
            ops.gemm_tn(dU, v, g[i_a0w], mixed=mixed)
            g[i_a0b] = du_cs if cs_fused else ops.colsum(dU, tgt.param(i_a0b, (dU.shape[1],)))
        w1t = wT.get("a0wT")
        fused_tail = None
        if fused:
            want16 = ops.dma_ok(dU.shape[1], a0w.shape[1], dU.shape[0])
            if w1t is None or (w1t.dtype == torch.bfloat16) != want16:
                w1t = _t(a0w).to(torch.bfloat16) if want16 else _t(a0w)
            if (carry16 and want16 and width_ok(sv["ylast"].shape[1]) and ps[i_ln] is not None
                    and ops.attn_ln_bwd_ok(sv["ylast"], dU, w1t, H, D, Bp, ps[i_ln])):
                fused_tail = (dU, w1t)        # dV = dU W1 is formed inside the LayerNorm backward below
                dV = None
            else:
                dV = ops.gemm_nt(dU, w1t, mixed=mixed,                   # dU W1; + a[t] dctx is added below
                                 out_bf16=carry16 and want16 and width_ok(sv["ylast"].shape[1]))
            pool_ctx = (sv["attn"], dctx, T, B, Bp)
        else:
            if w1t is None or w1t.dtype != torch.float32:
                w1t = _t(a0w)
            ops.gemm_nt(dU, w1t, out=dV, accumulate=True, mixed=mixed)
            pool_ctx = None

    # ---- post-LSTM LayerNorm (04:212)
    def affine(i):          # accumulation targets of a LayerNorm's weight / bias (None: nn.Identity in the ablation variants)
        if tgt is None or ps[i] is None:
            return None, None
        return tgt.param(i, ps[i].shape), tgt.param(i + 1, ps[i + 1].shape)
    dg_t, db_t = affine(i_ln)
    ylast = sv["ylast"]
    dx16 = carry16 and width_ok(ylast.shape[1])
    if a0w is not None and fused_tail is not None:
        dY, g[i_ln], g[i_ln + 1] = ops.attn_ln_bwd(ylast, ps[i_ln], ps[i_ln + 1], fused_tail[0], fused_tail[1], sv["attn"], dctx,
                                                   T, B, Bp, H, D, dg=dg_t, db=db_t)
    else:
        if ylast.dtype == torch.bfloat16 and not (dx16 and dV.dtype == torch.bfloat16):
            ylast = ylast.float()        # a storage switch was flipped between forward and backward: widen, stay correct
        dY, g[i_ln], g[i_ln + 1] = ops.layernorm_act_bwd(ylast, ps[i_ln], ps[i_ln + 1], dV, pool=pool_ctx, dg=dg_t, db=db_t,
                                                         dx_bf16=dx16)

    side_used = False
    pending_dw = None

    def _launch_pending():
        ev, dP_, inp_, Y_, out_ = pending_dw
        side = _side_stream(dP_.device)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            ops.lstm_dw(dP_, inp_, Y_, T, Bp, H, D, out=out_)
        for t_ in (dP_, inp_, Y_):
            t_.record_stream(side)         # the allocator must not hand these out again before the side stream is done

    def _join():          # the optimizer (and anything else on this stream) must see the side stream's weight gradients
        if side_used:
            torch.cuda.current_stream().wait_stream(_side_stream(dlogits.device))

    # ---- LSTM stack, last layer first (04:211)
    for layer in reversed(range(L)):
        lay = sv["layers"][layer]
        if layer + 1 < L and p_lstm > 0 and not lay["fused_drop"]:
            dY = ops.dropout(dY, p_lstm, _seed(seed, 10 + layer))
        base = 4 + layer * 4 * D
        # b_ih and b_hh have the same gradient.  With a sink the bf16 BPTT kernels add it into BOTH parameters' slices
        # themselves (two destinations); the fp32 kernels add into a temporary that is then added to both slices (the
        # slices hold earlier micro-batches: BPTT cannot add into one and copy to the other)
        two_dst = need_w and sink is not None and mixed and ops.bf16_rec(H, lay["G"].dtype == torch.bfloat16)
        ax_in = ax_h = ax_w = None       # operand ranges of the fp16-split fp32 GEMMs (None: the exact-fp32 / bf16 kernels)
        if two_dst:
            dP, dbias = ops.lstm_rec_bwd(lay["G"], lay["C"], lay["whh"], dY, T, Bp, H, D, dp_bf16=True,
                                         dbias=tgt.span(base + 2, 4, D, (D * 4 * H,)),
                                         dbias2=tgt.span(base + 3, 4, D, (D * 4 * H,)))
        else:
            # fp32 path, H = 128: BPTT and the three GEMMs on dP carry their fp32 products as two-way fp16 splits (as the
            # forward's gate GEMMs do); the BPTT kernel reports max|dP|, the other operands' ranges come from the forward
            br = lay.get("bwd_range")
            split = bool(not mixed and br is not None and lay["G"].dtype == torch.float32 and dY.dtype == torch.float32
                         and ops.f32_split_bwd_ok(H, Bp))
            amax_dp = zeros((1,)) if split else None
            dP, dbias = ops.lstm_rec_bwd(lay["G"], lay["C"], lay["whh"], dY, T, Bp, H, D, dp_bf16=mixed,
                                         dbias=zeros((D * 4 * H,)), amax_out=amax_dp,
                                         range=lay.get("rec_range") if split else None)
            if split:
                ax_in = (amax_dp, br[0]) if br[0] is not None else None       # dW_ih = dP^T x: x bounded by the LayerNorm / |h|
                ax_h, ax_w = (amax_dp, br[2]), (amax_dp, br[1])
        if pending_dw is not None:        # the layer above's weight gradients, next to the BPTT just launched
            _launch_pending()
            pending_dw = None
            side_used = True
        inp, Y, wih = lay["inp"], lay["Y"], lay["wih"]
        fused_dw = need_w and ops.can_fuse_dw(dP, inp, Y, T, Bp, H, D)
        deferred_dw = None
        if fused_dw:     # one contiguous target over both directions (the sink lays the two directions out side by side)
            dw_out = (tgt.span(base, 4, D, (D * 4 * H, inp.shape[1])), tgt.span(base + 1, 4, D, (D, 4 * H, H)))
            if sink is not None and layer > 0 and ops.OVERLAP_DW and ops.rec_underfilled(H, Bp, D):
                # small batches: the BPTT of the layer below occupies at most half of the CUs and is one long serial chain;
                # this layer's weight gradients (nobody waits for them before the optimizer) run next to it on a second
                # stream, launched AFTER dX -- the chain's only input -- has been queued
                deferred_dw = (dP, inp, Y, dw_out)
            else:
                dwih, dwhh_all = ops.lstm_dw(dP, inp, Y, T, Bp, H, D, out=dw_out)
        elif need_w:
            dwih = tgt.span(base, 4, D, wih.shape)
            ops.gemm_tn(dP, inp, dwih, mixed=mixed, amax=ax_in)
        if need_w and sink is not None and not two_dst:
            tgt.span(base + 2, 4, D, dbias.shape).add_(dbias)
            tgt.span(base + 3, 4, D, dbias.shape).add_(dbias)
        dbias2 = dbias.clone() if (need_w and sink is None) else None   # b_ih and b_hh: equal gradients, distinct tensors
        for d in range(D if need_w else 0):
            if sink is not None and fused_dw:
                continue
            dwhh = dwhh_all[d] if fused_dw else tgt.param(base + 4 * d + 1, ps[base + 4 * d + 1].shape)
            if T > 1 and not fused_dw:
                a_sl = dP[:, d * 4 * H:(d + 1) * 4 * H]
                y_sl = Y[:, d * H:(d + 1) * H]
                if d == 0:      # h_prev(t) = h(t-1)
                    ops.gemm_tn(a_sl[Bp:], y_sl[:(T - 1) * Bp], dwhh, mixed=mixed, amax=ax_h)
                else:           # reverse direction: h_prev(t) = h(t+1)
                    ops.gemm_tn(a_sl[:(T - 1) * Bp], y_sl[Bp:], dwhh, mixed=mixed, amax=ax_h)
            if sink is not None:
                continue
            g[base + 4 * d + 0] = dwih[d * 4 * H:(d + 1) * 4 * H]
            g[base + 4 * d + 1] = dwhh
            # b_ih and b_hh have the same gradient, but they must not receive the same tensor OBJECT: autograd may
            # then install one tensor as the .grad of both parameters, and an in-place clip_grad_norm_ (04:501) would
            # scale it twice
            g[base + 4 * d + 2] = dbias[d * 4 * H:(d + 1) * 4 * H]
            g[base + 4 * d + 3] = dbias2[d * 4 * H:(d + 1) * 4 * H]
        # dX of this layer = dY of the layer below; when that layer's output dropout was fused into its
        # producer, its backward (the same mask) is fused into this GEMM's epilogue
        below_fused = layer > 0 and sv["layers"][layer - 1]["fused_drop"]
        want16 = dP.dtype == torch.bfloat16 and ops.dma_ok(dP.shape[1], wih.shape[1], dP.shape[0])
        wt = lay.get("wihT")
        if wt is None or (wt.dtype == torch.bfloat16) != want16:
            wt = _t(wih).to(torch.bfloat16) if want16 else _t(wih)      # bf16 x bf16 -> LDS-DMA kernel
        # the consumer of a bf16 dX: the BPTT kernel of the layer below, or (layer 0) the projection LayerNorm backward
        dx16 = carry16 and want16 and (layer > 0 or width_ok(sv["pre"].shape[1]))
        dY = ops.gemm_nt(dP, wt, mixed=mixed, drop_p=p_lstm if below_fused else 0.0,
                         seed=_seed(seed, 10 + layer - 1), out_bf16=dx16, amax=ax_w)
        if deferred_dw is not None:
            # queued on the side stream only AFTER the BPTT of the layer below has been launched (top of the next iteration):
            # launched first, the GEMM's 256 workgroups take every CU and the chain starts a GEMM late (measured: no gain)
            ev = torch.cuda.Event()
            ev.record()
            pending_dw = (ev,) + deferred_dw
            deferred_dw = None
        del dP

    if pending_dw is not None:            # cannot happen (only layers > 0 defer); never lose a gradient to a logic slip
        _launch_pending()
        pending_dw = None
        side_used = True
    # ---- input projection: Linear -> LayerNorm -> GELU -> Dropout (04:173-178)
    # input_proj.0.bias gradient = column sums of dpre: emitted by the LayerNorm backward itself where it can
    db_fused = need_w and ops.can_fuse_colsum(sv["pre"].shape[1])
    db0 = tgt.param(1, (sv["pre"].shape[1],)) if db_fused else None
    dg_t, db_t = affine(2)
    # mixed path: dpre feeds only the bf16 TN GEMM below (which rounds it to bf16 on its way into the MFMA) and the column
    # sums taken inside the LayerNorm backward from the fp32 values -> store it as bf16: the same operand bits, half the
    # bytes written and read (an input gradient, if asked for, keeps the fp32 rows)
    dpre16 = bool(ops.DPRE_BF16 and sv.get("xb") is not None and need_w and db_fused and not needs_input_grad[0]
                  and dY.dtype == torch.bfloat16 and sv["pre"].shape[1] in (128, 256))
    if dpre16 and ops.input_proj_bwd_ok(sv["pre"], dY, sv["xb"], H):
        # one launch: LayerNorm / GELU / dropout backward + the Linear's weight gradient; dpre never goes to HBM
        xb = sv["xb"]
        dwp = zeros((ps[0].shape[0], xb.shape[1]))
        _, g[2], g[3] = ops.input_proj_bwd(sv["pre"], ps[2], ps[3], dY, xb, dwp, B, T, Bp, H, act=ACT_GELU, drop_p=p_in,
                                           seed=_seed(seed, 0), dg=dg_t, db=db_t, dbias=db0)
        g[0] = tgt.deliver(0, dwp[:, :C]) if sink is not None else dwp[:, :C].contiguous()
        g[1] = db0
        _join()
        return None, (tgt.result(g) if tgt is not None else g)
    dpre, g[2], g[3] = ops.layernorm_act_bwd(sv["pre"], ps[2], ps[3], dY, act=ACT_GELU, remap=(T, B, Bp),
                                             drop_p=p_in, seed=_seed(seed, 0), dx_colsum=db0, dg=dg_t, db=db_t,
                                             dx_bf16=dpre16)
    if sv.get("xb") is not None and need_w:      # mixed: dW through the bf16 TN kernel on the padded bf16 windows
        xb = sv["xb"]
        dwp = zeros((ps[0].shape[0], xb.shape[1]))
        ops.gemm_tn(dpre, xb, dwp, mixed=True)
        g[0] = tgt.deliver(0, dwp[:, :C]) if sink is not None else dwp[:, :C].contiguous()
        g[1] = db0 if db0 is not None else ops.colsum(dpre, tgt.param(1, (dpre.shape[1],)))
        gx2d = ops.gemm_nt(dpre, _t(ps[0])) if needs_input_grad[0] else None
    else:
        gx2d, g[0], g[1] = _linear_bwd(dpre, sv["x2d"], ps[0], need_dx=bool(needs_input_grad[0]), need_w=need_w,
                                       db=db0, tgt=tgt, iw=0)
    gx = gx2d.reshape(B, T, C) if gx2d is not None else None
    _join()
    return gx, (tgt.result(g) if tgt is not None else g)
