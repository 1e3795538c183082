"""Forward / backward pipelines of the drop-in model as ONE autograd node.

The whole of ``EnhancedLSTMModel.forward`` (04_lstm_model.py:206-222) is a single
``torch.autograd.Function``: forward launches the HIP kernels in order on the current
stream and keeps what BPTT needs; backward launches the mirrored kernels and hands torch
one gradient per parameter plus the input gradient (07_explainability.py:242-257 needs
d logits / d x).  Saved activations are never modified by backward, so
``backward(retain_graph=True)`` can be called repeatedly (07:254).
"""
from __future__ import annotations

import torch

from . import images, ops
from .ops import ACT_GELU, ACT_NONE, ACT_TANH, ceil32


COLLECT_HEAD = 4         # input_proj.0.{weight,bias}, input_proj.1.{weight,bias}
COLLECT_TAIL = 12        # layer_norm (2) + attention (4) + classifier (6) positions


def _collect(model):
    """Parameters in pipeline order.  The ablation variants (09_sensitivity_analysis.py:176-242) keep the
    positions and put None where a sub-module is nn.Identity / absent."""
    def wb(mod):
        return [getattr(mod, "weight", None), getattr(mod, "bias", None)]
    ps = [model.input_proj[0].weight, model.input_proj[0].bias] + wb(model.input_proj[1])
    for layer in range(model.num_layers):
        for tup in model.lstm.layer_params(layer):
            ps.extend(tup)
    att = model.attention.attention if model.attention is not None else None
    ps += wb(model.layer_norm)
    ps += [att[0].weight, att[0].bias, att[2].weight, att[2].bias] if att is not None else [None] * 4
    ps += [model.classifier[0].weight, model.classifier[0].bias,
           model.classifier[3].weight, model.classifier[3].bias,
           model.classifier[6].weight, model.classifier[6].bias]
    return ps


def _f32c(t):
    if t is None:
        return None
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _seed(seed, k):
    return (seed * 1000003 + k * 7919 + 12345) & ((1 << 63) - 1)


def _forward_impl(x, ps, cfg, save):
    """Returns (logits, attn, saved-dict or None)."""
    L, D, H, (p_in, p_lstm, p_cls), seed, mixed = cfg
    B, T, C = x.shape
    Bp = ceil32(B)
    W = H * D
    frag = ops.uses_frag(H)
    x2d = x.reshape(B * T, C)
    it = iter(ps)
    proj_w, proj_b, ln0_g, ln0_b = next(it), next(it), next(it), next(it)
    sv = {} if save else None
    # every derived weight image (concatenations, bf16 copies, transposes, b_ih + b_hh) from one launch
    img = images.build(ps, cfg, (B, T, C), save)
    bf16 = torch.bfloat16

    # mixed mode: the windows as a padded bf16 operand (61 -> 64 columns: 16-B aligned rows) for the projection GEMM
    # and, in the backward, its weight gradient -- nn.Linear under autocast (04:174, 04:487); the fp32 kernels walk the
    # unaligned 244-B rows at a third of the rate
    xb = None
    if mixed and frag and C % 8 != 0 and ops.input_proj_ok(x2d, H, C) and proj_w.stride(1) == 1:
        # one launch from the fp32 windows to the first layer's bf16 activations (same bits as the sequence below)
        a, pre, xb = ops.input_proj_ln(x2d, proj_w, proj_b, ln0_g, ln0_b, B, T, Bp, H, act=ACT_GELU, drop_p=p_in,
                                       seed=_seed(seed, 0), save=save)
    elif not mixed and frag and ops.input_proj_f32_ok(x2d, H, C, proj_w):
        # fp32 path: the same fusion with exact fp32 products (the pre-activations go to HBM only for a backward)
        a, pre = ops.input_proj_ln_f32(x2d, proj_w, proj_b, ln0_g, ln0_b, B, T, Bp, H, act=ACT_GELU, drop_p=p_in,
                                       seed=_seed(seed, 0), save=save)
    else:
        if mixed and C % 8 != 0 and H % 8 == 0:
            Cp = (C + 7) // 8 * 8
            xb = ops.pad_cast_bf16(x2d, Cp)
            wpad = img.get("wpad")
            if wpad is None:
                wpad = torch.zeros((proj_w.shape[0], Cp), device=x.device, dtype=torch.float32)
                wpad[:, :C] = proj_w
            pre = ops.gemm_nt(xb, wpad, proj_b, mixed=True)                      # (B*T, H), rows (b,t)
        else:
            pre = ops.gemm_nt(x2d, proj_w, proj_b)                               # (B*T, H), rows (b,t)
        a = ops.layernorm_act(pre, ln0_g, ln0_b, act=ACT_GELU, remap=(T, B, Bp), drop_p=p_in, seed=_seed(seed, 0),
                              out_bf16=mixed and frag)   # (T*Bp, H) time-major; bf16 when only bf16 GEMMs read it
    if save:
        sv["x2d"], sv["pre"], sv["a"], sv["xb"] = x2d, pre, a, xb
        sv["layers"] = []
    inp = a
    # mixed path: the last layer hands its output to the LayerNorm as bf16 only (ops.LN_X_BF16) -- no fp32 Y is written
    # or read.  When saving for a backward, only if that backward will carry bf16 gradients through this LayerNorm
    # (bf16 dV from the attention backward, bf16 dx: the conditions of backward_impl, restated on the forward's facts)
    a0w_p = ps[len(ps) - 10]
    ln_x16 = bool(mixed and ops.LN_X_BF16 and ops.can_fuse_dropout(H, mixed) and ops.ln_x_bf16_ok(W))
    if ln_x16 and save:
        ln_x16 = bool(a0w_p is not None and ops.dy_bf16_ok(H, mixed) and
                      ops.dma_ok(a0w_p.shape[0], W, T * Bp) and ps[len(ps) - 12] is not None)
    for layer in range(L):
        for _ in range(4 * D):
            next(it)
        wih, whh, bias = img[("wih", layer)], img[("whh", layer)], img[("bias", layer)]
        # bf16 x bf16 operands (bf16 activations from the layer below + a bf16 copy of the weights) take the
        # weight-stationary / LDS-DMA GEMMs; everything else the register-staged kernels
        w_in = wih
        if inp.dtype == bf16 and frag and (ops.gate_ws_ok(inp.shape[1], H) or
                                           ops.dma_ok(inp.shape[1], wih.shape[0], inp.shape[0])):
            w_in = img.get(("wih16", layer))
            if w_in is None:
                w_in = wih.to(bf16)
        # fp32 path, H = 128: operand ranges for the fp16-split kernels (images.build); without a bound on the layer's
        # activations (nn.Identity in place of the projection LayerNorm) the gate GEMM stays on the exact-fp32 kernel
        g_rng, r_rng = img.get(("gate_range", layer)), img.get(("rec_range", layer))
        P = ops.gate_gemm_x(inp, w_in, bias, T, Bp, H, D, frag, mixed=mixed, range=g_rng,
                            exact=(not mixed and g_rng is None and r_rng is not None))
        last = layer + 1 == L
        drop_here = not last and p_lstm > 0
        bf16_out = ops.can_fuse_dropout(H, mixed)        # the bf16-MFMA recurrent kernel emits bf16 copies itself
        # fp32 path on the fp16-split kernels: the saving forward writes dropout(Y) next to Y (fp32), its backward is the
        # mask epilogue of the dX GEMM of the layer above
        fuse32 = (drop_here and not mixed and not bf16_out and frag and r_rng is not None and P.dtype == torch.float32
                  and ops.can_fuse_dropout_f32(H, Bp, save))
        fuse = drop_here and (bf16_out or fuse32)
        # mixed mode: layers below the last never materialise fp32 Y (only bf16 consumers remain: the next
        # layer's GEMMs read Yd / Y16, dW_hh reads Y16); the last layer keeps fp32 Y for the LayerNorm
        Y, Cs, Y16, Yd = ops.lstm_rec_fwd(P, whh, T, Bp, H, D, save, mixed=mixed,
                                          drop_p=p_lstm if fuse else 0.0, seed=_seed(seed, 10 + layer),
                                          want_f32=(last and not ln_x16) or not bf16_out,
                                          want_bf16=bf16_out and (save or not last or ln_x16), nvalid=B, range=r_rng)
        if fuse:
            nxt = Yd
        elif bf16_out and not last:
            nxt = Y16
        elif drop_here:
            nxt = ops.dropout(Y, p_lstm, _seed(seed, 10 + layer))
        elif last and ln_x16:
            nxt = Y16
        else:
            nxt = Y
        if save:
            sv["layers"].append({"inp": inp, "G": P, "C": Cs, "Y": Y16 if Y16 is not None else Y, "wih": wih,
                                 "whh": whh, "fused_drop": fuse, "wihT": img.get(("wihT", layer)),
                                 "rec_range": r_rng, "bwd_range": img.get(("bwd_range", layer))})
        inp = nxt
    ln_g, ln_b = next(it), next(it)
    a0w, a0b, a2w, a2b = next(it), next(it), next(it), next(it)
    c0w, c0b, c3w, c3b, c6w, c6b = (next(it) for _ in range(6))
    # mixed mode: the normalised sequence v only feeds bf16 MFMA GEMMs and the pooling sums -> bf16
    w1 = None
    if a0w is not None and mixed and ln_g is not None and ops.attn_scores_ok(inp, H, D, Bp, a0w):
        # one launch: LayerNorm + score layer of the attention (same bits as the three kernels of the branch below)
        w1 = img.get("a0w16")
        if w1 is None:
            w1 = a0w.to(bf16)
        v, u, S = ops.attn_scores(inp, ln_g, ln_b, w1, a0b, a2w.reshape(-1), a2b, T, B, Bp, H, D, save=save)
        ctx, attn = ops.attn_pool_fwd_scores(v, S, T, B, Bp)
    elif (a0w is not None and not mixed and ln_g is not None and a0b is not None
          and ops.attn_scores_f32_ok(inp, H, D, Bp, a0w)):
        # fp32 path: the same fusion on the fp16-split arithmetic (v bit-identical to the LayerNorm kernel's)
        w1 = a0w
        v, u, S = ops.attn_scores_f32(inp, ln_g, ln_b, a0w, a0b, a2w.reshape(-1), a2b, T, B, Bp, H, D, save=save)
        ctx, attn = ops.attn_pool_fwd_scores(v, S, T, B, Bp)
    else:
        v = ops.layernorm_act(inp, ln_g, ln_b, out_bf16=mixed)                   # (T*Bp, W)
        if a0w is None:            # no-attention ablation: mean pooling over time (09:236)
            u = None
            ctx, attn = ops.attn_pool_fwd(v, None, None, None, T, B, Bp)
        else:
            w1 = a0w
            if v.dtype == bf16 and ops.dma_ok(v.shape[1], a0w.shape[0], v.shape[0]):
                w1 = img.get("a0w16")
                if w1 is None:
                    w1 = a0w.to(bf16)
            u = ops.gemm_nt(v, w1, a0b, act=ACT_TANH, mixed=mixed)               # (T*Bp, W/2)
            ctx, attn = ops.attn_pool_fwd(v, u, a2w.reshape(-1), a2b, T, B, Bp)
    if save:       # keep the pre-activations of the two classifier GELUs for their backward
        z1p = ops.gemm_nt(ctx, c0w, c0b)
        z1 = ops.act(z1p, ACT_GELU)
    else:
        z1p, z1 = None, ops.gemm_nt(ctx, c0w, c0b, act=ACT_GELU)
    z1d = ops.dropout(z1, p_cls, _seed(seed, 20)) if p_cls > 0 else z1
    if save:
        z2p = ops.gemm_nt(z1d, c3w, c3b)
        z2 = ops.act(z2p, ACT_GELU)
    else:
        z2p, z2 = None, ops.gemm_nt(z1d, c3w, c3b, act=ACT_GELU)
    z2d = ops.dropout(z2, p_cls, _seed(seed, 21)) if p_cls > 0 else z2
    logits = ops.gemm_nt(z2d, c6w, c6b)
    if save:
        sv.update(ylast=inp, v=v, u=u, ctx=ctx, attn=attn, z1p=z1p, z1d=z1d, z2p=z2p, z2d=z2d,
                  wT={k: img.get(k) for k in ("a0wT", "c0wT", "c3wT", "c6wT")})
    return logits, attn, sv


class _LobModelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cfg, sink, *params):
        ps = [_f32c(p) for p in params]
        xf = _f32c(x)
        need_grad = any(ctx.needs_input_grad)      # grad mode is off inside Function.forward
        logits, attn, sv = _forward_impl(xf, ps, cfg, save=need_grad)
        ctx.cfg = cfg
        ctx.sv = sv
        ctx.sink = sink
        ctx.ps = ps if need_grad else None
        ctx.x_shape = tuple(x.shape)
        ctx.mark_non_differentiable(attn)
        return logits, attn

    @staticmethod
    def backward(ctx, dlogits, _dattn):
        from .backward import backward_impl
        need = ctx.needs_input_grad
        # need[] is (x, cfg, sink, *params): backward_impl indexes the parameters from position 2
        nig = (need[0], False) + tuple(need[3:])
        with ops.on_device(dlogits.device):
            gx, gps = backward_impl(ctx.sv, ctx.ps, ctx.cfg, ctx.x_shape, dlogits.contiguous().float(), nig,
                                    sink=ctx.sink)
        return (gx, None, None) + tuple(g if need[3 + i] else None for i, g in enumerate(gps))


_PARAM_GRADS = True


class input_grad_only:
    """Context manager: forwards run inside it track the gradient w.r.t. the input windows only -- the parameters
    enter the autograd node detached, so its backward skips every weight-gradient GEMM (a third of the backward).
    Used by the attribution path (07_explainability.py:239-263 reads X.grad, never a parameter gradient)."""

    def __enter__(self):
        global _PARAM_GRADS
        self._old, _PARAM_GRADS = _PARAM_GRADS, False

    def __exit__(self, *exc):
        global _PARAM_GRADS
        _PARAM_GRADS = self._old


def lob_forward(model, x, drops, seed):
    if not x.is_cuda:
        raise ops._lib.LobError("EnhancedLSTMModel.forward: input must be on the GPU "
                                "(the MI355X path has no CPU fallback)")
    # mixed precision (bf16 MFMA inputs for the gate / attention GEMMs, fp32 everything else) when the
    # caller runs the model under autocast, as the reference's training and inference loops do on a GPU
    # (04_lstm_model.py:487, 06_lstm_ode_integration.py:349), or when model.gate_gemm_dtype == "bf16".
    mixed = bool(torch.is_autocast_enabled("cuda")) or getattr(model, "gate_gemm_dtype", "f32") == "bf16"
    cfg = (model.num_layers, model.num_directions, model.hidden_size, tuple(float(d) for d in drops), int(seed),
           mixed)
    params = _collect(model)
    ops.same_device([x] + params, "EnhancedLSTMModel.forward (input and parameters)")
    sink = None
    if not _PARAM_GRADS:
        params = [None if p is None else p.detach() for p in params]
    elif torch.is_grad_enabled():
        # an attached FusedAdamW (training.FusedAdamW(..., model=model)): the backward accumulates the parameter
        # gradients straight into its flat gradient buffer and hands autograd None for them
        from .training import grad_sink_of
        opt = grad_sink_of(model)
        if opt is not None:
            sink = opt.sink_for(params, model.num_directions)
    with ops.on_device(x.device), torch.autocast(device_type="cuda", enabled=False):
        if not torch.is_grad_enabled():
            # torch.no_grad() (every inference loop of the reference: 04_lstm_model.py:557, 06_lstm_ode_integration.py:347):
            # ctx.needs_input_grad still reports the parameters' requires_grad there, so going through the autograd node
            # would run the SAVING forward kernels (gates and cell states written for a backward that cannot happen)
            logits, attn, _ = _forward_impl(_f32c(x), [_f32c(p) for p in params], cfg, save=False)
            return logits, attn
        return _LobModelFn.apply(x, cfg, sink, *params)


class _AttentionFn(torch.autograd.Function):
    """Stand-alone ``Attention.forward`` (04_lstm_model.py:123-128) as one autograd node: both outputs (context and
    weights) are differentiable, gradients flow to the input sequence and to the four parameters."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        B, T, W = x.shape
        v = _f32c(x).transpose(0, 1).contiguous().reshape(T * B, W)          # time-major rows t*B + b (no padding)
        w1f, b1f, w2f, b2f = _f32c(w1), _f32c(b1), _f32c(w2), _f32c(b2)
        u = ops.gemm_nt(v, w1f, b1f, act=ACT_TANH)
        cx, attn = ops.attn_pool_fwd(v, u, w2f.reshape(-1), b2f, T, B, B)
        if any(ctx.needs_input_grad):
            ctx.sv = (v, u, attn, w1f, w2f, (B, T, W))
        return cx, attn

    @staticmethod
    def backward(ctx, dcx, dattn):
        v, u, attn, w1f, w2f, (B, T, W) = ctx.sv
        with ops.on_device(v.device):
            dcx = torch.zeros((B, W), device=v.device) if dcx is None else dcx.contiguous().float()
            dattn = None if dattn is None else dattn.contiguous().float()
            dV, dU, dw2 = ops.attn_pool_bwd(v, u, attn, dcx, w2f.reshape(-1), T, B, B, want_dv=True, dattn=dattn)
            ops.gemm_nt(dU, w1f.t().contiguous(), out=dV, accumulate=True)  # + dPreU W1
            dw1 = ops.gemm_tn(dU, v, torch.zeros_like(w1f))
            db1 = ops.colsum(dU)
            dx = dV.reshape(T, B, W).transpose(0, 1).contiguous()
        # b2 cancels in the softmax over time: its gradient is exactly zero
        return dx, dw1, db1, dw2.reshape(w2f.shape), torch.zeros((1,), device=v.device)


def attention_forward(lstm_output, w1, b1, w2, b2):
    """Stand-alone Attention.forward on a batch-first (B,T,W) tensor: (context (B,W), weights (B,T)), trainable like
    the reference's nn.Module (04_lstm_model.py:112-128)."""
    if not lstm_output.is_cuda:
        raise ops._lib.LobError("Attention.forward: input must be on the GPU")
    ops.same_device([lstm_output, w1, b1, w2, b2], "Attention.forward")
    with ops.on_device(lstm_output.device), torch.autocast(device_type="cuda", enabled=False):
        if not torch.is_grad_enabled():
            B, T, W = lstm_output.shape
            v = _f32c(lstm_output).transpose(0, 1).contiguous().reshape(T * B, W)
            u = ops.gemm_nt(v, _f32c(w1), _f32c(b1), act=ACT_TANH)
            return ops.attn_pool_fwd(v, u, _f32c(w2).reshape(-1), _f32c(b2), T, B, B)
        return _AttentionFn.apply(lstm_output, w1, b1, w2, b2)
