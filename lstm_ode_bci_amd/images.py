"""Derived weight images of one forward (+ backward), built by ONE launch (``lob_prep_weights``, csrc/prep.hip).

What the kernels take as operands is not what ``state_dict`` holds (04_lstm_model.py:181-188 keeps one W_ih / W_hh / b_ih /
b_hh per direction): the gate GEMM wants the two directions' W_ih concatenated (in bf16 on the mixed path), the dX GEMM
its transpose, the recurrent kernels W_hh stacked, everyone ``b_ih + b_hh``.  torch built those with ~50 tiny launches per
training step.  Here the list of images is decided on the host from the configuration alone, every destination is a
plain ``torch.empty`` (caching allocator: no launch), and one kernel fills them all from the LIVE parameters -- there is
no cache, so there is nothing to invalidate when an optimizer rewrites the parameters through raw pointers.

``build()`` returns a dict; consumers look an image up and fall back to the torch expression if it is absent, so the
arithmetic never depends on this module's bookkeeping being complete.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops


class _Plan:
    def __init__(self, dev):
        self.dev, self.items, self.keep = dev, [], []

    def add_range(self, dst, off, src=None, src2=None, lnbound=False, factor=1.0):
        """dst[off] <- max |src| (lnbound=False) or the LayerNorm-derived activation bound (lnbound=True; src = gain,
        src2 = bias, both may be None: bound = factor) -- the operand ranges of the fp16-split kernels (include/lob.h)."""
        rows = cols = 1
        if src is not None:
            src = src.detach()
            assert src.is_contiguous() and src.dtype == torch.float32 and src.dim() <= 2
            rows, cols = (1, src.shape[0]) if src.dim() == 1 else src.shape
        kind = _lib.PREP_LNBOUND if lnbound else _lib.PREP_ABSMAX
        op = _lib.PrepOp(0 if src is None else src.data_ptr(), 0 if src2 is None else src2.detach().data_ptr(),
                         dst.data_ptr() + off * 4, rows, cols, cols, 1, 0, kind, 0, int(round(factor * 1000)))
        self.items.append(op)
        self.keep.append((src, src2, dst))

    def add(self, src, dst, dst_ptr_off=0, ld_dst=None, src2=None, transpose=False, pad_to=0):
        """dst (a fresh tensor; dst_ptr_off in ELEMENTS selects a row / column block of it) <- src [rows, cols]."""
        src = src.detach()
        assert src.is_contiguous() and src.dtype == torch.float32 and src.dim() <= 2
        rows, cols = (1, src.shape[0]) if src.dim() == 1 else src.shape
        bf = dst.dtype == torch.bfloat16
        if ld_dst is None:
            ld_dst = dst.shape[-1]
        kind = (_lib.PREP_TRANSPOSE if transpose else 0) | (_lib.PREP_BF16 if bf else 0)
        op = _lib.PrepOp(src.data_ptr(), 0 if src2 is None else src2.detach().data_ptr(),
                         dst.data_ptr() + dst_ptr_off * dst.element_size(), rows, cols, cols, ld_dst, pad_to, kind, 0, 0)
        self.items.append(op)
        self.keep.append((src, src2, dst))

    def run(self):
        if not self.items:
            return
        for i in range(0, len(self.items), _lib.PREP_MAX):
            chunk = self.items[i:i + _lib.PREP_MAX]
            arr = (_lib.PrepOp * len(chunk))(*chunk)
            rc = _lib.lib().lob_prep_weights(C.cast(arr, C.c_void_p), len(chunk), ops._stream())
            _lib.check(rc, "lob_prep_weights")


def act_is_bf16(layer, cfg, frag):
    """Whether the activations entering LSTM layer `layer` are stored as bf16 (mirrors autograd._forward_impl)."""
    L, D, H, (p_in, p_lstm, p_cls), seed, mixed = cfg
    if layer == 0:
        return bool(mixed and frag and H in (128, 256, 512))
    return ops.can_fuse_dropout(H, mixed)            # the bf16-MFMA recurrent kernels hand bf16 copies to the next layer


def build(ps, cfg, x_shape, need_grad):
    """ps: parameters in ``autograd._collect`` order (fp32, contiguous, detached).  Returns {key: tensor}."""
    L, D, H, (p_in, p_lstm, _p_cls), _, mixed = cfg
    B, T, Cc = x_shape
    Bp = ops.ceil32(B)
    rows = T * Bp
    frag = ops.uses_frag(H)
    dev = ps[0].device
    bf16, f32 = torch.bfloat16, torch.float32
    plan = _Plan(dev)
    img = {}

    def new(shape, dtype=f32):
        return torch.empty(shape, device=dev, dtype=dtype)

    proj_w = ps[0]
    if mixed and Cc % 8 != 0 and H % 8 == 0:
        Cp = (Cc + 7) // 8 * 8
        img["wpad"] = new((proj_w.shape[0], Cp))
        plan.add(proj_w, img["wpad"], pad_to=Cp)
    # fp32 path at H = 128: the fp16-split kernels take the range of each operand tensor (max |W_ih| / |W_hh| per
    # direction, a bound on the activations) and choose their pre-scales from it -- a checkpoint with a large weight or
    # LayerNorm gain stays finite and accurate.  Layer 0's activations are dropout(GELU(LayerNorm(.))): bounded by
    # sqrt(H) max|gain| + max|bias| (no bound with nn.Identity in its place: that layer's GEMM runs exact); the other
    # layers see |h| < 1 (times the dropout scale in train mode).
    want_range = bool(not mixed and frag and H == 128 and _lib.get_variant("F32_SPLIT") != 0)
    # per layer: [max|W_ih| per direction (D)] [activation bound] [max|W_hh| per direction (D)] [max|W_ih| over the
    # directions] [1.0 = the bound of |h|]: the last two are operand ranges of the backward's fp16-split GEMMs
    RS = 2 * D + 3
    rng = torch.zeros((L * RS,), device=dev, dtype=f32) if want_range else None      # atomic-max targets
    base = 4
    for layer in range(L):
        dirs = [ps[base + 4 * d: base + 4 * d + 4] for d in range(D)]
        base += 4 * D
        K = dirs[0][0].shape[1]
        N = D * 4 * H
        use16 = bool(mixed and frag and act_is_bf16(layer, cfg, frag) and (ops.gate_ws_ok(K, H) or ops.dma_ok(K, N, rows)))
        wih = new((N, K))                      # fp32 image: the fp32 kernels' operand; small, and the fallbacks want it
        w16 = new((N, K), bf16) if use16 else None
        whh = new((D, 4 * H, H))
        bias = new((N,))
        # backward operand of dX = dP W_ih: W_ih^T as (K, N); bf16 when dP is bf16 and the LDS-DMA NT GEMM takes the shape
        wt = None
        if need_grad:
            wt16 = bool(mixed and ops.dma_ok(N, K, rows))
            wt = new((K, N), bf16 if wt16 else f32)
        if rng is not None:
            r0 = layer * RS
            for d, (w_ih, w_hh, _bi, _bh) in enumerate(dirs):
                plan.add_range(rng, r0 + d, w_ih)
                plan.add_range(rng, r0 + D + 1 + d, w_hh)
                plan.add_range(rng, r0 + 2 * D + 1, w_ih)
            plan.add_range(rng, r0 + 2 * D + 2, lnbound=True, factor=1.0)
            if layer == 0:
                if ps[2] is not None:
                    plan.add_range(rng, r0 + D, ps[2], ps[3], lnbound=True, factor=1.0 / (1.0 - p_in) if p_in < 1 else 1.0)
                    img[("gate_range", 0)] = rng[r0:r0 + D + 1]
                # else: no LayerNorm in front (ablation variant) -> no bound -> autograd runs this GEMM exact
            else:
                plan.add_range(rng, r0 + D, lnbound=True, factor=1.0 / (1.0 - p_lstm) if p_lstm < 1 else 1.0)
                img[("gate_range", layer)] = rng[r0:r0 + D + 1]
            img[("rec_range", layer)] = rng[r0 + D + 1:r0 + 2 * D + 1]
            # backward: (bound of the layer's input activations | None, max|W_ih| over the directions, bound of |h|)
            img[("bwd_range", layer)] = (rng[r0 + D:r0 + D + 1] if ("gate_range", layer) in img else None,
                                         rng[r0 + 2 * D + 1:r0 + 2 * D + 2], rng[r0 + 2 * D + 2:r0 + 2 * D + 3])
        for d, (w_ih, w_hh, b_ih, b_hh) in enumerate(dirs):
            plan.add(w_ih, wih, dst_ptr_off=d * 4 * H * K)
            if w16 is not None:
                plan.add(w_ih, w16, dst_ptr_off=d * 4 * H * K)
            plan.add(w_hh, whh, dst_ptr_off=d * 4 * H * H, ld_dst=H)
            plan.add(b_ih, bias, dst_ptr_off=d * 4 * H, src2=b_hh, ld_dst=4 * H)
            if wt is not None:
                plan.add(w_ih, wt, dst_ptr_off=d * 4 * H, ld_dst=N, transpose=True)
        img[("wih", layer)], img[("wih16", layer)], img[("whh", layer)] = wih, w16, whh
        img[("bias", layer)], img[("wihT", layer)] = bias, wt
    n = len(ps)
    a0w = ps[n - 10]
    W = H * D
    if a0w is not None:
        if mixed and ops.dma_ok(W, a0w.shape[0], rows):
            img["a0w16"] = new(tuple(a0w.shape), bf16)
            plan.add(a0w, img["a0w16"])
        if need_grad:
            t16 = bool(mixed and ops.dma_ok(a0w.shape[0], W, rows))
            img["a0wT"] = new((a0w.shape[1], a0w.shape[0]), bf16 if t16 else f32)
            plan.add(a0w, img["a0wT"], transpose=True)
    if need_grad:
        for key, i in (("c0wT", n - 6), ("c3wT", n - 4), ("c6wT", n - 2)):
            w = ps[i]
            img[key] = new((w.shape[1], w.shape[0]))
            plan.add(w, img[key], transpose=True)
    plan.run()
    return img
